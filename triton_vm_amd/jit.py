"""The memory-lean formulation of the hot path: extended tables evaluated coset by coset, never all at once.

ROLE (round 6): the PRODUCTION host is the C++ one (triton_vm_amd/host/: what bench.py times and what a Rust binding would call);
this Python module is its mirror for the parity tests -- test scaffolding above the C ABI, kept word-for-word equal to the C++ host by
tests/test_native_host.py and tests/test_sharded_host.py.  No algorithm lives here.

Mirrors the reference's "just in time" branches -- Prover::compute_quotient_segments_with_jit_lde
(/root/reference/triton-vm/src/stark.rs:805-1006), the JIT branch of hash_all_ldt_domain_rows
(master_table.rs:470-503) and of reveal_rows (master_table.rs:556-609) -- which it takes when the cached
extension does not fit in memory (master_table.rs:268-271).  As in the reference, nothing is cached: the
tables are re-evaluated for committing, for the quotients and for the openings.

The extended rows i = r (mod R) form the arithmetic domain (offset * generator^r, generator^R, length / R):
pass r of R extends, hashes or evaluates the AIR on that domain with the ordinary C-ABI calls (the "next" row
of a row stays in its group as long as R divides |LDT| / |trace|), and tvm_scatter_strided puts the group's
digests / quotient values back into row order.  Peak table memory is 1/R of the cached path
(41 GiB / R at 2^20 rows), which is what a proof of 2^23 rows or of expansion factor 32 needs on one GPU.
The same decomposition, with the passes on different GPUs, is triton_vm_amd/sharded.py.
"""
import gc

import numpy as np

from .prover import Prover
from .sharded import local_domain


class JitProver(Prover):
    """Prover that never holds more than 1/passes of an extended master table."""

    def __init__(self, ctx, params, passes, main_trace=None, aux_trace=None, seed=1):
        expansion = params.ldt.length // params.trace.length
        if params.quotient.length != params.ldt.length or passes < 1 or expansion % passes:
            raise ValueError("coset-wise evaluation needs |quotient| == |LDT| and a pass count dividing |LDT| / |trace|")
        super().__init__(ctx, params, main_trace, aux_trace, seed)
        self.passes = passes
        self.groups = [local_domain(params.ldt, r, passes) for r in range(passes)]

    def _extend_on(self, mt, r):
        mt.quotient_domain = mt.ldt_domain = self.groups[r]
        mt.maybe_low_degree_extend_all_columns()  # frees the previous group's table first

    def _extend_master_table(self, mt):
        pass  # nothing is cached

    def _commit_master_table(self, mt):
        ctx, lib, L = self.ctx, self.ctx.lib, self.p.ldt.length
        n_local = L // self.passes
        nodes = ctx.alloc(10 * L)
        digests = ctx.alloc(5 * n_local)
        for r in range(self.passes):
            self._extend_on(mt, r)
            ctx._check(lib.tvm_hash_rows(ctx.handle, mt._need_table(), n_local, digests.ptr), "hash_rows")
            ctx._check(lib.tvm_scatter_strided(ctx.handle, digests.ptr, 5, n_local, self.passes, r, nodes.ptr + 40 * L),
                       "scatter")
        mt.clear_cache()
        ctx._check(lib.tvm_merkle_tree(ctx.handle, nodes.ptr + 40 * L, L, nodes.ptr), "merkle_tree")
        return nodes

    def _quotient_codeword(self, challenges, quotient_weights):
        ctx, lib, L = self.ctx, self.ctx.lib, self.p.ldt.length
        n_local = L // self.passes
        ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(63, 3)
        w = np.ascontiguousarray(quotient_weights, dtype=np.uint64).reshape(604, 3)
        out, local = ctx.alloc(3 * L), ctx.alloc(3 * n_local)
        for r in range(self.passes):
            self._extend_on(self.main, r)
            self._extend_on(self.aux, r)
            ctx._check(lib.tvm_all_quotients_combined(ctx.handle, self.main._need_table(), self.aux._need_table(),
                                                      self.p.trace.c(), self.groups[r].c(), ch.ctypes.data, w.ctypes.data,
                                                      local.ptr), "all_quotients_combined")
            ctx._check(lib.tvm_scatter_strided(ctx.handle, local.ptr, 3, n_local, self.passes, r, out.ptr), "scatter")
        self.main.clear_cache()
        self.aux.clear_cache()
        return out

    def _reveal_master_rows(self, mt, row_indices):
        idx = np.asarray(row_indices, dtype=np.uint64)
        width = mt.n_cols * mt.fk
        rows = np.zeros((idx.size, width), np.uint64)
        for r in range(self.passes):
            mine = np.nonzero(idx % np.uint64(self.passes) == np.uint64(r))[0]
            if mine.size:
                self._extend_on(mt, r)
                rows[mine] = mt.reveal_rows(idx[mine] // np.uint64(self.passes)).reshape(mine.size, width)
        mt.clear_cache()
        return rows.reshape((idx.size, mt.n_cols) + ((3,) if mt.fk == 3 else ()))


def prove(ctx, params, main_trace=None, aux_trace=None, seed=1, capture=None):
    """The hot path with the reference's memory policy (master_table.rs:258-271, stark.rs:730-768): try the cached
    extension; if the device (or the context's memory limit) cannot hold it, start over on the coset-wise path with as
    few passes as fit.  The transcript is deterministic, so the restarted proof is the proof the cached path would have
    produced.  Returns (prover, proof_stream)."""
    from .capi import ERR_OUT_OF_MEMORY, TritonHipError

    expansion = params.ldt.length // params.trace.length
    passes = 0
    while True:
        prover = (JitProver(ctx, params, passes, main_trace, aux_trace, seed) if passes
                  else Prover(ctx, params, main_trace, aux_trace, seed))
        prover.capture = capture
        try:
            return prover, prover.prove()
        except TritonHipError as e:
            if e.status != ERR_OUT_OF_MEMORY or passes >= expansion:
                raise
        # outside the except block: the exception (and with it the traceback that references the failed attempt's
        # frames and their multi-GiB DeviceBuffers) is gone, so the collection below really frees them
        prover.release()
        del prover
        gc.collect()
        ctx.trim()
        passes = max(2 * passes, 2)
        while expansion % passes:
            passes += 1
