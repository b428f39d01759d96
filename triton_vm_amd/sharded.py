"""One proof across the GPUs of a node: coset (row) sharding of the extended master tables.

SURVEY.md section 8(e): the LDT / quotient domain g*<w_L> is X = L/N cosets of the trace domain; rank r of R
owns the cosets k = r (mod R), i.e. the extended rows i = r (mod R).  Those rows are themselves an arithmetic
domain -- offset g*w_L^r, generator w_L^R, length L/R -- so a rank's share of
`maybe_low_degree_extend_all_columns`, `hash_all_ldt_domain_rows` and `all_quotients_combined` (the next row
i + L/N stays on the same rank) are the *same* C-ABI calls on that local domain: no kernel knows about ranks.
The trace (5 GiB at 2^20 rows) is replicated.  Exchanges (RCCL all-gather over xGMI; gloo in the CPU tests):

    leaf digests of each master table   L x 40 B  (336 MB at 2^20), interleaved back into row order
    the quotient codeword               L x 24 B  (201 MB), likewise
    the opened rows (173 x 652 words)   all-gather of each owner's rows, put back into query order
    the out-of-domain rows              columns split over the ranks, all-gather of the shares (2 x 470 XFE)

Everything after the quotient codeword (segments, out-of-domain rows, combination, DEEP, FRI: ~55 ms of a
411 ms proof at 2^20 rows) is computed redundantly on every rank from identical inputs, so every rank derives
the same transcript and no broadcast is needed.  Splitting that tail is the next step for strong scaling.

Process set-up order matters: import torch (and select the device) BEFORE creating the Context -- torch brings
its own HIP runtime, and a process that has already initialised the system runtime through libtriton_hip.so
makes torch.cuda report "No HIP GPUs are available".  bench.py and the tests do it in that order.
"""
import numpy as np

from . import field, stark
from .arithmetic_domain import ArithmeticDomain
from .master_table import MasterTable
from .prover import NUM_AUX, NUM_MAIN, Prover


class _TensorBuffer:
    """A torch tensor of int64 words used where the C ABI wants a device pointer (`.ptr`)."""

    def __init__(self, tensor):
        self.tensor = tensor
        self.ptr = tensor.data_ptr()
        self.n_words = tensor.numel()

    def download(self, shape=None):
        a = self.tensor.cpu().numpy().view(np.uint64)
        return a.reshape(shape) if shape is not None else a


def local_domain(domain, rank, world):
    """rows i = rank (mod world) of `domain`, as a domain"""
    return ArithmeticDomain(field.mont_mul(domain.offset, field.mont_pow(domain.generator, rank)),
                            field.mont_pow(domain.generator, world), domain.length // world)


class ShardedProver(Prover):
    """Prover whose extended master tables are split by cosets over the ranks of a torch.distributed group."""

    def __init__(self, ctx, params, dist, device, main_trace=None, aux_trace=None, seed=1):
        import torch

        self.torch, self.dist, self.device = torch, dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        expansion = params.ldt.length // params.trace.length
        if params.quotient.length != params.ldt.length or expansion % self.world:
            raise ValueError("coset sharding needs |quotient| == |LDT| and a world size dividing |LDT| / |trace|")
        super().__init__(ctx, params, main_trace, aux_trace, seed)  # every rank holds the same traces (same seed)
        self.ldt_local = local_domain(params.ldt, self.rank, self.world)
        for mt in (self.main, self.aux):
            mt.quotient_domain = mt.ldt_domain = self.ldt_local

    # -- collectives ---------------------------------------------------------------------------------
    def _empty(self, n_words):
        return self.torch.empty(n_words, dtype=self.torch.int64, device=self.device)

    def _sync_device(self):
        self.ctx.sync()
        if self.device.type == "cuda":
            self.torch.cuda.synchronize(self.device)

    def _all_gather_rows(self, local, elem_words):
        """local: tensor [L/R * elem_words] holding this rank's rows -> tensor [L * elem_words] in row order
        (global row i = local row i // R of rank i % R)"""
        self._sync_device()  # the producing kernels ran on the context's stream
        gathered = self._empty(local.numel() * self.world)
        self.dist.all_gather_into_tensor(gathered, local)
        out = gathered.view(self.world, -1, elem_words).permute(1, 0, 2).contiguous().view(-1)
        self._sync_device()
        return out

    # -- the sharded steps -----------------------------------------------------------------------------
    def _commit_master_table(self, mt):
        ctx, L = self.ctx, self.p.ldt.length
        digests = self._empty(5 * self.ldt_local.length)
        ctx._check(ctx.lib.tvm_hash_rows(ctx.handle, mt._need_table(), self.ldt_local.length, digests.data_ptr()), "hash_rows")
        leaves = self._all_gather_rows(digests, 5)
        nodes = ctx.alloc(10 * L)
        ctx._check(ctx.lib.tvm_merkle_tree(ctx.handle, leaves.data_ptr(), L, nodes.ptr), "merkle_tree")
        self._keep = leaves  # until the stream has consumed it
        return nodes

    def _quotient_codeword(self, challenges, quotient_weights):
        ctx = self.ctx
        ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(63, 3)
        w = np.ascontiguousarray(quotient_weights, dtype=np.uint64).reshape(604, 3)
        local = self._empty(3 * self.ldt_local.length)
        ctx._check(ctx.lib.tvm_all_quotients_combined(ctx.handle, self.main._need_table(), self.aux._need_table(),
                                                      self.p.trace.c(), self.ldt_local.c(), ch.ctypes.data, w.ctypes.data,
                                                      local.data_ptr()), "all_quotients_combined")
        return _TensorBuffer(self._all_gather_rows(local, 3))

    def _all_gather_host(self, mine, cap):
        """all-gather of one equally sized (zero-padded to `cap` leading entries) block per rank: numpy [k, ...] ->
        numpy [world, cap, ...].  The payloads (out-of-domain rows, opened rows) are produced on the host by the C ABI
        and are a few hundred KB; the collective itself runs on device tensors (RCCL) / CPU tensors (gloo)."""
        block = np.zeros((cap,) + mine.shape[1:], np.uint64)
        block[:mine.shape[0]] = mine
        t = self.torch.from_numpy(block.view(np.int64).reshape(-1)).to(self.device)
        gathered = self._empty(t.numel() * self.world)
        self.dist.all_gather_into_tensor(gathered, t)
        return gathered.cpu().numpy().view(np.uint64).reshape((self.world,) + block.shape)

    def _out_of_domain_rows(self, mt, points):
        """columns split evenly over the ranks (the trace is replicated): rank r evaluates columns [r * per, (r+1) * per)
        at both points, the shares are all-gathered"""
        per = -(-mt.n_cols // self.world)
        c0 = min(self.rank * per, mt.n_cols)
        c1 = min(c0 + per, mt.n_cols)
        mine = mt.out_of_domain_rows(points, c0, c1 - c0)                      # [n_points, c1 - c0, 3]
        gathered = self._all_gather_host(np.ascontiguousarray(mine.transpose(1, 0, 2)), per)   # [world, per, n_points, 3]
        cols = gathered.reshape(self.world * per, len(points), 3)[:mt.n_cols]
        return np.ascontiguousarray(cols.transpose(1, 0, 2))

    def _reveal_master_rows(self, mt, row_indices):
        """row i of the extended table lives on rank i mod world (as its local row i // world); every rank opens the
        rows it owns, the shares are all-gathered and put back into query order"""
        idx = np.asarray(row_indices, dtype=np.uint64)
        owner = (idx % np.uint64(self.world)).astype(np.int64)
        width = mt.n_cols * mt.fk
        mine = np.nonzero(owner == self.rank)[0]
        rows = np.zeros((0, width), np.uint64)
        if mine.size:
            rows = mt.reveal_rows(idx[mine] // np.uint64(self.world)).reshape(mine.size, width)
        cap = max(int(np.bincount(owner, minlength=self.world).max()), 1)
        gathered = self._all_gather_host(rows, cap)                            # [world, cap, width]
        out = np.empty((idx.size, width), np.uint64)
        for r in range(self.world):
            pos = np.nonzero(owner == r)[0]
            out[pos] = gathered[r, :pos.size]
        return out.reshape((idx.size, mt.n_cols) + ((3,) if mt.fk == 3 else ()))
