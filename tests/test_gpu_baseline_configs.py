"""BASELINE.json's configurations FOR REAL, at their full sizes, on the MI355X: the programs are run in the oracle-side VM
(the stand-in for the reference's Rust VM: host work there too), the algebraic execution trace goes through the whole of
`Prover::prove(claim, aet)` on the device (/root/reference/triton-vm/src/stark.rs:331-719: fill, pad, extend, LDE, hashing,
AIR, quotient segments, DEEP, low-degree test, openings), and the proof is put through BOTH verifiers: the restated
`Verifier::verify` of the oracle (oracle/real_verifier.py, anchored to the reference-pinned proofs in
tests/test_verify_proof.py) and the product's own (triton_vm_amd/verifier.py).

    configs[1]  prove_fib (triton-dev-util/src/example_programs.rs:6-38) at 2^20 padded rows -- with LdtChoice::Fri and with
                the STIR that Stark::default() picks at this size (stark.rs:1944-1951)
    configs[2]  prove_fib at 2^22 padded rows (that configuration's height; here on ONE GPU: 163 GiB of extended tables)
    configs[3]  the many-u32-operations shape at 2^20 rows (a loop over the operations of example_programs.rs:70-99: the U32
                table sets the padded height)
    configs[4]  a hash-heavy program at 2^20 rows with FRI log-blowup 4, Stark::new(160, 4) (the recursive verifier itself
                lives outside the reference repository: a sponge loop fills the hash and cascade tables instead)

Size-independent properties checked besides acceptance: the C++ host's proof equals the Python host's word for word, a
second run with the same seed reproduces the proof (stark.rs:2434-2460's derandomization property), the exact row-by-row
AIR gives the proof of the valid-trace mode, and the same proof is rejected under a different claim.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (program, log2 padded height, low-degree test, log2 expansion, restated verifier too?)
# The restated STIR verifier is pure Python (~2.5 minutes per proof): it runs on ONE STIR proof.
CASES = [
    (("fib", None), 20, "fri", 2, True),
    (("fib", None), 20, "stir", 2, True),
    (("u32", None), 20, "fri", 2, True),
    (("sponge", None), 20, "fri", 4, True),
    (("fib", None), 22, "fri", 2, True),
    (("fib", None), 22, None, 2, False),     # Stark::default() at this height: STIR by the automatic rule
]


@pytest.fixture(scope="module")
def gctx():
    from triton_vm_amd import Context

    c = Context(device=0)
    yield c
    c.close()


_TRACES = {}


def execution(kind, log2):
    """one VM run per (program, height) for the whole module; one trace at a time (a 2^22-row AET is 1.3 GB of host arrays)"""
    from oracle.vm import workload
    from triton_vm_amd.proof_stream import Claim

    if (kind, log2) not in _TRACES:
        _TRACES.clear()
        e = workload.execution(kind, log2)
        e["claim"] = Claim(e["program_digest"], e["public_input"], e["public_output"])
        _TRACES[(kind, log2)] = e
    return _TRACES[(kind, log2)]


def _sharded_in_process(ctx, e, world, ldt, log2_expansion, split_all_trees, lockstep, seed, column_chunks=0):
    """one proof of execution `e` over `world` in-process ranks of the sharded C++ host (one thread, context and stream per rank,
    ONE copy of the replicated tables: TVMH_OPTION_SHARE_REPLICATED_TABLES) -> (every rank's proof words, rank 0's stats)"""
    import threading

    from triton_vm_amd import Context, native_host
    from triton_vm_amd.master_table import aet_to_device

    host = native_host.load_host_library()
    resident = aet_to_device(ctx, e["aet"])
    comms = native_host.LocalComms(host, world, lockstep=lockstep)
    contexts = [Context(device=0, lib=ctx.lib) for _ in range(world)]
    out, errors = [None] * world, []

    def run(r):
        try:
            out[r] = native_host.prove_execution_sharded(contexts[r], host, comms.ptrs[r], resident, e["padded_height"], e["claim"], seed, jit_passes=1,
                                                         log2_expansion=log2_expansion, ldt=ldt, split_tree_min_leaves=0 if split_all_trees else 1 << 21)
        except BaseException as err:   # noqa: BLE001
            errors.append((r, err))
            comms.abort()

    host.tvmh_set_option(native_host.OPTION_SHARE_REPLICATED_TABLES, 1)
    host.tvmh_set_option(native_host.OPTION_COLUMN_SPLIT, column_chunks)
    try:
        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=1200)
        assert not errors, errors
        assert all(not t.is_alive() for t in threads), "a rank hung"
    finally:
        host.tvmh_set_option(native_host.OPTION_SHARE_REPLICATED_TABLES, 0)
        host.tvmh_set_option(native_host.OPTION_COLUMN_SPLIT, 0)
        comms.close()
        for c in contexts:
            c.close()
        del resident
    return [o[0] for o in out], out[0][1]


# (log2 padded height, low-degree test, log2 expansion, every tree split?, lockstep?, chunks of the column split -- 0: coset sharding alone)
SHARDED_CASES = [
    (16, "fri", 2, True, False, 0),
    (16, "stir", 2, True, False, 0),
    (16, "fri", 2, False, True, 0),
    (16, "fri", 4, True, False, 0),
    (16, "stir", 2, True, False, 1),
    (20, "fri", 2, False, False, 0),
    (20, "stir", 2, False, True, 0),
    (20, "fri", 2, False, False, 2),
]


@pytest.mark.parametrize("log2,ldt,log2_expansion,split_all_trees,lockstep,column_chunks", SHARDED_CASES)
def test_sharded_cpp_host_over_eight_ranks_equals_the_single_gpu_proof(gctx, log2, ldt, log2_expansion, split_all_trees, lockstep, column_chunks):
    """The multi-GPU code path at real sizes (round 4's equality tests ran at 8 rows): prove_fib at 2^16 and 2^20 padded rows through
    `tvmh_prove_execution_sharded` over EIGHT ranks -- multi-chunk table extensions, multi-workgroup kernels on one-coset tables, the
    split-tree threshold of 2^21 leaves (at 2^20 rows: the three table trees and the first FRI rounds split, the later rounds
    whole), the valid-trace AIR dealt over the ranks, FRI and STIR, LDT expansion 16 (two cosets per rank; the quotient domain is
    the short one), and with the inverse transforms split by columns and the coefficients exchanged (north_star's column sharding,
    TVMH_OPTION_COLUMN_SPLIT) -- must emit the single-GPU proof, word for word, on every rank.  stark.rs:805-1006, master_table.rs:470-503."""
    from tests import test_proof_snapshot as snap
    from triton_vm_amd import native_host

    ctx = gctx
    e = execution("fib", log2)
    seed = snap.prover_seed(11)
    host = native_host.load_host_library()
    want = native_host.prove_execution(ctx, host, e["aet"], e["padded_height"], e["claim"], seed, log2_expansion=log2_expansion, ldt=ldt)
    ctx.trim()
    proofs, stats = _sharded_in_process(ctx, e, 8, ldt, log2_expansion, split_all_trees, lockstep, seed, column_chunks)
    for rank, got in enumerate(proofs):
        assert got.size == want.size and (got == want).all(), rank
    assert stats["world"] == 8 and stats["passes"] == 1
    assert ("main coefficients" in stats["exchanges"]) == bool(column_chunks)   # the column split's exchange (TVMH_OPTION_COLUMN_SPLIT)
    assert stats["split_trees_built"] >= (3 if (split_all_trees or log2 >= 18) else 0)
    assert stats["exchanges"]["main leaf digests"]["calls"] == 1
    ctx.trim()


@pytest.mark.parametrize("which,log2,ldt,log2_expansion,restated", CASES)
def test_baseline_config_proves_and_verifies(gctx, orc, which, log2, ldt, log2_expansion, restated):
    from oracle import proof_decode, real_verifier
    from tests import test_proof_snapshot as snap
    from triton_vm_amd import native_host, verifier as product
    from triton_vm_amd.prover import Prover

    ctx, kind = gctx, which[0]
    if ldt == "stir" and os.environ.get("TVM_SKIP_SLOW_VERIFIER") == "1":
        restated = False
    e = execution(kind, log2)
    arrays, claim, padded_height, heights = e["aet"], e["claim"], e["padded_height"], e["table_heights"]
    if kind == "u32":
        assert heights["U32"] > heights["Processor"]          # the co-processor table sets the height
    if kind == "sponge":
        assert heights["Hash"] > heights["Processor"]
    seed = snap.prover_seed(7)
    effective_ldt = ldt or ("fri" if log2 < 16 else "stir")

    prover = Prover.from_execution(ctx, arrays, padded_height, claim, seed, ldt=ldt, log2_expansion=log2_expansion)
    assert (prover.p.stir is not None) == (effective_ldt == "stir")
    assert prover.p.ldt.length == (1 << (log2 + 1 + log2_expansion))
    proof = prover.prove().proof()
    prover.release()
    del prover

    # Verifier::verify, twice over
    kw = dict(log2_expansion=log2_expansion)
    accepted_at = product.Verifier(ctx, ldt=ldt, **kw).verify(claim, proof.words)
    assert len(accepted_at) >= 160 // 2
    if restated:
        view = proof_decode.VerifierView(proof.words)      # the oracle's own decoder and sponge: nothing of the product
        assert real_verifier.verify(view, claim, ldt_choice=effective_ldt, **kw) == accepted_at
    from triton_vm_amd.proof_stream import Claim

    wrong = Claim(e["program_digest"], e["public_input"], list(e["public_output"]) + [1])
    with pytest.raises(product.VerificationError):
        product.Verifier(ctx, ldt=ldt, **kw).verify(wrong, proof.words)

    # the C++ host's Prover::prove(claim, aet) yields the same words, from host arrays and from a device-resident trace
    host_lib = native_host.load_host_library()
    words = native_host.prove_execution(ctx, host_lib, arrays, padded_height, claim, seed, log2_expansion=log2_expansion, ldt=ldt)
    assert words.size == proof.words.size and (words == proof.words).all()
    if log2 == 20 and ldt == "fri" and kind == "fib":
        from triton_vm_amd.master_table import aet_to_device

        resident = aet_to_device(ctx, arrays)
        again = native_host.prove_execution(ctx, host_lib, resident, padded_height, claim, seed, log2_expansion=log2_expansion, ldt=ldt)
        assert (again == proof.words).all()
        del resident
        # the exact row-by-row AIR (what the reference computes) instead of valid-trace mode: the same proof
        exact = Prover.from_execution(ctx, arrays, padded_height, claim, seed, ldt=ldt, log2_expansion=log2_expansion,
                                      assume_valid_trace=False)
        exact_proof = exact.prove().proof()
        exact.release()
        assert (exact_proof.words == proof.words).all()
    ctx.trim()


def test_configs2_height_over_eight_ranks_equals_the_single_gpu_proof(gctx):
    """BASELINE configs[2] -- prove_fib at 2^22 padded rows over 8 ranks -- through the code that would serve it: eight in-process
    ranks of the sharded C++ host in lockstep on this one GPU (21.8 GB of traces once, 8 x 20.4 GiB of one-coset tables), every
    rank's proof word for word the single-GPU proof."""
    from tests import test_proof_snapshot as snap
    from triton_vm_amd import native_host

    ctx = gctx
    ctx.trim()
    available, total = ctx.memory_info()
    if total < (250 << 30) or available < (235 << 30):   # 21.8 GB of traces + 8 x (20.4 GiB of tables + intermediates) on ONE device (268 GiB)
        pytest.skip(f"needs (nearly all of) the 288 GB of an MI355X: {available >> 30} GiB obtainable of {total >> 30}")
    e = execution("fib", 22)
    seed = snap.prover_seed(12)
    host = native_host.load_host_library()
    want = native_host.prove_execution(ctx, host, e["aet"], e["padded_height"], e["claim"], seed, ldt="fri")
    ctx.trim()
    proofs, stats = _sharded_in_process(ctx, e, 8, "fri", 2, False, True, seed)
    for rank, got in enumerate(proofs):
        assert got.size == want.size and (got == want).all(), rank
    assert stats["split_trees_built"] >= 4
    ctx.trim()
