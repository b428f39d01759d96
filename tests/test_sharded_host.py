"""The C++ host's sharded / coset-wise prover (triton_vm_amd/host/sharded_host.cpp) against the single-GPU C++ prover: the same
proof, word for word -- over 2, 4 and 8 ranks (the communicators between the contexts of one process, one proving thread per
rank; and two real processes over torch.distributed's gloo backend), with the Merkle trees split over the ranks and built
whole, FRI and STIR, LDT expansion 16 (quotient domain shorter than the LDT domain), from an execution trace (the
reference's snapshot digest, proof.rs:200-226), coset by coset on one rank, and under the reference's memory policy
(master_table.rs:268-271: a cached extension that does not fit is not an error)."""
import os
import socket
import sys
import threading

import numpy as np
import pytest

from triton_vm_amd import native_host
from triton_vm_amd.prover import Prover, StarkParameters

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 9


def host_library(ctx):
    backend = ctx.lib._name
    if getattr(ctx, "kind", "gpu") == "emu":
        return native_host.load_host_library(backend, os.path.join(os.path.dirname(backend), "libtriton_host_emu.so"))
    return native_host.load_host_library(backend)


def _params(kind, log2_rows=3):
    if kind == "fri":
        return StarkParameters(log2_rows, num_trace_randomizers=3, num_collinearity_checks=2)
    if kind == "fri16":
        return StarkParameters(log2_rows, num_trace_randomizers=3, num_collinearity_checks=2, log2_expansion=4)
    from triton_vm_amd import low_degree_test as ldt_module   # STIR at a security level that still has a quotienting round at this size

    stir = ldt_module.stark_stir(1 << log2_rows, security_level=8)
    p = StarkParameters(log2_rows, num_trace_randomizers=stir.num_trace_randomizers(), num_collinearity_checks=2)
    p.stir = stir
    return p


def _inputs(orc, p):
    rng = np.random.default_rng(77)
    n = p.trace.length
    return orc.random_elements(rng, (379, n)), orc.random_elements(rng, (91, n, 3))


def _new_context(ctx):
    from triton_vm_amd.capi import Context

    return Context(device=0, lib=ctx.lib)


def _single_proof(ctx, host, p, main_trace, aux_trace):
    py = Prover(ctx, p, main_trace, aux_trace, seed=SEED)
    if p.stir is not None:   # (tvmh_prove derives its STIR instance at security level 160; the Python host proves this small one --
        return py, py.prove().proof().words   # tests/test_native_host.py holds the two hosts' STIR proofs equal)
    native = native_host.NativeProver(ctx, host, p, py.main.d_trace, py.main.d_randomizers, py.aux.d_trace, py.aux.d_randomizers,
                                      py.quotient_randomizer)
    return py, native.prove()


def _run_ranks(world, body, comms=None):
    """one thread per rank; `body(rank)` -> result; exceptions are re-raised here"""
    results, errors = [None] * world, []

    def run(rank):
        try:
            results[rank] = body(rank)
        except BaseException as e:   # noqa: BLE001
            errors.append((rank, e))
            if comms is not None:
                comms.abort()   # the other ranks leave their collectives with an error instead of waiting for this one

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=1200)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads), "a rank hung"
    return results


# (CPU suite time; all of them on the GPU.  (4, False, "fri", False): the whole-tree path with a communicator -- gather_rows, then
# whole_tree and tvm_fri_commit_phase -- which tests/test_sharded_prover.py leaves to this file)
ON_THE_EMULATION_TOO = {(2, True, "fri", True), (8, True, "fri", False), (2, True, "fri16", False), (4, False, "fri", False)}


@pytest.mark.parametrize("world,split_trees,kind,lockstep", [(2, True, "fri", True), (4, False, "fri", False), (8, True, "fri", False), (8, True, "fri", True),
                                                             (2, True, "stir", False), (2, True, "fri16", False), (8, False, "fri16", False)])
def test_sharded_cpp_proof_equals_single_gpu_proof(ctx, orc, world, split_trees, kind, lockstep):
    if ctx.kind == "emu" and (world, split_trees, kind, lockstep) not in ON_THE_EMULATION_TOO:
        pytest.skip("on the GPU only (CPU suite time); tests/test_sharded_prover.py runs these shapes through the Python mirror on the emulation")
    host = host_library(ctx)
    p = _params(kind)
    main_trace, aux_trace = _inputs(orc, p)
    py, want = _single_proof(ctx, host, p, main_trace, aux_trace)
    comms = native_host.LocalComms(host, world, lockstep=lockstep)
    try:
        def rank_body(rank):
            c = _new_context(ctx)   # one context per proving thread (lib.rs:522-532)
            try:
                mine = Prover(c, p, main_trace, aux_trace, seed=SEED)   # the same traces and randomizers on every rank
                return native_host.prove_sharded(c, host, comms.ptrs[rank], p, mine.main.d_trace, mine.main.d_randomizers, mine.aux.d_trace,
                                                 mine.aux.d_randomizers, mine.quotient_randomizer, split_tree_min_leaves=0 if split_trees else 1 << 62,
                                                 stir_security_level=8)
            finally:
                c.close()

        proofs = _run_ranks(world, rank_body, comms)
        for rank, got in enumerate(proofs):
            assert got.size == want.size and (got == want).all(), rank
        if lockstep:
            report = comms.report()
            assert "main LDE" in report and len(report["main LDE"]) == world and all(ms > 0 for ms in report["AIR quotients"])
    finally:
        comms.close()


@pytest.mark.parametrize("world,kind,chunks,lockstep", [(2, "fri", 2, False), (8, "fri", 1, True), (4, "stir", 3, False), (8, "fri16", 2, False)])
def test_column_split_of_the_inverse_transforms_gives_the_same_proof(ctx, orc, world, kind, chunks, lockstep):
    """TVMH_OPTION_COLUMN_SPLIT (north_star's column sharding where it applies: rank r interpolates ITS columns, the coefficient
    forms are all-gathered in chunks, every rank extends all columns onto its cosets -- MasterTable::low_degree_extend_over) against
    the single-GPU proof, word for word, and the exchange shows up in the rank's statistics."""
    if ctx.kind == "emu" and world > 2:
        pytest.skip("on the GPU only (CPU suite time)")
    host = host_library(ctx)
    p = _params(kind)
    main_trace, aux_trace = _inputs(orc, p)
    py, want = _single_proof(ctx, host, p, main_trace, aux_trace)
    comms = native_host.LocalComms(host, world, lockstep=lockstep)
    host.tvmh_set_option(native_host.OPTION_COLUMN_SPLIT, chunks)
    try:
        def rank_body(rank):
            c = _new_context(ctx)
            try:
                mine = Prover(c, p, main_trace, aux_trace, seed=SEED)
                return native_host.prove_sharded(c, host, comms.ptrs[rank], p, mine.main.d_trace, mine.main.d_randomizers, mine.aux.d_trace,
                                                 mine.aux.d_randomizers, mine.quotient_randomizer, split_tree_min_leaves=0, stir_security_level=8)
            finally:
                c.close()

        for rank, got in enumerate(_run_ranks(world, rank_body, comms)):
            assert got.size == want.size and (got == want).all(), rank
    finally:
        host.tvmh_set_option(native_host.OPTION_COLUMN_SPLIT, 0)
        comms.close()


@pytest.mark.parametrize("passes", [2, 8])
def test_coset_wise_cpp_proof_equals_cached_proof(ctx, orc, passes):
    """the just-in-time path (stark.rs:805-1006, master_table.rs:470-503, 556-609) in the C++ host, one rank"""
    if ctx.kind == "emu" and passes == 8:
        pytest.skip("eight passes on the GPU only (CPU suite time)")
    host = host_library(ctx)
    p = _params("fri")
    main_trace, aux_trace = _inputs(orc, p)
    py, want = _single_proof(ctx, host, p, main_trace, aux_trace)
    got = native_host.prove_sharded(ctx, host, None, p, py.main.d_trace, py.main.d_randomizers, py.aux.d_trace, py.aux.d_randomizers,
                                    py.quotient_randomizer, jit_passes=passes)
    assert got.size == want.size and (got == want).all()


def _snapshot_inputs(orc):
    from tests import test_proof_snapshot as snap
    from tests import vm_fixture as vf
    from tests.test_fill import aet_arrays

    program, aet, public_input, output = vf.run("tiny")
    return aet_arrays(orc, aet), aet.padded_height(), snap.claim_of(orc, program, public_input, output), snap.prover_seed(snap.SEED_U64)


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_prove_execution_reproduces_the_reference_snapshot(ctx, orc, world):
    """Prover::prove(claim, aet) over the ranks: fill, pad, extend replicated, everything else split -- the proof hashes to the
    digest the reference holds (proof.rs:218-225), on every rank; the leaf digests went through the all-to-all"""
    from tests import test_proof_snapshot as snap
    from triton_vm_amd.proof_stream import Proof

    if ctx.kind == "emu":
        pytest.skip("on the GPU only (CPU suite time): test_two_gloo_processes_reproduce_the_reference_snapshot is the same proof on the emulation")
    host = host_library(ctx)
    aet, padded_height, claim, seed = _snapshot_inputs(orc)
    comms = native_host.LocalComms(host, world)
    try:
        def rank_body(rank):
            c = _new_context(ctx)
            try:
                return native_host.prove_execution_sharded(c, host, comms.ptrs[rank], aet, padded_height, claim, seed, jit_passes=1,
                                                           split_tree_min_leaves=0, profile=True)
            finally:
                c.close()

        for words, stats in _run_ranks(world, rank_body, comms):
            assert Proof(words).digest(ctx.lib) == snap.SNAPSHOT
            assert stats["world"] == world and stats["split_trees_built"] >= 3
            assert stats["exchanges"]["main leaf digests"]["calls"] == 1 and "AIR quotients" in stats["stage_ms"]
    finally:
        comms.close()


@pytest.mark.parametrize("world,lockstep", [(2, True), (8, False)])
def test_in_process_ranks_share_one_copy_of_the_replicated_tables(ctx, orc, world, lockstep):
    """TVMH_OPTION_SHARE_REPLICATED_TABLES (triton_host.hpp, tvmh_comm::share): rank 0 fills, pads and extends, the other ranks of the
    in-process communicator read its device arrays -- what lets bench.py --simulate-gpus put eight ranks of a 2^22-row proof on one
    GPU.  Same proof (the reference's snapshot digest) on every rank, twice in a row (the second proof releases the first one's
    tables), and the ranks other than 0 allocate no traces of their own."""
    from tests import test_proof_snapshot as snap
    from triton_vm_amd.proof_stream import Proof

    if ctx.kind == "emu" and world > 2:
        pytest.skip("eight ranks on the GPU only (CPU suite time)")
    host = host_library(ctx)
    aet, padded_height, claim, seed = _snapshot_inputs(orc)
    comms = native_host.LocalComms(host, world, lockstep=lockstep)
    contexts = [_new_context(ctx) for _ in range(world)]
    host.tvmh_set_option(native_host.OPTION_SHARE_REPLICATED_TABLES, 1)
    try:
        for _ in range(1 if ctx.kind == "emu" else 2):   # (the second proof releases the first one's tables: on the GPU; CPU suite time)
            def rank_body(rank):
                return native_host.prove_execution_sharded(contexts[rank], host, comms.ptrs[rank], aet, padded_height, claim, seed, jit_passes=1,
                                                           split_tree_min_leaves=0, profile=True)

            for words, stats in _run_ranks(world, rank_body, comms):
                assert Proof(words).digest(ctx.lib) == snap.SNAPSHOT
                assert stats["world"] == world and stats["split_trees_built"] >= 3
        if lockstep:
            report = comms.report()
            fill = report["trace tables (fill, pad, randomizers)"]
            assert fill[0] > 0 and "extend" in report
    finally:
        host.tvmh_set_option(native_host.OPTION_SHARE_REPLICATED_TABLES, 0)
        comms.close()   # (drops the tables the group keeps: rank 0's context must still be open)
        for c in contexts:
            c.close()


def test_memory_policy_with_a_communicator_is_decided_collectively(ctx, orc):
    """jit_passes = 0 over a communicator: the ranks' free memory differs (here: rank 1 has a memory limit, rank 0 has none), and a
    rank that restarted coset by coset on its own would issue collectives its peer does not expect.  Every rank plans with what ITS
    context can obtain, one all-gather makes the largest pass count everybody's -- both ranks report the same pass count, more than
    one, and the proof is the reference's."""
    from tests import test_proof_snapshot as snap
    from triton_vm_amd.proof_stream import Proof

    host = host_library(ctx)
    aet, padded_height, claim, seed = _snapshot_inputs(orc)
    p = StarkParameters(padded_height.bit_length() - 1)
    n, L = p.trace.length, p.ldt.length
    world = 2
    comms = native_host.LocalComms(host, world)
    contexts = [_new_context(ctx) for _ in range(world)]
    # rank 1 may hold its traces and a quarter of the extended rows (its share of the cached extension would be half of them)
    contexts[1].set_memory_limit(contexts[1].memory_held() + 8 * (379 + 273) * (2 * n + L // 4))
    try:
        def rank_body(rank):
            return native_host.prove_execution_sharded(contexts[rank], host, comms.ptrs[rank], aet, padded_height, claim, seed, jit_passes=0,
                                                       split_tree_min_leaves=0)

        results = _run_ranks(world, rank_body, comms)
        passes = [stats["passes"] for _, stats in results]
        assert passes[0] == passes[1] and passes[0] >= 2, passes
        for words, _ in results:
            assert Proof(words).digest(ctx.lib) == snap.SNAPSHOT
    finally:
        comms.close()
        for c in contexts:
            c.close()


def test_memory_policy_of_the_cpp_host(ctx, orc):
    """master_table.rs:268-271, stark.rs:730-768: when the cached extension does not fit (the context's memory limit stands in
    for a full device) tvmh_prove_execution_sharded with jit_passes = 0 starts over coset by coset; the proof does not change"""
    from tests import test_proof_snapshot as snap
    from triton_vm_amd.proof_stream import Proof

    host = host_library(ctx)
    aet, padded_height, claim, seed = _snapshot_inputs(orc)
    if ctx.kind != "emu":   # (CPU suite time: the unconstrained and the tvmh_prove_execution legs on the GPU only)
        words, stats = native_host.prove_execution_sharded(ctx, host, None, aet, padded_height, claim, seed, jit_passes=0)
        assert stats["passes"] == 1 and Proof(words).digest(ctx.lib) == snap.SNAPSHOT
    ctx.trim()
    p = StarkParameters(padded_height.bit_length() - 1)
    n, L = p.trace.length, p.ldt.length
    traces_bytes = 8 * n * (379 + 273)
    full_tables = 8 * L * (379 + 273)
    ctx.set_memory_limit(ctx.memory_held() + 2 * traces_bytes + full_tables // 2)
    try:
        words, stats = native_host.prove_execution_sharded(ctx, host, None, aet, padded_height, claim, seed, jit_passes=0)
    finally:
        ctx.set_memory_limit(0)
    assert stats["passes"] >= 2 and Proof(words).digest(ctx.lib) == snap.SNAPSHOT
    # and tvmh_prove_execution itself -- the entry point of the Rust shim's stage 2 -- takes the same fallback
    if ctx.kind == "emu":
        return
    ctx.trim()
    ctx.set_memory_limit(ctx.memory_held() + 2 * traces_bytes + full_tables // 2)
    try:
        words = native_host.prove_execution(ctx, host, aet, padded_height, claim, seed)
    finally:
        ctx.set_memory_limit(0)
    assert Proof(words).digest(ctx.lib) == snap.SNAPSHOT


# ---- two real processes over gloo -------------------------------------------------------------------------------------
def _gloo_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from oracle import oracle as orc
    from tests.emu_fixture import emu_context

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    ctx = emu_context()
    ctx.kind = "emu"
    host = host_library(ctx)
    comm = native_host.gloo_comm(dist)
    aet, padded_height, claim, seed = _snapshot_inputs(orc)
    words, stats = native_host.prove_execution_sharded(ctx, host, comm.ptr, aet, padded_height, claim, seed, jit_passes=1, split_tree_min_leaves=0)
    out.put((rank, words, stats))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


def test_two_gloo_processes_reproduce_the_reference_snapshot(orc):
    import queue
    import time

    import torch.multiprocessing as mp

    from tests import test_proof_snapshot as snap
    from tests.emu_fixture import emu_context
    from triton_vm_amd.proof_stream import Proof

    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpctx = mp.get_context("spawn")
    out = mpctx.Queue()
    procs = [mpctx.Process(target=_gloo_worker, args=(r, world, port, out)) for r in range(world)]
    for pr in procs:
        pr.start()
    results, deadline = {}, time.time() + 900
    while len(results) < world:
        try:
            rank, words, stats = out.get(timeout=5)
            results[rank] = (words, stats)
        except queue.Empty:
            assert all(pr.exitcode in (None, 0) for pr in procs), "a rank died"
            assert time.time() < deadline, "timed out"
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    ctx = emu_context()
    try:
        for rank in range(world):
            words, stats = results[rank]
            assert Proof(words).digest(ctx.lib) == snap.SNAPSHOT, rank
            assert stats["rank"] == rank and stats["split_trees_built"] >= 3
    finally:
        ctx.close()


# ---- two real GPUs over RCCL (runs as soon as two are visible; the boxes of the builder's rounds had one) -----------------
def _rccl_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import oracle as orc

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(rank)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)   # carries the ncclUniqueId; the proof's collectives are RCCL
    from triton_vm_amd import Context

    ctx = Context(device=rank)
    host = native_host.load_host_library()
    uid = torch.from_numpy(native_host.RcclComm.unique_id() if rank == 0 else np.zeros(128, np.uint8))
    dist.broadcast(uid, 0)
    comm = native_host.RcclComm(uid.numpy(), rank, world, rank)
    aet, padded_height, claim, seed = _snapshot_inputs(orc)
    words, stats = native_host.prove_execution_sharded(ctx, host, comm.ptr, aet, padded_height, claim, seed, jit_passes=1, split_tree_min_leaves=0)
    out.put((rank, words, stats))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.gpu
def test_two_gpus_over_rccl_reproduce_the_reference_snapshot(orc):
    import queue
    import time

    import torch
    import torch.multiprocessing as mp

    from tests import test_proof_snapshot as snap
    from triton_vm_amd import Context
    from triton_vm_amd.proof_stream import Proof

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpctx = mp.get_context("spawn")
    out = mpctx.Queue()
    procs = [mpctx.Process(target=_rccl_worker, args=(r, world, port, out)) for r in range(world)]
    for pr in procs:
        pr.start()
    results, deadline = {}, time.time() + 600
    while len(results) < world:
        try:
            rank, words, stats = out.get(timeout=5)
            results[rank] = (words, stats)
        except queue.Empty:
            assert all(pr.exitcode in (None, 0) for pr in procs), "a rank died"
            assert time.time() < deadline, "timed out"
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    ctx = Context(device=0)
    try:
        for rank in range(world):
            assert Proof(results[rank][0]).digest(ctx.lib) == snap.SNAPSHOT, rank
    finally:
        ctx.close()


@pytest.mark.gpu
def test_one_rank_over_rccl_reproduces_the_reference_snapshot(orc):
    """the RCCL communicator (triton_vm_amd/host/rccl_comm.cpp) with a single rank, in this process: every collective of the
    sharded proof runs through RCCL on the context's stream"""
    from tests import test_proof_snapshot as snap
    from triton_vm_amd import Context
    from triton_vm_amd.proof_stream import Proof

    ctx = Context(device=0)
    comm = None
    try:
        host = native_host.load_host_library()
        comm = native_host.RcclComm(native_host.RcclComm.unique_id(), 0, 1, 0)
        aet, padded_height, claim, seed = _snapshot_inputs(orc)
        words, stats = native_host.prove_execution_sharded(ctx, host, comm.ptr, aet, padded_height, claim, seed, jit_passes=1, split_tree_min_leaves=0)
        assert Proof(words).digest(ctx.lib) == snap.SNAPSHOT
        assert stats["split_trees_built"] >= 3 and stats["exchanges"]["quotient codeword"]["calls"] == 1
    finally:
        if comm is not None:
            comm.close()
        ctx.close()


# ---- the valid-trace AIR dealt over the ranks (sharded_host.cpp: quotient_codeword_by_classes) ---------------------------------
@pytest.mark.parametrize("world", [2, pytest.param(4, marks=pytest.mark.gpu), pytest.param(8, marks=pytest.mark.gpu)])
def test_valid_trace_air_over_the_ranks_yields_the_single_gpu_proof(ctx, orc, world):
    """prove_fib at 2^9 padded rows, security level 32 (60 trace randomizers: the degree bounds of the valid-trace classes hold):
    the ranks evaluate the constraint classes on the cosets dealt to them, exchange the values once and rebuild the quotient
    codeword -- the proof must be the single-GPU prover's, word for word (whose valid-trace AIR is held equal to the row-by-row
    one by tests/test_kernels_air.py and tests/test_gpu_baseline_configs.py)"""
    if world != 2 and ctx.kind == "emu":
        pytest.skip("one world size on the emulation (CPU suite time); all on the GPU")
    from oracle.vm import workload
    from triton_vm_amd.prover import Claim

    host = host_library(ctx)
    e = workload.execution("fib", 9)
    claim = Claim(e["program_digest"], e["public_input"], e["public_output"])
    seed = bytes(range(32))
    want = native_host.prove_execution(ctx, host, e["aet"], e["padded_height"], claim, seed, security_level=32, ldt="fri")
    comms = native_host.LocalComms(host, world)
    try:
        def rank_body(rank):
            c = _new_context(ctx)
            try:
                return native_host.prove_execution_sharded(c, host, comms.ptrs[rank], e["aet"], e["padded_height"], claim, seed, security_level=32,
                                                           ldt="fri", jit_passes=1, split_tree_min_leaves=0)
            finally:
                c.close()

        for words, stats in _run_ranks(world, rank_body, comms):
            assert words.size == want.size and (words == want).all()
            assert stats["exchanges"]["quotient class values"]["calls"] == 1 and "quotient codeword" not in stats["exchanges"]
    finally:
        comms.close()


def test_a_failing_rank_does_not_hang_the_others(ctx, orc):
    """one rank of two hands the prover a null trace: it fails before its first collective, aborts the group, and the other rank
    leaves its collective with an error instead of waiting forever"""
    host = host_library(ctx)
    p = _params("fri")
    main_trace, aux_trace = _inputs(orc, p)
    comms = native_host.LocalComms(host, 2)
    try:
        def rank_body(rank):
            c = _new_context(ctx)
            try:
                mine = Prover(c, p, main_trace, aux_trace, seed=SEED)
                bad = type("Null", (), {"ptr": None})()
                return native_host.prove_sharded(c, host, comms.ptrs[rank], p, bad if rank == 1 else mine.main.d_trace, mine.main.d_randomizers,
                                                 mine.aux.d_trace, mine.aux.d_randomizers, mine.quotient_randomizer, split_tree_min_leaves=0)
            finally:
                c.close()

        with pytest.raises(AssertionError) as info:
            _run_ranks(2, rank_body, comms)
        assert "tvmh_prove_sharded failed" in str(info.value)
    finally:
        comms.close()
