"""One proof split over 2, 4 and 8 ranks (triton_vm_amd/sharded.py: coset sharding, gloo on CPU, the kernels on the
TEST-ONLY fiber emulation) must commit to the same roots, sample the same challenges and hand FRI the same
combination codeword as the single-process prover on the same traces, and produce the same proof word for word --
with the Merkle trees built redundantly and with the trees split over the ranks (subtree per rank, exchanged roots,
authentication nodes fetched from their owners).  World size 8 is the full node: with the
default expansion factor every rank then owns exactly one coset of the trace domain."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG2_ROWS, H, QUERIES, SEED = 3, 3, 2, 5


def _traces(n):
    from oracle import oracle as orc

    rng = np.random.default_rng(77)
    return orc.random_elements(rng, (379, n)), orc.random_elements(rng, (91, n, 3))


def _params(ldt):
    from triton_vm_amd.prover import StarkParameters

    if ldt == "fri":
        return StarkParameters(LOG2_ROWS, num_trace_randomizers=H, num_collinearity_checks=QUERIES)
    if ldt == "fri16":   # LDT expansion 16 (BASELINE config 5's FRI log-blowup 4): the quotient domain is a quarter of the LDT domain
        return StarkParameters(LOG2_ROWS, num_trace_randomizers=H, num_collinearity_checks=QUERIES, log2_expansion=4)
    from triton_vm_amd import low_degree_test as ldt_module   # STIR at a security level that still has a quotienting round at this size

    stir = ldt_module.stark_stir(1 << LOG2_ROWS, security_level=8)
    p = StarkParameters(LOG2_ROWS, num_trace_randomizers=stir.num_trace_randomizers(), num_collinearity_checks=QUERIES)
    p.stir = stir
    return p


def _capture(prover):
    prover.capture = {}
    proof = prover.prove().proof().words
    c = prover.capture
    keys = ("main_root", "aux_root", "quot_root", "challenges", "alpha", "ood_main", "ood_aux", "combination")
    return {k: np.array(c[k]) for k in keys} | {"last_polynomial": prover.last_polynomial,
                                                "main rows": prover.opened["main"], "aux rows": prover.opened["aux"],
                                                "proof": np.array(proof)}


def _worker(rank, world, port, out, split_trees, ldt):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from tests.emu_fixture import emu_context
    from triton_vm_amd.prover import StarkParameters
    from triton_vm_amd.sharded import ShardedProver

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    ctx = emu_context()
    p = _params(ldt)
    main_trace, aux_trace = _traces(p.trace.length)
    prover = ShardedProver(ctx, p, dist, torch.device("cpu"), main_trace, aux_trace, seed=SEED)
    if split_trees:
        prover.split_tree_min_leaves = 0   # every tree with at least two leaves per rank is built split (production: >= 2^21 leaves)
    got = _capture(prover)
    # three table trees and the FRI rounds with at least two leaves per rank
    assert getattr(prover, "split_trees_built", 0) >= ((4 if ldt.startswith("fri") else 3) if split_trees else 0) and (split_trees or not hasattr(prover, "split_trees_built"))
    out.put((rank, got))
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.parametrize("world,split_trees,ldt", [(2, True, "fri"), (4, False, "fri"), (8, True, "fri"), (2, True, "stir"), (2, True, "fri16"),
                                                   (8, False, "fri16")])
def test_sharded_proof_equals_single_process_proof(world, split_trees, ldt):
    import torch.multiprocessing as mp

    if (world, split_trees, ldt) in ((4, False, "fri"), (2, True, "fri16")):
        pytest.skip("the production host's tests cover these shapes (tests/test_sharded_host.py); the Python mirror keeps four (CPU suite time)")

    from tests.emu_fixture import emu_context
    from triton_vm_amd.prover import Prover, StarkParameters

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpctx = mp.get_context("spawn")
    out = mpctx.Queue()
    procs = [mpctx.Process(target=_worker, args=(r, world, port, out, split_trees, ldt)) for r in range(world)]
    for pr in procs:
        pr.start()

    ctx = emu_context()
    p = _params(ldt)
    main_trace, aux_trace = _traces(p.trace.length)
    single = Prover(ctx, p, main_trace, aux_trace, seed=SEED)
    want = _capture(single)
    ctx.close()

    results = {}
    import queue
    import time

    deadline = time.time() + 900
    while len(results) < world:
        try:
            rank, got = out.get(timeout=5)
            results[rank] = got
        except queue.Empty:
            assert all(pr.exitcode in (None, 0) for pr in procs), "a rank died"
            assert time.time() < deadline, "timed out"
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    for rank in range(world):
        for key, value in want.items():
            assert (results[rank][key] == value).all(), (rank, key)


def _execution_worker(rank, world, port, out, backend):
    """Prover::prove(claim, aet) over the ranks (ShardedProver.from_execution): `halt` at security level 32"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import oracle as orc
    from tests import test_proof_snapshot as snap, vm_fixture as vf
    from tests.test_fill import aet_arrays
    from triton_vm_amd.sharded import ShardedProver

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if backend == "nccl":
        torch.cuda.set_device(rank)
        device = torch.device("cuda", rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
        from triton_vm_amd import Context

        ctx = Context(device=rank)
        which, level = ("fib", 100), 160
    else:
        device = torch.device("cpu")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        from tests.emu_fixture import emu_context

        ctx = emu_context()
        which, level = "halt", 32
    program, aet, public_input, output = vf.run(which)
    claim = snap.claim_of(orc, program, public_input, output)
    prover = ShardedProver.from_execution(ctx, dist, device, aet_arrays(orc, aet), aet.padded_height(), claim, snap.prover_seed(3),
                                          security_level=level, ldt="fri")
    prover.split_tree_min_leaves = 0
    words = prover.prove().proof().words
    out.put((rank, np.array(words), prover.leaf_exchange))
    dist.destroy_process_group()
    ctx.close()


def _run_execution_ranks(world, backend, single_ctx, which, level):
    import queue
    import time

    import torch.multiprocessing as mp

    from oracle import oracle as orc
    from tests import test_proof_snapshot as snap, vm_fixture as vf
    from tests.test_fill import aet_arrays
    from triton_vm_amd.prover import Prover

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpctx = mp.get_context("spawn")
    out = mpctx.Queue()
    procs = [mpctx.Process(target=_execution_worker, args=(r, world, port, out, backend)) for r in range(world)]
    for pr in procs:
        pr.start()
    program, aet, public_input, output = vf.run(which)
    claim = snap.claim_of(orc, program, public_input, output)
    want = Prover.from_execution(single_ctx, aet_arrays(orc, aet), aet.padded_height(), claim, snap.prover_seed(3), security_level=level,
                                 ldt="fri").prove().proof().words
    results, deadline = {}, time.time() + 900
    while len(results) < world:
        try:
            rank, words, exchange = out.get(timeout=5)
            results[rank] = (words, exchange)
        except queue.Empty:
            assert all(pr.exitcode in (None, 0) for pr in procs), "a rank died"
            assert time.time() < deadline, "timed out"
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    for rank in range(world):
        words, exchange = results[rank]
        assert exchange == "all_to_all"                       # the leaf digests travelled once, to the rank that builds on them
        assert words.size == want.size and (words == want).all(), rank


def test_sharded_prove_execution_equals_single_process_proof():
    """fill, pad, extend replicated on two gloo ranks, the extended tables split: the reference-shaped proof, word for word"""
    from tests.emu_fixture import emu_context

    ctx = emu_context()
    try:
        _run_execution_ranks(2, "gloo", ctx, "halt", 32)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_sharded_prove_execution_over_rccl_on_two_gpus():
    """the same over RCCL with one rank per GPU -- skipped on a single-GPU box, so the first multi-GPU node exercises the
    all-to-all of leaf digests, the all-gather of the quotient codeword and the split trees for real"""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    from triton_vm_amd import Context

    ctx = Context(device=0)
    try:
        _run_execution_ranks(2, "nccl", ctx, ("fib", 100), 160)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_sharded_prover_on_one_gpu_matches_plain_prover():
    """The torch-tensor / RCCL plumbing of ShardedProver on the real device (a one-rank nccl group: the
    collectives are identities, the data path through torch-owned device memory is not)."""
    import torch
    import torch.distributed as dist

    from triton_vm_amd import Context
    from triton_vm_amd.prover import Prover, StarkParameters
    from triton_vm_amd.sharded import ShardedProver

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ctx = Context(0)
        p = StarkParameters(10)
        want = _capture(Prover(ctx, p, seed=9))
        for split_trees in (False, True):
            prover = ShardedProver(ctx, p, dist, torch.device("cuda", 0), seed=9)
            if split_trees:  # one rank: the "subtree" is the whole tree, the exchanges run through RCCL all the same
                prover.split_tree_min_leaves = 0
            got = _capture(prover)
            assert bool(getattr(prover, "split_trees_built", 0)) == split_trees
            for key, value in want.items():
                assert (got[key] == value).all(), (split_trees, key)
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("log_leaves", [4, 11, 23, 40])
def test_split_tree_locations_against_the_heap_layout(world, log_leaves):
    """where a node of the whole tree lives once the lowest levels are split into `world` subtrees by contiguous leaf
    ranges: brute force -- walk down from each subtree root and number its nodes in heap order"""
    from triton_vm_amd.sharded import split_tree_locations

    rng = np.random.default_rng(world * 100 + log_leaves)
    n = 1 << log_leaves
    picks = np.unique(np.concatenate([[1, 2, 3, world, 2 * world - 1, min(2 * world, 2 * n - 1), n, 2 * n - 1, n - 1],
                                      rng.integers(1, 2 * n, 500)]))
    in_top, owner, local = split_tree_locations(picks, world)
    for k, top, r, loc in zip(picks.tolist(), in_top.tolist(), owner.tolist(), local.tolist()):
        if k < 2 * world:
            assert top and loc == k
            continue
        # climb from k to its ancestor at the depth of the subtree roots, recording the path
        path, a = [], k
        while a >= 2 * world:
            path.append(a & 1)
            a >>= 1
        assert not top and r == a - world                 # the a-th node of that depth is subtree a - world
        node = 1
        for bit in reversed(path):
            node = 2 * node + bit
        assert loc == node, (k, world)
