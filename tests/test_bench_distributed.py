"""The N > 1 path of bench.py on CPU: two gloo ranks run bench.timed_steps (the driver's timing contract:
barrier on both sides, MAX over ranks) around stand-in steps of different length.  Ranks prove independent
instances, so this harness is all of the multi-process logic."""
import os
import socket
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    import bench

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.02 * (rank + 1))  # rank 1 is the slow one

    elapsed = bench.timed_steps(step, steps=3, warmup=1, device_sync=lambda: None, dist=dist, device="cpu")
    out.put((rank, elapsed, len(calls)))
    dist.destroy_process_group()


def test_two_ranks_report_the_slowest():
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, e0, c0), (r1, e1, c1) = got
    assert (c0, c1) == (4, 4)                      # 1 warm-up + 3 timed steps on every rank
    assert e0 == pytest.approx(e1)                 # both ranks hold the max
    assert e0 >= 3 * 0.04 - 1e-3                   # ... which is the slow rank's 3 x 40 ms
    assert e0 < 3 * 0.04 + 0.5


def test_single_process_needs_no_collective():
    sys.path.insert(0, ROOT)
    import bench

    n = []
    assert bench.timed_steps(lambda: n.append(1), steps=2, warmup=1, device_sync=lambda: None) >= 0
    assert len(n) == 3


@pytest.mark.parametrize("mode", ["sharded"])   # "replicas" runs in test_bench_script_spawns_its_own_ranks
def test_bench_script_runs_under_torchrun_with_two_ranks(mode):
    """bench.py itself, launched the way the driver launches it for N = 2 (torch.distributed.run, one process per
    rank), on CPU: gloo instead of RCCL and the test-only emulation of the kernels (tests/bench_on_emulation.py replaces
    bench.py's process set-up hooks; bench.py itself carries no test switch).  The
    sharded mode runs collectives inside prove(); a rank that skips one of them would hang this test."""
    import json
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "bench_on_emulation.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--data", "synthetic", "--log2-rows", "3", "--trace-randomizers", "3", "--queries", "2", "--no-cpu-baseline"] + (["--replicas"] if mode == "replicas" else [])
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 1 and rec["value"] > 0
    assert rec["scaling"] == ("strong" if mode == "sharded" else "weak")
    assert {"roofline", "stage_ms", "config"} <= set(rec)


def test_bench_script_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (the driver's N > 1 form may be either): the script starts the two
    ranks itself through torch.distributed.run and rank 0 prints the one JSON line."""
    import json
    import subprocess

    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_on_emulation.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--data", "synthetic",
           "--log2-rows", "3", "--trace-randomizers", "3", "--queries", "2", "--no-cpu-baseline", "--replicas"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak"


@pytest.mark.gpu
def test_bench_sharded_code_path_over_rccl_with_one_rank():
    """`bench.py --sharded`: the code path of the driver's N > 1 runs -- process group over RCCL, ShardedProver.from_execution
    on the real prove_fib trace, leaf digests through the all-to-all, split Merkle trees, all-gather of the quotient codeword,
    the proof verified -- with a single rank on the single-GPU box (the collectives are identities, the plumbing is not)."""
    import json
    import subprocess

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--sharded", "--split-all-trees", "--steps", "1", "--warmup", "0", "--log2-rows", "16",
           "--no-cpu-baseline", "--no-extras"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["data"] == "real" and rec["verified"]["accepted"] and rec["n_gpus"] == 1
    assert "one proof over" in rec["config"]["parallelism"]
    # the C++ sharded host over the RCCL communicator: its own stage times and exchanges are in the line, every tree was built split
    assert "sharded_host.cpp" in rec["config"]["host"] and rec["ranks"][0]["split_trees_built"] >= 4
    assert rec["ranks"][0]["exchanges"]["main leaf digests"]["calls"] == 1


@pytest.mark.gpu
def test_bench_lockstep_measurement_of_the_multi_gpu_code_path():
    """`bench.py --simulate-gpus 4`: four ranks of the sharded C++ prover in one process on one GPU, in lockstep --
    every rank proves the single-GPU proof, and the line carries per-rank stage times and the bytes of the exchanges"""
    import json
    import subprocess

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--simulate-gpus", "4", "--steps", "1", "--warmup", "0", "--log2-rows", "16",
           "--no-cpu-baseline", "--no-extras"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    sim = rec["simulated_multi_gpu"]
    assert sim["ranks"] == 4 and sim["all_ranks_same_proof"] and sim["same_proof_as_single_gpu"], sim.get("error")
    assert len(sim["stage_ms_per_rank"]["AIR quotients"]) == 4 and sim["bytes_sent_per_rank"] > 0
    assert sim["projected_ms_per_proof"] > 0 and sim["slowest_rank_sum_ms"] > 0


def test_bench_memory_policy():
    """`bench.py --memory-policy` (the C++ host's sharded entry with jit_passes = 0 on one GPU: the reference's policy,
    master_table.rs:268-271) -- on the emulation, where the trace fits: one pass, the proof verified, the host's own stage times"""
    import json
    import subprocess

    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_on_emulation.py"), "--steps", "1", "--warmup", "0", "--log2-rows", "9",
           "--no-cpu-baseline", "--no-extras", "--memory-policy"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["memory_policy"]["passes"] == 1 and rec["verified"]["accepted"] and rec["config"]["parallelism"] == "single GPU"
    assert set(rec["stage_ms"]) >= {"main LDE", "main Merkle", "AIR quotients", "FRI"}   # (the sharded entry's own stage times)
