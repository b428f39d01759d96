#!/bin/bash
# Round-4 GPU visit E: host phases of the one-rank sharded proof (TVMH_TRACE), the valid-trace AIR dealt over the ranks (GPU tests + lockstep
# measurements with 2, 4, 8 ranks)
TAG=${1:-visit}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_sharded_host.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/${TAG}_pytest_gpu.log
( TVMH_TRACE=1 timeout 600 python bench.py --sharded --steps 3 --warmup 1 --no-extras --no-cpu-baseline 2>gpurun_out/${TAG}_sharded_trace.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_sharded_1rank_rccl.json
for N in 2 4 8; do
  ( timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --simulate-gpus $N 2>gpurun_out/${TAG}_sim$N.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_simulated_${N}_ranks.json
done
cat gpurun_out/${TAG}_pytest_gpu.log
grep "tvmh sharded" gpurun_out/${TAG}_sharded_trace.err | tail -24
python - <<P
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d = json.load(open(f))
        print(f, d["ms_per_step"], d["value"], d.get("verified", {}).get("accepted"))
        if "simulated_multi_gpu" in d:
            s = d["simulated_multi_gpu"]
            print("   sim", s.get("ranks"), s.get("slowest_rank_sum_ms"), s.get("projected_ms_per_proof"), s.get("bytes_sent_per_rank"), s.get("same_proof_as_single_gpu"), s.get("error"))
            print("   ", json.dumps({k: max(v) for k, v in s.get("stage_ms_per_rank", {}).items()}))
            print("   AIR per rank", s.get("stage_ms_per_rank", {}).get("AIR quotients"))
    except Exception as e:
        print(f, "unreadable:", e)
P
for f in gpurun_out/${TAG}_sim*.err; do echo $f; tail -2 $f; done
