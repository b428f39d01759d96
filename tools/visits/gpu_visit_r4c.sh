#!/bin/bash
# Round-4 GPU visit C: the whole GPU suite, the sharded code path over RCCL (production tree threshold), the lockstep measurement with 2 and 4
# ranks, LDE chunk widths at 2^21 / 2^22 rows.
TAG=${1:-visit}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/${TAG}_pytest_gpu.log
( timeout 600 python bench.py --sharded --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>gpurun_out/${TAG}_sharded.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_sharded_1rank_rccl.json
for N in 2 4; do
  ( timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --simulate-gpus $N 2>gpurun_out/${TAG}_sim$N.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_simulated_${N}_ranks.json
done
for CH in 64 96; do
  ( TVM_LDE_CHUNK=$CH timeout 600 python bench.py --log2-rows 22 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>gpurun_out/${TAG}_2p22_chunk$CH.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_2p22_chunk$CH.json
done
( TVM_LDE_CHUNK=96 timeout 600 python bench.py --log2-rows 21 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>gpurun_out/${TAG}_2p21_chunk96.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_2p21_chunk96.json
cat gpurun_out/${TAG}_pytest_gpu.log
python - <<P
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d = json.load(open(f))
        print(f, d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d.get("verified", {}).get("accepted"))
        if "simulated_multi_gpu" in d:
            s = d["simulated_multi_gpu"]
            print("   sim", s.get("ranks"), s.get("slowest_rank_sum_ms"), s.get("projected_ms_per_proof"), s.get("bytes_sent_per_rank"), s.get("same_proof_as_single_gpu"), s.get("error"))
            print("   ", json.dumps({k: max(v) for k, v in s.get("stage_ms_per_rank", {}).items()}))
    except Exception as e:
        print(f, "unreadable:", e)
P
tail -3 gpurun_out/${TAG}_*.err
