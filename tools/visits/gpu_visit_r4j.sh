#!/bin/bash
# Round-4 GPU visit J: the whole GPU suite on the end-of-round tree, the lockstep measurements with 2 / 4 / 8 ranks (batched exchanges)
TAG=${1:-visit}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/${TAG}_pytest_gpu.log
for N in 2 4 8; do
  ( timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --simulate-gpus $N 2>gpurun_out/${TAG}_sim$N.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_simulated_${N}_ranks.json
done
( timeout 600 python bench.py --sharded --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>gpurun_out/${TAG}_sharded.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_sharded_1rank_rccl.json
cat gpurun_out/${TAG}_pytest_gpu.log
python - <<P
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench*.json")):
    try:
        d = json.load(open(f))
        print(f, d["ms_per_step"], d["value"], d.get("verified", {}).get("accepted"))
        if "simulated_multi_gpu" in d:
            s = d["simulated_multi_gpu"]
            print("   sim", s.get("ranks"), s.get("slowest_rank_sum_ms"), s.get("projected_ms_per_proof"), s.get("bytes_sent_per_rank"), s.get("collective_calls"), s.get("same_proof_as_single_gpu"), s.get("error"))
            print("   ", json.dumps({k: max(v) for k, v in s.get("stage_ms_per_rank", {}).items()}))
    except Exception as e:
        print(f, "unreadable:", e)
P
