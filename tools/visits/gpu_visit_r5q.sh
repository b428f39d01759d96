#!/bin/bash
# Round-5 GPU visit Q (last): counters of the 2^22-row shape with the final kernels; BASELINE configs[2] over eight lockstep ranks again.
TAG=${1:-r05_q}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash tools/pmc.sh ${TAG}_pmc_2p22 python $R/tools/probe.py 22 96 0 1
( timeout 900 python bench.py --log2-rows 22 --simulate-gpus 8 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>gpurun_out/${TAG}_sim22.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_simulated_8_ranks_2p22.json
python - <<P
import json
d = json.load(open("gpurun_out/${TAG}_bench_simulated_8_ranks_2p22.json"))
s = d["simulated_multi_gpu"]
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"), "sim", s.get("ranks"), s.get("slowest_rank_sum_ms"), s.get("projected_ms_per_proof"), s.get("same_proof_as_single_gpu"), s.get("all_ranks_same_proof"), s.get("error"))
print(json.dumps({k: max(v) for k, v in s.get("stage_ms_per_rank", {}).items()}))
print(json.dumps(s.get("column_split_bracket")))
P
