#!/bin/bash
# Round 5, after the closing visit: the lockstep scaling curve on the final tree (2 and 4 ranks at 2^20 rows; 8 is in the default line),
# configs[4]'s shape over 8 ranks, configs[3]'s shape on one GPU.
TAG=${1:-r05_s}
mkdir -p gpurun_out
export TMPDIR=/tmp
for n in 2 4; do
( timeout 600 python bench.py --simulate-gpus $n --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>gpurun_out/${TAG}_sim$n.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_simulated_${n}_ranks_2p20.json
done
( timeout 900 python bench.py --program sponge --log2-expansion 4 --simulate-gpus 8 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>gpurun_out/${TAG}_sim_sponge.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_simulated_8_ranks_sponge_blowup4.json
( timeout 600 python bench.py --program u32 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>gpurun_out/${TAG}_u32.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_u32_2p20.json
python - <<P
import json
for f in ("simulated_2_ranks_2p20", "simulated_4_ranks_2p20", "simulated_8_ranks_sponge_blowup4", "u32_2p20"):
    try:
        d = json.load(open("gpurun_out/${TAG}_bench_%s.json" % f))
    except Exception as e:
        print(f, "no line", e); continue
    s = d.get("simulated_multi_gpu", {})
    print(f, d["ms_per_step"], d["value"], s.get("ranks"), s.get("slowest_rank_sum_ms"), s.get("projected_ms_per_proof"), s.get("bytes_sent_per_rank"), s.get("same_proof_as_single_gpu"), s.get("error"), d.get("verified", {}).get("accepted"))
P
