#!/bin/bash
# Round-4 GPU visit I (end of round): the default bench with all its extras, the lockstep measurements with 2 and 4 ranks, the sharded code
# path over RCCL with one rank, PMC counters of the shipped hot kernels, a kernel trace, smoke.
TAG=${1:-visit}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/${TAG}_smoke.log
( timeout 900 python bench.py 2>gpurun_out/${TAG}_bench.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench.json
for N in 2 4; do
  ( timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --simulate-gpus $N 2>gpurun_out/${TAG}_sim$N.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_simulated_${N}_ranks.json
done
( timeout 600 python bench.py --sharded --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>gpurun_out/${TAG}_sharded.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_sharded_1rank_rccl.json
bash tools/pmc.sh ${TAG}_pmc python $R/tools/probe.py 20 96 0 1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras 2>&1 | tail -3 ) > gpurun_out/${TAG}_rocprof.log
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_kernels.txt
rm -rf gpurun_out/${TAG}_prof
cat gpurun_out/${TAG}_smoke.log
python - <<P
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench*.json")):
    try:
        d = json.load(open(f))
        print(f, d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"].get("bound"), d.get("verified", {}).get("accepted"))
        for k in ("exact_air_real", "reference_default_ldt", "pcie_inclusive", "synthetic_hot_path"):
            if k in d: print("  ", k, d[k]["ms_per_step"])
        if "simulated_multi_gpu" in d:
            s = d["simulated_multi_gpu"]
            print("   sim", s.get("ranks"), s.get("slowest_rank_sum_ms"), s.get("projected_ms_per_proof"), s.get("bytes_sent_per_rank"), s.get("same_proof_as_single_gpu"), s.get("error"))
            print("   ", json.dumps({k: max(v) for k, v in s.get("stage_ms_per_rank", {}).items()}))
        if "cpu_baseline" in d: print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["estimated_prove_seconds"], json.dumps(d["cpu_baseline"]["rates"]), d["cpu_baseline"]["sample_seconds"])
    except Exception as e:
        print(f, "unreadable:", e)
P
head -16 gpurun_out/${TAG}_kernels.txt | cut -c1-150
