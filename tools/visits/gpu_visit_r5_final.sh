#!/bin/bash
# Round-5 closing GPU visit on the final tree: the whole GPU suite, smoke, the default bench with the driver's --steps 20 --warmup 5
# (extras, CPU baseline, 8-rank lockstep, the C++ host's stage times), counters of the shipped hot kernels at the default shape,
# a kernel trace and the idle gaps of one proof, the 2^22-row proof.
TAG=${1:-r05_final}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 2400 python -m pytest tests/ -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 ) > gpurun_out/${TAG}_pytest_gpu_full_suite.log
cat gpurun_out/${TAG}_pytest_gpu_full_suite.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/${TAG}_smoke.log
cat gpurun_out/${TAG}_smoke.log
( timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/${TAG}_bench.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_2p20.json
bash tools/pmc.sh ${TAG}_pmc python $R/tools/probe.py 20 96 0 1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -3 ) > gpurun_out/${TAG}_rocprof.log
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_bench_2p20_kernels.txt
[ -n "$DB" ] && python tools/rocprof_gaps.py $DB > gpurun_out/${TAG}_device_idle_gaps.txt 2>&1
rm -rf gpurun_out/${TAG}_prof
( timeout 900 python bench.py --log2-rows 22 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>gpurun_out/${TAG}_bench22.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_2p22.json
python - <<P
import json
for f in ("gpurun_out/${TAG}_bench_2p20.json", "gpurun_out/${TAG}_bench_2p22.json"):
    d = json.load(open(f))
    print(f, d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"].get("traffic_source_commit"), d.get("verified", {}).get("accepted"))
    for k in ("exact_air_real", "reference_default_ldt", "pcie_inclusive", "synthetic_hot_path"):
        if k in d: print("  ", k, d[k]["ms_per_step"])
    s = d.get("simulated_multi_gpu", {})
    if s: print("   sim", s.get("ranks"), s.get("slowest_rank_sum_ms"), s.get("projected_ms_per_proof"), s.get("same_proof_as_single_gpu"), s.get("error"))
    print(json.dumps(d.get("stage_ms")))
    print("cpp host:", json.dumps(d.get("stage_ms_cpp_host")))
    c = d.get("cpu_baseline", {})
    if c: print("cpu", c.get("value"), c.get("cores"), c.get("estimated_prove_seconds"))
P
head -3 gpurun_out/${TAG}_device_idle_gaps.txt
