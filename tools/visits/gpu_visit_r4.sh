#!/bin/bash
# Round-4 GPU visit: parity of what changed, the default bench (with extras), the sharded code path over RCCL with one rank, PMC
# counters of the shipped hot kernels, a kernel trace of one proof.   usage: bash tools/gpu_visit_r4.sh <tag> [pytest targets...]
TAG=${1:-visit}; shift
TARGETS=${@:-tests/test_sharded_host.py tests/test_native_host.py tests/test_proof_snapshot.py tests/test_kernels_poly.py tests/test_bench_distributed.py}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest $TARGETS -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${TAG}_pytest_gpu.log
( timeout 900 python bench.py 2>gpurun_out/${TAG}_bench.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench.json
( timeout 600 python bench.py --sharded --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>gpurun_out/${TAG}_sharded.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_sharded_1rank_rccl.json
bash tools/pmc.sh ${TAG}_pmc python $R/tools/probe.py 20 96 0 1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras 2>&1 | tail -3 ) > gpurun_out/${TAG}_rocprof.log
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_kernels.txt
rm -rf gpurun_out/${TAG}_prof
cat gpurun_out/${TAG}_pytest_gpu.log
python - <<P
import json
for name in ("bench", "bench_sharded_1rank_rccl"):
    try:
        d = json.load(open("gpurun_out/${TAG}_%s.json" % name))
        print(name, d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"].get("bound"), d.get("verified", {}).get("accepted"))
        print("  stage_ms", json.dumps(d["stage_ms"]))
        for k in ("exact_air_real", "reference_default_ldt", "pcie_inclusive", "synthetic_hot_path"):
            if k in d: print("  ", k, d[k]["ms_per_step"])
        if "simulated_multi_gpu" in d: print("  simulated", json.dumps({k: v for k, v in d["simulated_multi_gpu"].items() if k not in ("exchanges_of_rank_0", "note")})[:3000])
        if "cpu_baseline" in d: print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["estimated_prove_seconds"], json.dumps(d["cpu_baseline"]["rates"]))
    except Exception as e:
        print(name, "unreadable:", e)
P
tail -5 gpurun_out/${TAG}_bench.err gpurun_out/${TAG}_sharded.err
head -30 gpurun_out/${TAG}_kernels.txt | cut -c1-150
grep -A20 "k_hash_rows_mfma\|k_lde_pass2" gpurun_out/${TAG}_pmc_summary.txt | head -70
