#!/bin/bash
# Full GPU visit: the parity suite (without the pure-Python STIR verifier unless FULL=1), smoke, bench, kernel trace, and the
# PMC passes over the main-table LDE (SQ activity, FETCH_SIZE, WRITE_SIZE in separate runs) -> profiles/lde_traffic.json input.
# usage: bash tools/gpu_visit4.sh <tag>
TAG=${1:-visit}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ "${FULL:-0}" = "1" ]; then SKIP=0; else SKIP=1; fi
( TVM_SKIP_SLOW_VERIFIER=$SKIP timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25 ) > gpurun_out/${TAG}_pytest_gpu.log
( timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 ) > gpurun_out/${TAG}_smoke.log
( timeout 600 python bench.py --steps 10 --warmup 3 2>gpurun_out/${TAG}_bench.err | tail -1 ) > gpurun_out/${TAG}_bench.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras 2>&1 | tail -3 ) > gpurun_out/${TAG}_rocprof.log
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_kernels.txt
rm -rf gpurun_out/${TAG}_prof
bash tools/pmc.sh ${TAG}_pmc_lde python $R/tools/probe.py 20 96 0 1 > /dev/null 2>&1
cat gpurun_out/${TAG}_pytest_gpu.log gpurun_out/${TAG}_smoke.log
python - <<P
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print(d["ms_per_step"], d["value"], json.dumps(d["roofline"]), json.dumps(d["stage_ms"]))
for k in ("pcie_inclusive","reference_default_ldt","synthetic_hot_path","cpu_baseline","verified"):
    print(k, json.dumps(d.get(k))[:400])
P
tail -3 gpurun_out/${TAG}_bench.err
head -16 gpurun_out/${TAG}_kernels.txt
grep -A20 "k_lde_pass[123]_rows" gpurun_out/${TAG}_pmc_lde_summary.txt | head -50
