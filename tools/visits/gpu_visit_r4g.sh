#!/bin/bash
# Round-4 GPU visit G: pass 2 of the LDE with the store-phase factors from a table (TVM_LDE_STORE_TABLE=1) against the running products
TAG=${1:-visit}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for T in 0 1 0 1; do
  ( TVM_LDE_STORE_TABLE=$T timeout 300 python tools/probe.py 20 379 91 4 2>&1 | grep -i "lde" | tail -8 ) >> gpurun_out/${TAG}_probe_table_$T.log
done
( cd /tmp && TVM_LDE_STORE_TABLE=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/tools/probe.py 20 379 91 3 2>&1 | tail -2 ) > gpurun_out/${TAG}_rocprof.log
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_probe_table_kernels.txt
rm -rf gpurun_out/${TAG}_prof
( TVM_LDE_STORE_TABLE=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>gpurun_out/${TAG}_t1.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_table_1.json
( TVM_LDE_STORE_TABLE=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>gpurun_out/${TAG}_t0.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_table_0.json
echo "--- table 0"; cat gpurun_out/${TAG}_probe_table_0.log
echo "--- table 1"; cat gpurun_out/${TAG}_probe_table_1.log
head -8 gpurun_out/${TAG}_probe_table_kernels.txt | cut -c1-150
python - <<P
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d = json.load(open(f))
        print(f, d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d.get("verified", {}).get("accepted"), d["stage_ms"]["main LDE"], d["stage_ms"]["aux LDE"])
    except Exception as e:
        print(f, "unreadable:", e)
P
