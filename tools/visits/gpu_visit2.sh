#!/bin/bash
# GPU visit for kernel work: parity suite (without the pure-Python STIR verifier), bench, an experiment knob, kernel trace.
# usage: bash tools/gpu_visit2.sh <tag> [env assignments for the experiment run, e.g. TVM_LDE_PASS3_ROWS=4]
TAG=${1:-visit}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
( TVM_SKIP_SLOW_VERIFIER=1 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${TAG}_pytest_gpu.log
( timeout 600 python bench.py --steps 5 --warmup 2 2>gpurun_out/${TAG}_bench.err | tail -1 ) > gpurun_out/${TAG}_bench.json
for E in "$@"; do
  ( env $E timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline 2>&1 | tail -1 ) > gpurun_out/${TAG}_bench_${E}.json
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras 2>&1 | tail -3 ) > gpurun_out/${TAG}_rocprof.log
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_kernels.txt
rm -rf gpurun_out/${TAG}_prof
cat gpurun_out/${TAG}_pytest_gpu.log
python - <<P
import json,glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench*.json")):
    try:
        d=json.load(open(f))
        print(f, d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"], json.dumps(d["stage_ms"]))
    except Exception as e:
        print(f, "unreadable", e)
P
tail -3 gpurun_out/${TAG}_bench.err
head -24 gpurun_out/${TAG}_kernels.txt
