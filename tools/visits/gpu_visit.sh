#!/bin/bash
# One GPU visit: parity suite, smoke, bench (real prove_fib at 2^20), rocprof kernel trace of one bench pass.
# usage (from the repo root on the GPU box):  bash tools/gpu_visit.sh <tag> [pytest-args]
TAG=${1:-visit}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 ${2:-} 2>&1 | tail -45 ) > gpurun_out/${TAG}_pytest_gpu.log
( timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 ) > gpurun_out/${TAG}_smoke.log
( timeout 900 python bench.py --steps 5 --warmup 2 2>gpurun_out/${TAG}_bench_2p20.err | tail -1 ) > gpurun_out/${TAG}_bench_2p20.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras 2>&1 | tail -3 ) > gpurun_out/${TAG}_rocprof.log
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_kernels.txt
# the trace database itself is large: keep only the summary
rm -rf gpurun_out/${TAG}_prof
cat gpurun_out/${TAG}_pytest_gpu.log gpurun_out/${TAG}_smoke.log gpurun_out/${TAG}_bench_2p20.json
tail -5 gpurun_out/${TAG}_bench_2p20.err
head -30 gpurun_out/${TAG}_kernels.txt
