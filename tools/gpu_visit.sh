#!/bin/bash
# One GPU visit: parity suite, smoke, bench (2^12 quick + 2^20 full), rocprof kernel trace of one bench pass.
# usage (from the repo root on the GPU box):  bash tools/gpu_visit.sh <tag> [pytest-args]
TAG=${1:-visit}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q ${2:-} 2>&1 | tail -25 ) > gpurun_out/${TAG}_pytest_gpu.log
( timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 ) > gpurun_out/${TAG}_smoke.log
( timeout 300 python bench.py --log2-rows 12 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 ) > gpurun_out/${TAG}_bench_2p12.log
( timeout 900 python bench.py --steps 3 --warmup 1 2>&1 | tail -3 ) > gpurun_out/${TAG}_bench_2p20.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -3 ) > gpurun_out/${TAG}_rocprof.log
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_kernels.txt
# the trace database itself is large: keep only the summary
rm -rf gpurun_out/${TAG}_prof
cat gpurun_out/${TAG}_pytest_gpu.log gpurun_out/${TAG}_smoke.log gpurun_out/${TAG}_bench_2p12.log gpurun_out/${TAG}_bench_2p20.log
head -40 gpurun_out/${TAG}_kernels.txt
