#!/bin/bash
# second GPU visit: full parity suite, bench with roofline + cpu_baseline, rocprof kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/pytest_gpu.log
( timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 ) > gpurun_out/smoke.log
( timeout 300 python bench.py --log2-rows 12 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 ) > gpurun_out/bench_2p12.log
( timeout 900 python bench.py --steps 3 --warmup 1 2>&1 | tail -3 ) > gpurun_out/bench_2p20.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -3 ) > gpurun_out/rocprof.log
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench_2p12.log gpurun_out/bench_2p20.log
