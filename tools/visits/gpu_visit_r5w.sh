#!/bin/bash
# Round 5: the kernel trace of the 2^22-row proof (configs[2]'s height on one GPU) on the final tree.
TAG=${1:-r05_w}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 800 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/bench.py --log2-rows 22 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_2p22_under_rocprof.json
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_bench_2p22_kernels.txt
[ -n "$DB" ] && python tools/rocprof_gaps.py $DB > gpurun_out/${TAG}_device_idle_gaps_2p22.txt 2>&1
rm -rf gpurun_out/${TAG}_prof
head -14 gpurun_out/${TAG}_bench_2p22_kernels.txt | cut -c1-170
grep "^segment" gpurun_out/${TAG}_device_idle_gaps_2p22.txt
