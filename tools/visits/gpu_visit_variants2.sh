#!/bin/bash
# as gpu_visit_variants.sh, at a given log2 of rows: bash tools/gpu_visit_variants2.sh <tag> <log2 rows> <variant> ...
TAG=$1; LOG=$2; shift; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in "$@"; do
  ( cd /tmp && TVM_LIB_VARIANT=${V#-} timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_$V -o p -- python $R/tools/probe.py $LOG 96 0 4 2>&1 | tail -2 ) > gpurun_out/${TAG}_rocprof_$V.log
  DB=$(find gpurun_out/${TAG}_prof_$V -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB | grep "k_lde\|k_ntt2" | sed "s/^/2^$LOG $V  /" | cut -c1-185 >> gpurun_out/${TAG}_variants.txt
  rm -rf gpurun_out/${TAG}_prof_$V gpurun_out/${TAG}_rocprof_$V.log
done
cat gpurun_out/${TAG}_variants.txt
