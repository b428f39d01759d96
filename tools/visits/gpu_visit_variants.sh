#!/bin/bash
# time the LDE kernels of one 96-column chunk at 2^20 rows under variant libraries: bash tools/gpu_visit_variants.sh <tag> <variant> ...
# ("-" = the product library); per-kernel rocprofv3 --stats lines -> gpurun_out/<tag>_variants.txt
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > gpurun_out/${TAG}_variants.txt
for V in "$@"; do
  ( cd /tmp && TVM_LIB_VARIANT=${V#-} timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_$V -o p -- python $R/tools/probe.py 20 96 0 5 2>&1 | tail -2 ) > gpurun_out/${TAG}_rocprof_$V.log
  DB=$(find gpurun_out/${TAG}_prof_$V -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB | grep "k_lde\|k_hash" | sed "s/^/$V  /" | cut -c1-175 >> gpurun_out/${TAG}_variants.txt
  rm -rf gpurun_out/${TAG}_prof_$V gpurun_out/${TAG}_rocprof_$V.log
done
cat gpurun_out/${TAG}_variants.txt
