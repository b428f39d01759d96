#!/bin/bash
# Round-5 GPU visit C: what bounds pass 2 of the LDE?  Timing-only variants of k_lde_pass2_fused (wrong results by construction):
# no store phase / the same words stored as contiguous 64 KB blocks / no workgroup barriers -- against the product kernel.
TAG=${1:-r05_c}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in - nostore linstore nobarrier; do
  ( cd /tmp && TVM_LIB_VARIANT=${V#-} timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_$V -o p -- python $R/tools/probe.py 20 96 0 4 2>&1 | tail -2 ) > gpurun_out/${TAG}_rocprof_$V.log
  DB=$(find gpurun_out/${TAG}_prof_$V -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB | grep "k_lde" | sed "s/^/$V  /" | cut -c1-170 >> gpurun_out/${TAG}_pass2_variants.txt
  rm -rf gpurun_out/${TAG}_prof_$V
done
cat gpurun_out/${TAG}_pass2_variants.txt
