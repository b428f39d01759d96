#!/bin/bash
# Per-kernel times of one main-table LDE at 2^20 rows (rocprofv3 kernel trace).  usage (GPU box): bash tools/lde_kernel_times.sh [variant]
export TMPDIR=/tmp; cd /tmp
TVM_LIB_VARIANT=$1 rocprofv3 --kernel-trace --stats -d /tmp/lx_$1 -o r -- python $GRAFT_REPO_ROOT/tools/probe.py 20 379 0 2 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/lx_$1 -name "*.db" | head -1) | grep -E "k_lde_pass|k_ntt2" | cut -c1-150
