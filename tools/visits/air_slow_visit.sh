#!/bin/bash
# DIAGNOSTIC visit: time the stages; when the AIR stage is slow on this box (> 100 ms at 2^20 rows), collect what could explain it.
TAG=${1:-probe}
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/variant_probe.py > gpurun_out/${TAG}_stages.log 2>&1
AIR=$(grep -o "'AIR quotients': [0-9.]*" gpurun_out/${TAG}_stages.log | grep -o "[0-9.]*$")
echo "AIR stage: $AIR ms" >> gpurun_out/${TAG}_stages.log
if python -c "import sys; sys.exit(0 if float('${AIR:-0}') > 100 else 1)"; then
  ( rocm-smi -a 2>&1 | head -150 ) > gpurun_out/${TAG}_smi.log
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -o run -- python $GRAFT_REPO_ROOT/tools/variant_probe.py 2>&1 | tail -3 ) > gpurun_out/${TAG}_rocprof.log
  DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_kernels.txt
  rm -rf gpurun_out/${TAG}_prof
  # again without the profiler: still slow?  and with the C++ host (bench)
  python tools/variant_probe.py 2>&1 | grep "AIR" >> gpurun_out/${TAG}_stages.log
  ( HSA_NO_SCRATCH_RECLAIM=1 python tools/variant_probe.py 2>&1 | grep "AIR" | sed 's/^/HSA_NO_SCRATCH_RECLAIM=1 /' ) >> gpurun_out/${TAG}_stages.log
  ( HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 python tools/variant_probe.py 2>&1 | grep "AIR" | sed 's/^/HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 /' ) >> gpurun_out/${TAG}_stages.log
  ( dmesg 2>/dev/null | tail -30 ) > gpurun_out/${TAG}_dmesg.log
fi
cat gpurun_out/${TAG}_stages.log | tail -6
