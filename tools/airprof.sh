#!/bin/bash
# Per-kernel times of the AIR parts for one or more library variants (triton_vm_amd/build.py: build(variant=...)).
# usage (on the GPU box): bash tools/airprof.sh default [variant ...]
export TMPDIR=/tmp; cd /tmp
for v in "${@:-default}"; do
  rocprofv3 --kernel-trace --stats -d /tmp/ap_$v -o r -- python $GRAFT_REPO_ROOT/tools/air_bench.py $v 2>&1 | grep air_ms
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/ap_$v -name "*.db" | head -1) | grep "k_air" | awk '{print $1, $3, $4}'
done
