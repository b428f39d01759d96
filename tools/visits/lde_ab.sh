#!/bin/bash
# A/B timings of the main-table LDE (tools/probe.py: HIP events around tvm_lde_table, 379 columns at 2^20 rows, 3 repetitions)
mkdir -p gpurun_out
OUT=gpurun_out/${1:-ab}_lde_ab.txt
: > $OUT
run() { echo "== $*" >> $OUT; env "$@" python tools/probe.py 20 379 0 3 2>&1 | grep lde_ms | sed 's/.*"rep": \([0-9]\), "lde_ms": \([0-9.]*\).*/rep \1 lde_ms \2/' >> $OUT; }
run X=1
run TVM_LDE_PASS2_TILE=16
run TVM_LDE_PASS1_TILE=16
run TVM_LDE_PASS2_TILE=16 TVM_LDE_PASS1_TILE=16
run TVM_LDE_ROWS=0
cat $OUT
