#!/bin/bash
# Round-4 GPU visit H: pass 2 of the LDE, store-phase factors from a table with 4 / 8 / 16 loads in flight (variant libraries), against the running products
TAG=${1:-visit}
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
for V in base tab4 tab8 tab16; do
  case $V in
    base) ( TVM_LDE_STORE_TABLE=0 timeout 300 python tools/probe.py 20 379 0 4 2>&1 | grep lde_ms | tail -2 ) >> gpurun_out/${TAG}_probe_$V.log ;;
    tab4) ( TVM_LDE_STORE_TABLE=1 timeout 300 python tools/probe.py 20 379 0 4 2>&1 | grep lde_ms | tail -2 ) >> gpurun_out/${TAG}_probe_$V.log ;;
    *) ( TVM_LDE_STORE_TABLE=1 TVM_LIB_VARIANT=$V timeout 300 python tools/probe.py 20 379 0 4 2>&1 | grep lde_ms | tail -2 ) >> gpurun_out/${TAG}_probe_$V.log ;;
  esac
done
done
for V in base tab4 tab8 tab16; do echo "--- $V"; cut -c1-120 gpurun_out/${TAG}_probe_$V.log; done
