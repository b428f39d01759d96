#!/bin/bash
# What do the wavefronts of a kernel wait on?  rocprofv3 PMC passes that split SQ_WAIT_ANY / SQ_WAIT_INST_ANY further: instruction
# fetch (SQ_IFETCH*, SQC_ICACHE_*), memory-instruction levels (SQ_INST_LEVEL_*), scratch (FLAT) traffic, scalar activity.
# Only counters the installed rocprofv3 lists (`rocprofv3 -L`) are requested; one pass per group (--kernel-trace + --pmc only).
# usage: bash tools/pmc_wait_split.sh <tag> <command ...>      summary -> gpurun_out/<tag>_summary.txt
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
timeout 120 rocprofv3 -L > $ROOT/gpurun_out/${TAG}_counter_list.txt 2>&1
have() { grep -qw "$1" $ROOT/gpurun_out/${TAG}_counter_list.txt; }
CMD=("$@")
run() {
  local name=$1; shift
  local ok=()
  for c in "$@"; do have $c && ok+=($c); done
  [ ${#ok[@]} -eq 0 ] && return
  echo "pass $name: ${ok[*]}" >> $ROOT/gpurun_out/${TAG}_passes.txt
  timeout 600 rocprofv3 --kernel-trace --pmc "${ok[@]}" --output-format csv -d $ROOT/gpurun_out/${TAG}_$name -o r -- "${CMD[@]}" > $ROOT/gpurun_out/${TAG}_$name.log 2>&1
}
: > $ROOT/gpurun_out/${TAG}_passes.txt
run a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_IFETCH SQ_IFETCH_LEVEL
run b SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_INSTS_SALU
run c SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES
run d SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run e SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_EXP_GDS SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS
cd $ROOT
python tools/pmc_summary.py gpurun_out/${TAG}_a gpurun_out/${TAG}_b gpurun_out/${TAG}_c gpurun_out/${TAG}_d gpurun_out/${TAG}_e > gpurun_out/${TAG}_summary.txt 2>&1
cat gpurun_out/${TAG}_passes.txt >> gpurun_out/${TAG}_summary.txt
grep -i "SQ_\|SQC_" gpurun_out/${TAG}_counter_list.txt | tr -s ' ' | cut -c1-200 | sort -u | head -400 > gpurun_out/${TAG}_sq_counters_available.txt
rm -rf gpurun_out/${TAG}_[a-e] gpurun_out/${TAG}_[a-e].log gpurun_out/${TAG}_counter_list.txt
