#!/bin/bash
# rocprofv3 PMC passes (SQ activity, then HBM bytes; separate runs, --kernel-trace only) over a command.
# usage: bash tools/pmc.sh <tag> <command ...>      summary -> gpurun_out/<tag>_summary.txt
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp  # (rocprofv3 wants a writable cwd; commands must use absolute paths: $GRAFT_REPO_ROOT/...)
run() {  # name, counters... (the command is in "${CMD[@]}")
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/gpurun_out/${TAG}_$name -o r -- "${CMD[@]}" > $ROOT/gpurun_out/${TAG}_$name.log 2>&1
}
CMD=("$@")
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $ROOT
python tools/pmc_summary.py gpurun_out/${TAG}_sq gpurun_out/${TAG}_sq2 gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write > gpurun_out/${TAG}_summary.txt 2>&1
rm -rf gpurun_out/${TAG}_sq gpurun_out/${TAG}_sq2 gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_*.log
