export TMPDIR=/tmp; ROOT=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQC?_[A-Z0-9_]*(ICACHE|IFETCH|INST_LEVEL|WAVE_DEP|INSTS_SMEM|DCACHE|IB|ISSUE|EXP|THREAD_CYCLES|VALU_MFMA)[A-Z0-9_]*)" | sort -u | head -60
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $ROOT/gpurun_out/ic -o r -- python $ROOT/tools/air_bench.py default > $ROOT/gpurun_out/ic.log 2>&1
tail -3 $ROOT/gpurun_out/ic.log
cd $ROOT; python tools/pmc_summary.py gpurun_out/ic 2>&1 | grep -A8 "k_air_part_[259]" | head -40
rm -rf gpurun_out/ic
