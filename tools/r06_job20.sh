# round 6, job 20: merged kernel / host-API timelines of one proof: 2^10 rows (FRI) and 2^16 rows (STIR)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=r06_v
mkdir -p gpurun_out
for CFG in "10 fri" "16 stir"; do set -- $CFG; L=$1; LDT=$2
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --hip-trace -d $R/gpurun_out/prof_h$L -o bench -- python $R/bench.py --log2-rows $L --ldt $LDT --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -2 ) > gpurun_out/${T}_rocprof_hip_2p$L.log
DB=$(find gpurun_out/prof_h$L -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_host_timeline.py $DB 25 gpurun_out/${T}_timeline_2p${L}_$LDT.txt > gpurun_out/${T}_host_timeline_2p${L}_$LDT.txt 2>&1
rm -rf gpurun_out/prof_h$L
done
wc -l gpurun_out/${T}_timeline_*.txt
