# round 6, job 13: the heights nobody measured below 2^16 rows (BASELINE configs[0] is 2^10): bench lines, kernel trace, idle gaps
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=r06_p
mkdir -p gpurun_out
for L in 10 12 14; do
  ( timeout 600 python bench.py --log2-rows $L --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_2p$L.err | tail -1 ) > gpurun_out/${T}_bench_2p${L}.json
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$L -o bench -- python $R/bench.py --log2-rows $L --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -2 ) > gpurun_out/${T}_rocprof_2p$L.log
  DB=$(find gpurun_out/prof_$L -name '*.db' | head -1)
  if [ -n "$DB" ]; then
    python tools/rocprof_summary.py $DB > gpurun_out/${T}_bench_2p${L}_kernels.txt
    python tools/rocprof_gaps.py $DB > gpurun_out/${T}_bench_2p${L}_gaps.txt 2>&1
  fi
  rm -rf gpurun_out/prof_$L
done
for L in 10 12 14; do python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p${L}.json')); print($L, d['ms_per_step'], d['value'], d.get('verified',{}).get('accepted'), d.get('stage_ms_cpp_host'))"; head -30 gpurun_out/${T}_bench_2p${L}_kernels.txt | cut -c1-150; tail -15 gpurun_out/${T}_bench_2p${L}_gaps.txt; done
