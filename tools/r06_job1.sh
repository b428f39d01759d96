export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python bench.py --steps 10 --warmup 2 2>gpurun_out/r06_a_bench_2p20.err | tail -1 ) > gpurun_out/r06_a_bench_2p20.json
for L in 16 18; do
  ( timeout 600 python bench.py --log2-rows $L --ldt auto --steps 10 --warmup 2 --no-cpu-baseline 2>gpurun_out/r06_a_bench_2p$L.err | tail -1 ) > gpurun_out/r06_a_bench_2p$L.json
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$L -o bench -- python $R/bench.py --log2-rows $L --ldt auto --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -3 ) > gpurun_out/r06_a_rocprof_2p$L.log
  DB=$(find gpurun_out/prof_$L -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/r06_a_bench_2p${L}_kernels.txt
  rm -rf gpurun_out/prof_$L
done
tail -c 600 gpurun_out/r06_a_bench_2p20.json
head -30 gpurun_out/r06_a_bench_2p16_kernels.txt | cut -c1-150
head -30 gpurun_out/r06_a_bench_2p18_kernels.txt | cut -c1-150
