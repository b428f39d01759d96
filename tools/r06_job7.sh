# round 6, job 7: pair synchronisation of the two wavefronts of a 2048-point row (k_lde_pass2_fused<11>, k_lde_pass1_rows<11, 8>) against
# the workgroup barrier (variant nops), 2^21 / 2^22 rows, one box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=r06_i
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_ntt.py -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/${T}_pytest_gpu.log
cat gpurun_out/${T}_pytest_gpu.log
for V in pair_sync nops; do
  [ $V = nops ] && cp triton_vm_amd/libtriton_hip_nops.so triton_vm_amd/libtriton_hip.so
  for L in 22; do
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$V -o bench -- python $R/bench.py --log2-rows $L --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -2 ) > gpurun_out/${T}_rocprof_${V}_2p$L.log
    DB=$(find gpurun_out/prof_$V -name '*.db' | head -1)
    [ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${T}_bench_2p${L}_${V}_kernels.txt
    rm -rf gpurun_out/prof_$V
    ( timeout 900 python bench.py --log2-rows $L --steps 3 --warmup 1 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_${V}_2p$L.err | tail -1 ) > gpurun_out/${T}_bench_2p${L}_${V}.json
  done
done
for V in pair_sync nops; do echo $V; grep "k_lde_pass" gpurun_out/${T}_bench_2p22_${V}_kernels.txt | cut -c1-150; python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p22_${V}.json')); print(d['ms_per_step'], d['value'], d.get('verified',{}).get('accepted'))"; done
