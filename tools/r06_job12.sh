# round 6, job 11: the last round of a permutation whose rate words the next absorb overwrites, without their recombination chains
# (-DTVM_TIP5_LEAN_ROUND=1, variant "lean") against the shipped kernel, one box
export TMPDIR=/tmp
T=r06_o
mkdir -p gpurun_out
cp triton_vm_amd/libtriton_hip.so /tmp/default.so
for V in default split default2 split2; do
  case $V in split*) cp triton_vm_amd/libtriton_hip_split.so triton_vm_amd/libtriton_hip.so;; *) cp /tmp/default.so triton_vm_amd/libtriton_hip.so;; esac
  ( timeout 600 python -m pytest tests/test_kernels_hash.py tests/test_proof_snapshot.py -m gpu -x -q 2>&1 | tail -2 ) > gpurun_out/${T}_${V}_pytest.log
  for i in 1 2; do timeout 300 tools/ubench/bin/tip5_floor 20 | grep "^product:"; done > gpurun_out/${T}_${V}_tip5_floor.txt
  ( timeout 600 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_${V}.err | tail -1 ) > gpurun_out/${T}_bench_2p20_${V}.json
done
cp /tmp/default.so triton_vm_amd/libtriton_hip.so
for V in default split default2 split2; do tail -1 gpurun_out/${T}_${V}_pytest.log; cat gpurun_out/${T}_${V}_tip5_floor.txt | cut -c1-140; python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p20_${V}.json')); s=d['stage_ms_cpp_host']; print('$V', d['ms_per_step'], s['main Merkle'], s['aux Merkle'], s['quotient Merkle'], d['verified']['accepted'])"; done
