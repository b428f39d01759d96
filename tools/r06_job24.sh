# round 6, job 24: same-box alternation of the library of the closing visit before (4a284f5: libtriton_hip_old.so) and the final one, 2^20 rows
export TMPDIR=/tmp
T=r06_ab
mkdir -p gpurun_out
cp triton_vm_amd/libtriton_hip.so /tmp/new.so
for ROUND in 1 2; do for V in new old; do
  [ $V = old ] && cp triton_vm_amd/libtriton_hip_old.so triton_vm_amd/libtriton_hip.so || cp /tmp/new.so triton_vm_amd/libtriton_hip.so
  ( timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 ) > gpurun_out/${T}_${V}_$ROUND.json
  python -c "
import json; d=json.load(open('gpurun_out/${T}_${V}_$ROUND.json')); s=d['stage_ms_cpp_host']; print('$V', $ROUND, d['ms_per_step'], s['main Merkle'], s['AIR quotients'], s['main LDE'], s['FRI'])"
done; done | tee gpurun_out/${T}_final_against_4a284f5_same_box.txt
cp /tmp/new.so triton_vm_amd/libtriton_hip.so
