# round 6, job 25: downloads through the pinned ring (d2h_small) against the plain path (libtriton_hip_old.so = the closing visit's library), same box
export TMPDIR=/tmp
T=r06_d2h
mkdir -p gpurun_out
cp triton_vm_amd/libtriton_hip.so /tmp/new.so
for ROUND in 1 2; do for V in new old; do
  [ $V = old ] && cp triton_vm_amd/libtriton_hip_old.so triton_vm_amd/libtriton_hip.so || cp /tmp/new.so triton_vm_amd/libtriton_hip.so
  for L in 10 12 14; do
  ( timeout 600 python bench.py --log2-rows $L --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 ) > gpurun_out/${T}_${V}_${ROUND}_2p$L.json
  python -c "
import json; d=json.load(open('gpurun_out/${T}_${V}_${ROUND}_2p$L.json')); s=d['stage_ms_cpp_host']; print('$V', $ROUND, $L, d['ms_per_step'], s['open trace leafs'], s['out-of-domain rows'], s['FRI'])"
  done
  ( timeout 600 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 ) > gpurun_out/${T}_${V}_${ROUND}_2p20.json
  python -c "
import json; d=json.load(open('gpurun_out/${T}_${V}_${ROUND}_2p20.json')); s=d['stage_ms_cpp_host']; print('$V', $ROUND, 20, d['ms_per_step'], s['open trace leafs'], s['out-of-domain rows'], s['FRI'])"
done; done | tee gpurun_out/${T}_pinned_downloads_same_box.txt
cp /tmp/new.so triton_vm_amd/libtriton_hip.so
