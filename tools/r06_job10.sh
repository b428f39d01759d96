export TMPDIR=/tmp
T=r06_m
mkdir -p gpurun_out
cp triton_vm_amd/libtriton_hip.so /tmp/default.so
for V in default air15; do
  [ $V = air15 ] && cp triton_vm_amd/libtriton_hip_air15.so triton_vm_amd/libtriton_hip.so
  ( timeout 600 python -m pytest tests/test_kernels_air.py tests/test_proof_snapshot.py -m gpu -x -q 2>&1 | tail -2 ) > gpurun_out/${T}_${V}_pytest.log
  ( timeout 600 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_${V}.err | tail -1 ) > gpurun_out/${T}_bench_2p20_${V}.json
done
cp /tmp/default.so triton_vm_amd/libtriton_hip.so
for V in default air15; do cat gpurun_out/${T}_${V}_pytest.log; python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p20_${V}.json')); print('$V', d['ms_per_step'], d['stage_ms_cpp_host']['AIR quotients'], d['verified']['accepted'])"; done
