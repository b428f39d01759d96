#!/bin/bash
# round 4, visit P: the f64 matrix form of the Tip5 MDS layer against the i8 form (libtriton_hip_i8.so), same box
tag=${1:-r04_p}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_hash.py -m gpu -x -q > gpurun_out/${tag}_pytest_hash.log 2>&1
tail -3 gpurun_out/${tag}_pytest_hash.log
for v in f64 i8; do
  if [ $v = i8 ]; then export TVM_LIB_VARIANT=i8; else unset TVM_LIB_VARIANT; fi
  timeout 600 python tools/probe.py 20 379 91 3 > gpurun_out/${tag}_probe_${v}.txt 2>&1
  tail -4 gpurun_out/${tag}_probe_${v}.txt
done
unset TVM_LIB_VARIANT
timeout 900 python bench.py --steps 3 --warmup 1 2>gpurun_out/${tag}_bench.err | grep '^{' | tail -1 > gpurun_out/${tag}_bench_2p20.json
python - <<P
import json
d=json.load(open("gpurun_out/${tag}_bench_2p20.json"))
print(d["ms_per_step"], d["value"], d["verified"]["accepted"], d["stage_ms"])
P
