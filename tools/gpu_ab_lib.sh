#!/bin/bash
# A/B of two builds of libtriton_hip.so on ONE box: parity tests and bench + kernel trace on the default library, then the same
# bench + trace with triton_vm_amd/libtriton_hip_<variant>.so copied over it (the box's copy of the tree is scratch).
# usage: bash tools/gpu_ab_lib.sh <tag> <variant>
TAG=${1:-ab}; VAR=$2
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_kernels_field.py tests/test_kernels_ntt.py tests/test_kernels_hash.py tests/test_proof_snapshot.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/${TAG}_pytest_gpu.log
one() {
  ( timeout 600 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>gpurun_out/${TAG}_$1_bench.err | tail -1 ) > gpurun_out/${TAG}_$1_bench.json
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras 2>&1 | tail -3 ) > gpurun_out/${TAG}_$1_rocprof.log
  DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_$1_kernels.txt
  rm -rf gpurun_out/${TAG}_prof
}
one default
if [ -n "$VAR" ]; then
  cp triton_vm_amd/libtriton_hip_${VAR}.so triton_vm_amd/libtriton_hip.so
  one $VAR
fi
cat gpurun_out/${TAG}_pytest_gpu.log
python - <<P
import json,glob
for f in sorted(glob.glob("gpurun_out/${TAG}_*_bench.json")):
    try:
        d=json.load(open(f))
        print(f, d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d.get("verified"), json.dumps(d["stage_ms"]))
    except Exception as e:
        print(f, "unreadable", e)
P
for f in gpurun_out/${TAG}_*_kernels.txt; do echo $f; head -28 $f | cut -c1-118; done
