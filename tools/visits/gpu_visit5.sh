#!/bin/bash
# Light GPU visit: kernel parity files + the full-size quotient segments + the proof snapshots, bench, kernel trace.
# usage: bash tools/gpu_visit5.sh <tag>
TAG=${1:-visit}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests/test_kernels_field.py tests/test_kernels_ntt.py tests/test_kernels_hash.py tests/test_kernels_poly.py tests/test_kernels_air.py tests/test_proof_snapshot.py tests/test_prover_pipeline.py tests/test_native_host.py "tests/test_gpu_fullsize.py::test_full_size_quotient_segments" -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/${TAG}_pytest_gpu.log
( timeout 600 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>gpurun_out/${TAG}_bench.err | tail -1 ) > gpurun_out/${TAG}_bench.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras 2>&1 | tail -3 ) > gpurun_out/${TAG}_rocprof.log
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_kernels.txt
rm -rf gpurun_out/${TAG}_prof
cat gpurun_out/${TAG}_pytest_gpu.log
python - <<P
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print(d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d.get("verified"), json.dumps(d["stage_ms"]))
P
tail -3 gpurun_out/${TAG}_bench.err
head -24 gpurun_out/${TAG}_kernels.txt | cut -c1-130
