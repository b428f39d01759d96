#!/bin/bash
# Round-4 GPU visit F: LDE passes 2 / 3 with one 2048-point row per wavefront (k_lde_pass{2,3}_rows<11, 8>) against the tile kernels
# (TVM_LDE_ROWS11=0) at 2^21 and 2^22 rows; parity of the LDE kernels.
TAG=${1:-visit}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_kernels_ntt.py "tests/test_gpu_fullsize.py" -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/${TAG}_pytest_gpu.log
for LOG in 21 22; do
  for ROWS11 in 0 1; do
    ( TVM_LDE_ROWS11=$ROWS11 timeout 600 python bench.py --log2-rows $LOG --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>gpurun_out/${TAG}_2p${LOG}_rows11_$ROWS11.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_2p${LOG}_rows11_$ROWS11.json
  done
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/bench.py --log2-rows 22 --steps 1 --warmup 0 --no-cpu-baseline --no-extras 2>&1 | tail -2 ) > gpurun_out/${TAG}_rocprof.log
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_bench_2p22_kernels.txt
rm -rf gpurun_out/${TAG}_prof
cat gpurun_out/${TAG}_pytest_gpu.log
python - <<P
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d = json.load(open(f))
        print(f, d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d.get("verified", {}).get("accepted"), d["stage_ms"]["main LDE"], d["stage_ms"]["aux LDE"])
    except Exception as e:
        print(f, "unreadable:", e)
P
head -12 gpurun_out/${TAG}_bench_2p22_kernels.txt | cut -c1-150
for f in gpurun_out/${TAG}_2p*.err; do echo $f; tail -2 $f; done
