#!/bin/bash
# Round-5 GPU visit I: pass 2 on 2048-point axes (2^21 / 2^22 rows): k_lde_pass2_fused<11> against the tile kernel k_lde_pass2_v3<11, 10>
# (TVM_OPTION_LDE_PASS2_TILES = 1): parity, per-kernel times of one 96-column chunk, the whole proof at 2^22 and 2^21 rows.
TAG=${1:-r05_i}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_kernels_ntt.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 ) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log
: > gpurun_out/${TAG}_lde_kernels.txt
for LOG in 22 21; do for TILES in 0 1; do
  ( cd /tmp && TVM_PROBE_OPTIONS="4=$TILES" timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o p -- python $R/tools/probe.py $LOG 96 0 3 2>&1 | tail -2 ) > gpurun_out/${TAG}_rocprof.log
  DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB | grep "k_lde" | sed "s/^/2^$LOG tiles=$TILES  /" | cut -c1-185 >> gpurun_out/${TAG}_lde_kernels.txt
  rm -rf gpurun_out/${TAG}_prof
done; done
cat gpurun_out/${TAG}_lde_kernels.txt
for LOG in 22 21; do
  ( timeout 900 python bench.py --log2-rows $LOG --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>gpurun_out/${TAG}_bench$LOG.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_2p$LOG.json
done
python - <<P
import json
for L in (22, 21):
    d = json.load(open(f"gpurun_out/${TAG}_bench_2p{L}.json"))
    print(L, d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d.get("verified", {}).get("accepted"))
    print(json.dumps(d.get("stage_ms")))
P
