#!/bin/bash
# Round-5 GPU visit L: pass 1 on 2048-point axes (k_lde_pass1_rows<11, 8>), the paired stores of pass 2: parity, kernel times at 2^22 /
# 2^20 rows (TVM_OPTION_LDE_PASS2_TILES = 1: the kernels they replace), the proofs at 2^22 and 2^20 rows.
TAG=${1:-r05_l}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests/test_kernels_ntt.py tests/test_gpu_fullsize.py tests/test_proof_snapshot.py tests/test_sharded_host.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 ) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log
: > gpurun_out/${TAG}_lde_kernels.txt
for LOG in 22 20; do for TILES in 0 1; do
  ( cd /tmp && TVM_PROBE_OPTIONS="4=$TILES" timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o p -- python $R/tools/probe.py $LOG 96 0 4 2>&1 | tail -2 ) > gpurun_out/${TAG}_rocprof.log
  DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB | grep "k_lde\|k_ntt2" | sed "s/^/2^$LOG tiles=$TILES  /" | cut -c1-185 >> gpurun_out/${TAG}_lde_kernels.txt
  rm -rf gpurun_out/${TAG}_prof
done; done
cat gpurun_out/${TAG}_lde_kernels.txt
for LOG in 22 20; do
  ( timeout 900 python bench.py --log2-rows $LOG --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>gpurun_out/${TAG}_bench$LOG.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_2p$LOG.json
done
python - <<P
import json
for L in (22, 20):
    d = json.load(open(f"gpurun_out/${TAG}_bench_2p{L}.json"))
    print(L, d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"].get("traffic"), d.get("verified", {}).get("accepted"))
    print(json.dumps(d.get("stage_ms")))
P
