#!/bin/bash
# Round-5 GPU visit B: the fused pass 2 of the LDE (k_lde_pass2_fused) against the position-major tile of rounds 3-4 (k_lde_pass2_rows,
# a temporary context option, removed with that kernel after this visit): parity at full size, per-kernel times of one 96-column chunk and of the whole main table, counters.
TAG=${1:-r05_b}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_kernels_ntt.py tests/test_gpu_fullsize.py tests/test_kernels_hash.py tests/test_kernels_poly.py tests/test_sharded_host.py tests/test_proof_snapshot.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 ) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log
for FORM in 0 1; do
  ( TVM_PROBE_OPTIONS="4=$FORM" timeout 300 python tools/probe.py 20 379 0 3 2>&1 | tail -3 ) > gpurun_out/${TAG}_probe_main_form$FORM.txt
  ( cd /tmp && TVM_PROBE_OPTIONS="4=$FORM" timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof$FORM -o p -- python $R/tools/probe.py 20 96 0 3 2>&1 | tail -2 ) > gpurun_out/${TAG}_rocprof$FORM.log
  DB=$(find gpurun_out/${TAG}_prof$FORM -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_kernels_form$FORM.txt
  rm -rf gpurun_out/${TAG}_prof$FORM
done
cat gpurun_out/${TAG}_probe_main_form0.txt gpurun_out/${TAG}_probe_main_form1.txt
grep -h "k_lde" gpurun_out/${TAG}_kernels_form0.txt gpurun_out/${TAG}_kernels_form1.txt | cut -c1-160
bash tools/pmc.sh ${TAG}_pmc python $R/tools/probe.py 20 96 0 1
grep -A20 "k_lde_pass2_fused" gpurun_out/${TAG}_pmc_summary.txt | head -24
( timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>gpurun_out/${TAG}_bench.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_2p20.json
python - <<P
import json
d = json.load(open("gpurun_out/${TAG}_bench_2p20.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d.get("verified", {}).get("accepted"))
print(json.dumps(d.get("stage_ms")))
P
