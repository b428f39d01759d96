#!/bin/bash
# Round-5 GPU visit H: the pass-2 kernel with LDS-staged coset factors: parity (kernels, full size, sharded, snapshots), the default bench,
# counters of the shipped hot kernels (-> profiles/kernel_counters.json), a kernel trace of one proof.
TAG=${1:-r05_h}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests/test_kernels_ntt.py tests/test_gpu_fullsize.py tests/test_sharded_host.py tests/test_proof_snapshot.py tests/test_prover_pipeline.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5 ) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log
( timeout 900 python bench.py --steps 10 --warmup 3 2>gpurun_out/${TAG}_bench.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_2p20.json
bash tools/pmc.sh ${TAG}_pmc python $R/tools/probe.py 20 96 0 1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -3 ) > gpurun_out/${TAG}_rocprof.log
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_bench_2p20_kernels.txt
rm -rf gpurun_out/${TAG}_prof
python - <<P
import json
d = json.load(open("gpurun_out/${TAG}_bench_2p20.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d.get("verified", {}).get("accepted"))
for k in ("exact_air_real", "reference_default_ldt", "pcie_inclusive", "synthetic_hot_path"):
    if k in d: print("  ", k, d[k]["ms_per_step"])
s = d.get("simulated_multi_gpu", {})
print("   sim", s.get("ranks"), s.get("slowest_rank_sum_ms"), s.get("projected_ms_per_proof"), s.get("same_proof_as_single_gpu"), s.get("error"))
print(json.dumps(d.get("stage_ms")))
P
head -14 gpurun_out/${TAG}_bench_2p20_kernels.txt | cut -c1-160
grep -A20 "k_lde_pass2_fused" gpurun_out/${TAG}_pmc_summary.txt | head -22
