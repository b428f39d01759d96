#!/bin/bash
# Round-4 GPU visit AA: second end-of-round validation: the whole GPU suite, smoke(), the default bench (with all its extras)
TAG=${1:-r04_aa}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | tail -8 ) > gpurun_out/${TAG}_pytest_gpu.log
cat gpurun_out/${TAG}_pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > gpurun_out/${TAG}_smoke.log
cat gpurun_out/${TAG}_smoke.log
( timeout 900 python bench.py 2>gpurun_out/${TAG}_bench.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_2p20.json
python - <<P
import json
d=json.load(open("gpurun_out/${TAG}_bench_2p20.json"))
print(d["ms_per_step"], d["value"], d["verified"]["accepted"], d["roofline"]["frac"], d["cpu_baseline"]["estimated_prove_seconds"])
print(d["stage_ms"])
s=d["simulated_multi_gpu"]; print(s["ranks"], s["slowest_rank_sum_ms"], s["projected_ms_per_proof"], s["same_proof_as_single_gpu"])
print("stir", d["reference_default_ldt"]["ms_per_step"], "exact", d["exact_air_real"]["ms_per_step"], "pcie", d["pcie_inclusive"]["ms_per_step"], "synthetic", d["synthetic_hot_path"]["ms_per_step"])
P
