#!/bin/bash
# round 4, visit U: single-coset tables walked in storage order (eight ranks), tiled table linear combination: parity tests, the default
# bench with its 8-rank lockstep simulation
TAG=${1:-r04_u}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_poly.py tests/test_kernels_hash.py tests/test_sharded_host.py tests/test_bench_distributed.py tests/test_proof_snapshot.py -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1
tail -3 gpurun_out/${TAG}_pytest.log
timeout 900 python bench.py --steps 3 --warmup 1 2>gpurun_out/${TAG}_bench.err | grep '^{' | tail -1 > gpurun_out/${TAG}_bench_2p20.json
python - <<P
import json
d=json.load(open("gpurun_out/${TAG}_bench_2p20.json"))
print(d["ms_per_step"], d["value"], d["verified"]["accepted"], d["stage_ms"])
s=d["simulated_multi_gpu"]
print(s["ranks"], s["slowest_rank_sum_ms"], s.get("projected_ms_per_step"), {k:v[0] for k,v in s["stage_ms_per_rank"].items()})
print(d["reference_default_ldt"]["ms_per_step"])
P
