#!/bin/bash
# Round-5 GPU visit P: pass 3 on 2048-point rows as two 1024-point halves (k_lde_pass3_halves): parity on everything that runs at 2^22 /
# 2^23 rows, the proof at 2^22 rows, the default-shape proof.
TAG=${1:-r05_p}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_kernels_ntt.py tests/test_gpu_fullsize.py tests/test_gpu_baseline_configs.py tests/test_proof_snapshot.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5 ) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log
for LOG in 22 20; do
  ( timeout 900 python bench.py --log2-rows $LOG --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>gpurun_out/${TAG}_bench$LOG.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_2p$LOG.json
done
python - <<P
import json
for L in (22, 20):
    d = json.load(open(f"gpurun_out/${TAG}_bench_2p{L}.json"))
    print(L, d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d.get("verified", {}).get("accepted"))
    print(json.dumps(d.get("stage_ms")))
P
