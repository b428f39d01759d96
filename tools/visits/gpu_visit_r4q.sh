#!/bin/bash
# round 4, visit Q: STIR after three changes (interpolation kernel takes base-field points first; the answer polynomial is evaluated
# on the work coset by a transform; stacked leaves hashed in the matrix-core form): parity tests, bench both tests on one box, trace
TAG=${1:-r04_q}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_stir.py tests/test_wider_pins.py tests/test_native_host.py tests/test_kernels_poly.py tests/test_extend.py tests/test_proof_snapshot.py tests/test_verifier.py -m gpu -x -q > gpurun_out/${TAG}_pytest_stir.log 2>&1
tail -3 gpurun_out/${TAG}_pytest_stir.log
for ldt in stir fri; do
  timeout 600 python bench.py --ldt $ldt --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 > gpurun_out/${TAG}_bench_2p20_${ldt}.json
  python - <<P
import json
d=json.load(open("gpurun_out/${TAG}_bench_2p20_${ldt}.json")); print("$ldt", d["ms_per_step"], d["verified"]["accepted"])
P
done
bash tools/gpu_visit_r4k.sh ${TAG} > /dev/null 2>&1
python - <<P
rows=[l.split(None,2) for l in open("gpurun_out/${TAG}_stir_tail.txt")]
tot={}
for t,d,name in rows:
    name=name.strip().split('(')[0][:40]
    tot[name]=tot.get(name,0)+float(d)
for k,v in sorted(tot.items(), key=lambda kv:-kv[1])[:14]: print(f"{v:9.1f} us  {k}")
P
