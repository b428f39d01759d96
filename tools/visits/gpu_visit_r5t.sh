#!/bin/bash
# Round 5: a 2^23-row proof on ONE GPU through the production host's memory policy (the cached extension would be 326 GiB: the host
# finds that out and proves coset-wise), verified; and the same with the pass count given.
TAG=${1:-r05_t}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python bench.py --log2-rows 23 --memory-policy --steps 1 --warmup 1 --no-cpu-baseline --no-extras 2>gpurun_out/${TAG}_policy.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_2p23_memory_policy.json
tail -3 gpurun_out/${TAG}_policy.err
python - <<P
import json
try:
    d = json.load(open("gpurun_out/${TAG}_bench_2p23_memory_policy.json"))
    print(d["ms_per_step"], d["value"], d.get("memory_policy"), d.get("verified", {}).get("accepted"), d["config"]["parallelism"])
    print(json.dumps(d.get("stage_ms")))
except Exception as e:
    print("no line:", e)
P
