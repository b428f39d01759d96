#!/bin/bash
# Round-4 GPU visit B: the wider pins (STIR regression digest of the second snapshot program, whole-proof equality with the oracle prover at
# 2^12 / 2^14 rows), BASELINE configs 3-5 timed with kernel traces, the sharded code path over RCCL with one rank.
TAG=${1:-visit}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp tests/golden/stir_regression_digests.json gpurun_out/stir_regression_digests.json
( timeout 900 python tests/golden/make_stir_regression_digests.py gpu $R/gpurun_out/stir_regression_digests.json 2>&1 | tail -2 ) > gpurun_out/${TAG}_stir_digests.log
cp gpurun_out/stir_regression_digests.json tests/golden/stir_regression_digests.json
( timeout 1500 python -m pytest tests/test_wider_pins.py tests/test_sharded_host.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/${TAG}_pytest_gpu.log
( timeout 600 python bench.py --sharded --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>gpurun_out/${TAG}_sharded.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_sharded_1rank_rccl.json
run_config() {  # name, bench arguments...
  local name=$1; shift
  ( timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" 2>gpurun_out/${TAG}_${name}.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_${name}.json
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_${name} -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" 2>&1 | tail -2 ) > gpurun_out/${TAG}_rocprof_${name}.log
  DB=$(find gpurun_out/${TAG}_prof_${name} -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_bench_${name}_kernels.txt
  rm -rf gpurun_out/${TAG}_prof_${name}
}
run_config u32_2p20 --program u32
run_config sponge_blowup4 --program sponge --log2-expansion 4
run_config 2p22 --log2-rows 22
run_config 2p21 --log2-rows 21
cat gpurun_out/${TAG}_stir_digests.log gpurun_out/${TAG}_pytest_gpu.log
python - <<P
import json
for name in ("bench_sharded_1rank_rccl", "bench_u32_2p20", "bench_sponge_blowup4", "bench_2p22", "bench_2p21"):
    try:
        d = json.load(open("gpurun_out/${TAG}_%s.json" % name))
        print(name, d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d.get("verified", {}).get("accepted"))
        print("  stage_ms", json.dumps(d["stage_ms"]))
        if "ranks" in d: print("  ranks", json.dumps(d["ranks"])[:1500])
    except Exception as e:
        print(name, "unreadable:", e)
P
tail -3 gpurun_out/${TAG}_*.err
head -14 gpurun_out/${TAG}_bench_2p22_kernels.txt | cut -c1-150
head -14 gpurun_out/${TAG}_bench_2p21_kernels.txt | cut -c1-150
