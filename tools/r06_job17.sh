# round 6, job 17: short traces -- smaller pass-2 tiles, the pinned staging ring (no drained stream per small host array), the five
# segment polynomials in one round trip; tests, bench lines 2^10 .. 2^16, default 2^20, host timeline at 2^10
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=r06_t
mkdir -p gpurun_out
( timeout 1800 python -m pytest tests/test_kernels_ntt.py tests/test_kernels_poly.py tests/test_kernels_hash.py tests/test_kernels_air.py tests/test_proof_snapshot.py tests/test_native_host.py tests/test_wider_pins.py tests/test_stir.py -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/${T}_pytest.log
cat gpurun_out/${T}_pytest.log
for L in 10 11 12 13 14 15 16; do
  ( timeout 600 python bench.py --log2-rows $L --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_2p$L.err | tail -1 ) > gpurun_out/${T}_bench_2p${L}.json
  python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p${L}.json')); print($L, d['ms_per_step'], d['value'], d.get('verified',{}).get('accepted'), d.get('stage_ms_cpp_host'))"
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --hip-trace -d $R/gpurun_out/prof_h -o bench -- python $R/bench.py --log2-rows 10 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -2 ) > gpurun_out/${T}_rocprof_hip_2p10.log
DB=$(find gpurun_out/prof_h -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python tools/rocprof_summary.py $DB > gpurun_out/${T}_bench_2p10_kernels.txt
  python tools/rocprof_host_timeline.py $DB 25 > gpurun_out/${T}_bench_2p10_host_timeline.txt 2>&1
fi
rm -rf gpurun_out/prof_h
head -24 gpurun_out/${T}_bench_2p10_host_timeline.txt | cut -c1-200
head -16 gpurun_out/${T}_bench_2p10_kernels.txt | cut -c1-150
( timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_2p20.err | tail -1 ) > gpurun_out/${T}_bench_2p20.json
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p20.json')); print(20, d['ms_per_step'], d['value'], d.get('verified',{}).get('accepted'), d.get('stage_ms_cpp_host'))"
