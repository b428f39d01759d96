# round 6, job 15: trace randomizers drawn on the device + the fork threshold swept at 2^13 .. 2^16 rows; host timeline at 2^10 / 2^14
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=r06_r
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_randomness.py tests/test_kernels_air.py tests/test_proof_snapshot.py tests/test_native_host.py -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/${T}_pytest.log
cat gpurun_out/${T}_pytest.log
for L in 10 12; do
  ( timeout 600 python bench.py --log2-rows $L --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_2p$L.err | tail -1 ) > gpurun_out/${T}_bench_2p${L}.json
  python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p${L}.json')); print($L, d['ms_per_step'], d['value'], d.get('verified',{}).get('accepted'), d.get('stage_ms_cpp_host'))"
done
for L in 13 14 15 16; do for F in 0 64 128 256 512; do
  ( timeout 600 python bench.py --log2-rows $L --air-fork $F --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_2p${L}_f$F.err | tail -1 ) > gpurun_out/${T}_bench_2p${L}_f$F.json
  python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p${L}_f$F.json')); print($L, 'fork', $F, d['ms_per_step'], d.get('verified',{}).get('accepted'), 'AIR', d['stage_ms_cpp_host']['AIR quotients'])"
done; done
for L in 10 14; do
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --hip-trace -d $R/gpurun_out/prof_h$L -o bench -- python $R/bench.py --log2-rows $L --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -2 ) > gpurun_out/${T}_rocprof_hip_2p$L.log
DB=$(find gpurun_out/prof_h$L -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python tools/rocprof_summary.py $DB > gpurun_out/${T}_bench_2p${L}_kernels.txt
  python tools/rocprof_host_timeline.py $DB 25 > gpurun_out/${T}_bench_2p${L}_host_timeline.txt 2>&1
fi
rm -rf gpurun_out/prof_h$L
cat gpurun_out/${T}_bench_2p${L}_host_timeline.txt | cut -c1-300 | head -90
done
