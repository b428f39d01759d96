# round 6, job 22: short traces, third step -- every column in one LDE chunk, Merkle levels seven to a launch, node 0 written by the top kernel:
# tests, bench lines 2^10 .. 2^16 (+ the subtrees A/B), 2^20, concurrent provers
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=r06_x
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests/test_kernels_ntt.py tests/test_kernels_hash.py tests/test_kernels_air.py tests/test_staging_ring.py tests/test_proof_snapshot.py tests/test_native_host.py tests/test_wider_pins.py tests/test_stir.py tests/test_sharded_host.py tests/test_jit_prover.py -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/${T}_pytest.log
cat gpurun_out/${T}_pytest.log
for L in 10 11 12 13 14 15 16; do
  ( timeout 600 python bench.py --log2-rows $L --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_2p$L.err | tail -1 ) > gpurun_out/${T}_bench_2p${L}.json
  python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p${L}.json')); print($L, d['ms_per_step'], d['value'], d.get('verified',{}).get('accepted'), d.get('stage_ms_cpp_host'))"
done
for L in 10 12 14; do
  ( timeout 600 python bench.py --log2-rows $L --ctx-option 6=0 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 ) > gpurun_out/${T}_bench_2p${L}_one_level_per_launch.json
  python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p${L}_one_level_per_launch.json')); print($L, 'one level per launch', d['ms_per_step'])"
done
( timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_2p20.err | tail -1 ) > gpurun_out/${T}_bench_2p20.json
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p20.json')); print(20, d['ms_per_step'], d['value'], d.get('verified',{}).get('accepted'), d.get('stage_ms_cpp_host'))"
( timeout 600 python bench.py --steps 10 --warmup 3 --ctx-option 6=0 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 ) > gpurun_out/${T}_bench_2p20_one_level_per_launch.json
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p20_one_level_per_launch.json')); print(20, 'one level per launch', d['ms_per_step'], d['stage_ms_cpp_host']['FRI'])"
for L in 10 14; do
  ( timeout 900 python tools/concurrent_provers.py $L 60 1,4,8 2>/dev/null | tail -1 ) > gpurun_out/${T}_concurrent_provers_2p$L.json
  cat gpurun_out/${T}_concurrent_provers_2p$L.json
done
