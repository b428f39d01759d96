# round 6, job 5: GPU tests of the round's parity additions and of the side lane; lockstep 8-rank runs with the asynchronous exchange
export TMPDIR=/tmp
T=r06_e
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests/test_wider_pins.py tests/test_sharded_host.py tests/test_kernels_ntt.py tests/test_error_paths.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/${T}_pytest_gpu.log
cat gpurun_out/${T}_pytest_gpu.log
for CS in 0 2 4; do
  ( timeout 900 python bench.py --simulate-gpus 8 --column-split $CS --steps 3 --warmup 1 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_sim8_cs$CS.err | tail -1 ) > gpurun_out/${T}_bench_simulated_8_ranks_2p20_column_split_$CS.json
done
python - <<P
import json,glob
for f in sorted(glob.glob("gpurun_out/${T}_bench_simulated*.json")):
    try:
        d=json.load(open(f)); s=d.get("simulated_multi_gpu") or d
        print(f, d.get("ms_per_step"), json.dumps({k:v for k,v in s.items() if not isinstance(v,(dict,list))})[:600])
    except Exception as e:
        print(f, "unreadable", e)
P
