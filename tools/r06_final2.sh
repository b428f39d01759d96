# round 6, the last visit: the full GPU suite (with the parity tests added after the closing visit), smoke, the default bench line
export TMPDIR=/tmp
T=r06_last
mkdir -p gpurun_out
( timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/${T}_pytest_gpu_full_suite.log
( timeout 600 python __graft_entry__.py --smoke 2>&1 | tail -2 ) > gpurun_out/${T}_smoke.log
( timeout 900 python bench.py 2>gpurun_out/${T}_bench.err | tail -1 ) > gpurun_out/${T}_bench_2p20_default_flags.json
cat gpurun_out/${T}_pytest_gpu_full_suite.log gpurun_out/${T}_smoke.log
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p20_default_flags.json')); print(d['ms_per_step'], d['value'], d['steps'], d['warmup'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline'].get('full_run_on_record',{}).get('value'))"
