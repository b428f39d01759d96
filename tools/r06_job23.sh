# round 6, job 23: the default bench line once more (the closing visit's box ran every VALU-bound kernel 5-7 % slower than the boxes before it)
export TMPDIR=/tmp
T=r06_y
mkdir -p gpurun_out
( timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/${T}_bench.err | tail -1 ) > gpurun_out/${T}_bench_2p20.json
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p20.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['stage_ms_cpp_host'])"
for L in 10 12 14; do
  ( timeout 600 python bench.py --log2-rows $L --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 ) > gpurun_out/${T}_bench_2p${L}.json
  python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p${L}.json')); print($L, d['ms_per_step'], d.get('verified',{}).get('accepted'))"
done
rocm-smi --showclocks 2>/dev/null | head -12
