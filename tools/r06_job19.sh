# round 6, job 19: the same with more hardware queues for the process's streams (an APPLICATION setting: GPU_MAX_HW_QUEUES, default 4)
export TMPDIR=/tmp
T=r06_u
mkdir -p gpurun_out
for Q in 8 16; do for L in 10 14; do
  ( GPU_MAX_HW_QUEUES=$Q timeout 900 python tools/concurrent_provers.py $L 40 1,4,8,16 2>gpurun_out/${T}_concurrent_q${Q}_2p$L.err | tail -1 ) > gpurun_out/${T}_concurrent_provers_hwq${Q}_2p$L.json
  echo "queues $Q"; cat gpurun_out/${T}_concurrent_provers_hwq${Q}_2p$L.json
done; done
( GPU_MAX_HW_QUEUES=8 timeout 600 python bench.py --log2-rows 10 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 ) > gpurun_out/${T}_bench_2p10_hwq8.json
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p10_hwq8.json')); print('2^10 hwq8', d['ms_per_step'], d['stage_ms_cpp_host']['AIR quotients'])"
