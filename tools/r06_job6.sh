# round 6, job 6: the CPU port over ALL the work of one prove() (--cpu-baseline full); lockstep 2 / 4 / 8 ranks (+ the column split's
# bracket at 8); configs[2]'s height on one GPU
export TMPDIR=/tmp
T=r06_f
mkdir -p gpurun_out
( timeout 900 python bench.py --steps 3 --warmup 1 --no-extras --cpu-baseline full 2>gpurun_out/${T}_cpu_full.err | tail -1 ) > gpurun_out/${T}_bench_2p20_with_cpu_baseline_full.json
python - <<P
import json
d=json.load(open("gpurun_out/${T}_bench_2p20_with_cpu_baseline_full.json"))
json.dump(d["cpu_baseline"], open("gpurun_out/${T}_cpu_baseline_full_2p20.json","w"), indent=1)
print(json.dumps(d["cpu_baseline"])[:900])
P
for N in 2 4 8; do
  ( timeout 900 python bench.py --simulate-gpus $N --steps 3 --warmup 1 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_sim$N.err | tail -1 ) > gpurun_out/${T}_bench_simulated_${N}_ranks_2p20.json
done
( timeout 900 python bench.py --simulate-gpus 8 --column-split 2 --steps 3 --warmup 1 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_sim8cs2.err | tail -1 ) > gpurun_out/${T}_bench_simulated_8_ranks_2p20_column_split_2.json
for L in 21 22; do
  ( timeout 900 python bench.py --log2-rows $L --steps 3 --warmup 1 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_2p$L.err | tail -1 ) > gpurun_out/${T}_bench_2p$L.json
done
python - <<P
import json,glob
for f in sorted(glob.glob("gpurun_out/${T}_bench_*.json")):
    try:
        d=json.load(open(f)); s=d.get("simulated_multi_gpu") or {}
        print(f, d.get("ms_per_step"), d.get("value"), s.get("slowest_rank_sum_ms"), s.get("projected_ms_per_proof"), json.dumps(s.get("column_split_bracket"))[:400])
    except Exception as e:
        print(f, "unreadable", e)
P
