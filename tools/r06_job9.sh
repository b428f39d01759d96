# round 6, job 9: BASELINE's other configurations on the final tree (one GPU, each proof verified)
export TMPDIR=/tmp
T=r06_k
mkdir -p gpurun_out
( timeout 900 python bench.py --program u32 --steps 3 --warmup 1 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_u32.err | tail -1 ) > gpurun_out/${T}_bench_u32_2p20.json
( timeout 1500 python bench.py --program sponge --log2-expansion 4 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_sponge.err | tail -1 ) > gpurun_out/${T}_bench_sponge_blowup4.json
( timeout 900 python bench.py --ldt stir --steps 3 --warmup 1 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_stir.err | tail -1 ) > gpurun_out/${T}_bench_2p20_stir.json
python - <<P
import json,glob
for f in sorted(glob.glob("gpurun_out/${T}_bench_*.json")):
    try:
        d=json.load(open(f)); print(f, d["ms_per_step"], d["value"], (d.get("verified") or {}).get("accepted"))
    except Exception as e:
        print(f, "unreadable", e)
P
