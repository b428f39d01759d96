# round 6, closing visit of the LAST session (short-trace changes on top of the closing visit's kernels): the full GPU suite, smoke, the driver's bench command, a kernel trace of it, the PMC counters of
# the shipped hot kernels (2^20 rows and configs[2]'s height) -> profiles/kernel_counters.json (stamped with COMMIT), static properties,
# device idle gaps, the tip5 floor.   usage: bash tools/r06_final.sh <commit>
COMMIT=${1:-unknown}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=r06_z
mkdir -p gpurun_out
( timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/${T}_pytest_gpu_full_suite.log
( timeout 600 python __graft_entry__.py --smoke 2>&1 | tail -3 ) > gpurun_out/${T}_smoke.log
( timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/${T}_bench.err | tail -1 ) > gpurun_out/${T}_bench_2p20.json
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -2 ) > gpurun_out/${T}_rocprof.log
DB=$(find gpurun_out/${T}_prof -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python tools/rocprof_summary.py $DB > gpurun_out/${T}_bench_2p20_kernels.txt
  python tools/rocprof_gaps.py $DB > gpurun_out/${T}_device_idle_gaps.txt 2>&1
fi
rm -rf gpurun_out/${T}_prof
python tools/kernel_static_properties.py > gpurun_out/${T}_kernel_static_properties.txt 2>&1
# counters (separate --pmc passes, tools/pmc.sh) over one 96-column chunk + its hashing: 2^20 rows, then 2^22 rows as a further shape
bash tools/pmc.sh ${T}_pmc_lde_hash python $R/tools/probe.py 20 96 0 1 > /dev/null 2>&1
cp gpurun_out/${T}_pmc_lde_hash_summary.txt gpurun_out/${T}_pmc_lde_hash.txt
bash tools/pmc.sh ${T}_pmc_lde_hash_2p22 python $R/tools/probe.py 22 96 0 1 > /dev/null 2>&1
cp gpurun_out/${T}_pmc_lde_hash_2p22_summary.txt gpurun_out/${T}_pmc_lde_hash_2p22.txt
cp profiles/kernel_counters.json gpurun_out/kernel_counters_before.json
python tools/kernel_counters.py gpurun_out/${T}_pmc_lde_hash.txt 96 20 $COMMIT > /dev/null && python tools/kernel_counters.py gpurun_out/${T}_pmc_lde_hash_2p22.txt 96 22 $COMMIT 8 2p22_x8 > /dev/null
cp profiles/kernel_counters.json gpurun_out/kernel_counters.json

( timeout 900 python bench.py --log2-rows 22 --steps 3 --warmup 1 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_2p22.err | tail -1 ) > gpurun_out/${T}_bench_2p22.json
for L in 10 12 14; do
  ( timeout 600 python bench.py --log2-rows $L --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_2p$L.err | tail -1 ) > gpurun_out/${T}_bench_2p${L}.json
done
( timeout 600 python bench.py --log2-rows 16 --ldt auto --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_2p16.err | tail -1 ) > gpurun_out/${T}_bench_2p16_ldt_auto.json
( timeout 600 python bench.py --log2-rows 18 --ldt auto --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_2p18.err | tail -1 ) > gpurun_out/${T}_bench_2p18_ldt_auto.json
for L in 10 12 14; do python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p${L}.json')); print($L, d['ms_per_step'], d['value'], d.get('verified',{}).get('accepted'))"; done
for L in 16 18; do python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_2p${L}_ldt_auto.json')); print($L, d['ms_per_step'], d['value'], d.get('verified',{}).get('accepted'), d['config'].get('ldt'))"; done
cat gpurun_out/${T}_pytest_gpu_full_suite.log gpurun_out/${T}_smoke.log
python - <<P
import json
d=json.load(open("gpurun_out/${T}_bench_2p20.json"))
print(d["ms_per_step"], d["value"], json.dumps(d["roofline"])[:700])
print(json.dumps(d.get("stage_ms_cpp_host")))
print(json.dumps(d["cpu_baseline"])[:300])
d=json.load(open("gpurun_out/${T}_bench_2p22.json")); print("2^22", d["ms_per_step"], d["value"])
k=json.load(open("gpurun_out/kernel_counters.json")); print(k["commit"], k["lde"]["hbm_bytes_per_trace_cell"], k["lde"]["wave_valu_instructions_per_trace_cell"], k["hash_rows"]["wave_valu_instructions_per_row_permutation"])
print(k["shapes"]["2p22_x8"]["lde"]["hbm_bytes_per_trace_cell"], {n: (v["fetch_bytes_per_cell"], v["lds_bank_conflict_share"]) for n, v in k["shapes"]["2p22_x8"]["lde"]["kernels"].items()})
P
head -12 gpurun_out/${T}_bench_2p20_kernels.txt | cut -c1-150
