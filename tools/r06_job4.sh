# round 6, job 4: row hashing with the next block staged through LDS a permutation ahead (default build) and with the paired / lean
# tail on top (variant pt1); the 256- / 512-point row kernels of the LDE at 2^16 .. 2^19 rows
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=r06_d
mkdir -p gpurun_out
cp triton_vm_amd/libtriton_hip.so /tmp/default.so
one() {  # name
  ( timeout 900 python -m pytest tests/test_kernels_hash.py tests/test_kernels_ntt.py tests/test_stir.py tests/test_proof_snapshot.py -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/${T}_$1_pytest_gpu.log
  ( timeout 600 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_$1_bench.err | tail -1 ) > gpurun_out/${T}_$1_bench.json
  bash tools/r06_tip5_floor.sh ${T}_$1 > /dev/null
}
one default
cp triton_vm_amd/libtriton_hip_pt1.so triton_vm_amd/libtriton_hip.so
one pt1
cp /tmp/default.so triton_vm_amd/libtriton_hip.so
for L in 16 17 18 19; do
  ( timeout 600 python bench.py --log2-rows $L --ldt auto --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>gpurun_out/${T}_bench_2p$L.err | tail -1 ) > gpurun_out/${T}_bench_2p$L.json
done
for L in 16 18; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$L -o bench -- python $R/bench.py --log2-rows $L --ldt auto --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -3 ) > gpurun_out/${T}_rocprof_2p$L.log
  DB=$(find gpurun_out/prof_$L -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${T}_bench_2p${L}_kernels.txt
  rm -rf gpurun_out/prof_$L
done
( timeout 1500 python -m pytest tests/test_wider_pins.py tests/test_sharded_host.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/${T}_pytest_gpu_pins_sharded.log
cat gpurun_out/${T}_*pytest*.log
python - <<P
import json,glob
for f in sorted(glob.glob("gpurun_out/${T}_*bench*.json")):
    try:
        d=json.load(open(f))
        print(f, d["ms_per_step"], d["value"], (d.get("verified") or {}).get("accepted"), json.dumps(d.get("stage_ms_cpp_host") or d["stage_ms"]))
    except Exception as e:
        print(f, "unreadable", e)
P
tail -4 gpurun_out/${T}_default_tip5_floor.txt gpurun_out/${T}_pt1_tip5_floor.txt
head -12 gpurun_out/${T}_bench_2p18_kernels.txt | cut -c1-150
