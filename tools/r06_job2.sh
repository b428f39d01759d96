# round 6, job 2: the two-byte S-box table (k_hash_rows_lut16) against the byte-table kernel on one box; side-lane tests on the GPU
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=r06_b
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_kernels_hash.py tests/test_kernels_field.py tests/test_sharded_host.py tests/test_proof_snapshot.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/${T}_pytest_gpu.log
for G in 0 256 512; do
  ( timeout 600 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --hash-lut16 $G 2>gpurun_out/${T}_lut16_$G.err | tail -1 ) > gpurun_out/${T}_bench_lut16_$G.json
done
for G in 0 256; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --hash-lut16 $G 2>&1 | tail -3 ) > gpurun_out/${T}_rocprof_$G.log
  DB=$(find gpurun_out/${T}_prof -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${T}_kernels_lut16_$G.txt
  rm -rf gpurun_out/${T}_prof
done
cat gpurun_out/${T}_pytest_gpu.log
python - <<P
import json,glob
for f in sorted(glob.glob("gpurun_out/${T}_bench_lut16_*.json")):
    try:
        d=json.load(open(f))
        print(f, d["ms_per_step"], d.get("verified"), json.dumps(d.get("stage_ms_cpp_host") or d["stage_ms"]))
    except Exception as e:
        print(f, "unreadable", e)
P
for f in gpurun_out/${T}_kernels_lut16_*.txt; do echo $f; head -8 $f | cut -c1-150; done
