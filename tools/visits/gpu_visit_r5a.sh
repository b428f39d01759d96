#!/bin/bash
# Round-5 GPU visit A: (1) the parity tests touched by the round's first changes (LDE dispatcher without experiment knobs, pool
# intermediates, communicator share / abort hooks, the collective pass-count decision) and the NEW full-size sharded equality tests
# (8 ranks at 2^16 / 2^20 / 2^22 rows); (2) BASELINE configs[2] (2^22 rows, 8 ranks) and configs[4]'s shape (expansion 16, 8 ranks)
# through the sharded C++ host in lockstep with ONE copy of the replicated tables; (3) the default bench; (4) where pass 2 of the
# LDE waits (PMC).
TAG=${1:-r05_a}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests/test_kernels_ntt.py tests/test_kernels_hash.py tests/test_kernels_poly.py tests/test_sharded_host.py tests/test_proof_snapshot.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/${TAG}_pytest_kernels_sharded.log
( timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q -k "eight_ranks" 2>&1 | tail -8 ) > gpurun_out/${TAG}_pytest_fullsize_sharded.log
cat gpurun_out/${TAG}_pytest_kernels_sharded.log gpurun_out/${TAG}_pytest_fullsize_sharded.log
( timeout 900 python bench.py --log2-rows 22 --simulate-gpus 8 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>gpurun_out/${TAG}_sim22.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_simulated_8_ranks_2p22.json
( timeout 900 python bench.py --program sponge --log2-expansion 4 --simulate-gpus 8 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>gpurun_out/${TAG}_sim_sponge.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_simulated_8_ranks_sponge_blowup4.json
( timeout 900 python bench.py 2>gpurun_out/${TAG}_bench.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_2p20.json
bash tools/pmc_wait_split.sh ${TAG}_pmc_wait python $R/tools/probe.py 20 96 0 1
tail -5 gpurun_out/${TAG}_sim22.err gpurun_out/${TAG}_sim_sponge.err gpurun_out/${TAG}_bench.err 2>/dev/null | cut -c1-300
python - <<P
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench*.json")):
    try:
        d = json.load(open(f))
        print(f, d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d.get("verified", {}).get("accepted"))
        for k in ("exact_air_real", "reference_default_ldt", "pcie_inclusive", "synthetic_hot_path"):
            if k in d: print("  ", k, d[k]["ms_per_step"])
        if "simulated_multi_gpu" in d:
            s = d["simulated_multi_gpu"]
            print("   sim", s.get("ranks"), s.get("slowest_rank_sum_ms"), s.get("projected_ms_per_proof"), s.get("bytes_sent_per_rank"), s.get("same_proof_as_single_gpu"), s.get("all_ranks_same_proof"), s.get("error"))
            print("   ", json.dumps({k: max(v) for k, v in s.get("stage_ms_per_rank", {}).items()}))
        print("   stages", json.dumps(d.get("stage_ms")))
    except Exception as e:
        print(f, "unreadable:", e)
P
grep -A12 "k_lde_pass2_rows" gpurun_out/${TAG}_pmc_wait_summary.txt | head -60
