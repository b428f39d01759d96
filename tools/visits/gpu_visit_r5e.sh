#!/bin/bash
# Round-5 GPU visit E: the column split of the inverse transforms (TVMH_OPTION_COLUMN_SPLIT): parity at full size over eight ranks,
# the lockstep measurement with and without it at 2^20 and 2^22 rows; LDE kernels after the removal of k_lde_pass2_rows.
TAG=${1:-r05_e}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests/test_kernels_ntt.py tests/test_sharded_host.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 ) > gpurun_out/${TAG}_pytest_kernels.log
( timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q -k "eight_ranks" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 ) > gpurun_out/${TAG}_pytest_fullsize_sharded.log
cat gpurun_out/${TAG}_pytest_kernels.log gpurun_out/${TAG}_pytest_fullsize_sharded.log
for CS in 0 2 4; do
  ( timeout 600 python bench.py --simulate-gpus 8 --column-split $CS --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>gpurun_out/${TAG}_sim20_$CS.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_simulated_8_ranks_2p20_column_split_$CS.json
done
for CS in 0 4; do
  ( timeout 900 python bench.py --log2-rows 22 --simulate-gpus 8 --column-split $CS --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>gpurun_out/${TAG}_sim22_$CS.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_simulated_8_ranks_2p22_column_split_$CS.json
done
tail -3 gpurun_out/${TAG}_sim2*.err | cut -c1-300
python - <<P
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench*.json")):
    try:
        d = json.load(open(f))
        s = d["simulated_multi_gpu"]
        print(f, d["ms_per_step"], "sim", s.get("ranks"), s.get("slowest_rank_sum_ms"), s.get("projected_ms_per_proof"), s.get("bytes_sent_per_rank"), s.get("same_proof_as_single_gpu"), s.get("all_ranks_same_proof"), s.get("error"))
        print("   ", json.dumps({k: max(v) for k, v in s.get("stage_ms_per_rank", {}).items()}))
        print("   ", json.dumps(s.get("column_split") or s.get("column_split_bracket")))
    except Exception as e:
        print(f, "unreadable:", e)
P
