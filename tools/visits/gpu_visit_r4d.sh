#!/bin/bash
# Round-4 GPU visit D: where the one-rank sharded proof (RCCL) idles (kernel trace + gap analysis), and the 2^22-row LDE with 1024-point
# axes for passes 1 and 3 (TVM_LDE_SPLIT_N1=10: pass 2 on 4096-point rows).
TAG=${1:-visit}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/${TAG}_gaps_prof -o bench -- python $R/bench.py --sharded --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1 )
DB=$(find gpurun_out/${TAG}_gaps_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_gaps.py $DB > gpurun_out/${TAG}_sharded_gaps.txt 2>&1
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_sharded_kernels.txt 2>&1
rm -rf gpurun_out/${TAG}_gaps_prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/${TAG}_gaps_prof2 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1 )
DB=$(find gpurun_out/${TAG}_gaps_prof2 -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_gaps.py $DB > gpurun_out/${TAG}_plain_gaps.txt 2>&1
rm -rf gpurun_out/${TAG}_gaps_prof2
( TVM_LDE_SPLIT_N1=10 timeout 600 python bench.py --log2-rows 22 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>gpurun_out/${TAG}_2p22_n1_10.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_2p22_split_10_12.json
( timeout 600 python bench.py --log2-rows 22 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>gpurun_out/${TAG}_2p22.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_2p22.json
head -40 gpurun_out/${TAG}_sharded_gaps.txt
head -24 gpurun_out/${TAG}_plain_gaps.txt
head -30 gpurun_out/${TAG}_sharded_kernels.txt | cut -c1-140
python - <<P
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d = json.load(open(f))
        print(f, d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d.get("verified", {}).get("accepted"), json.dumps(d["stage_ms"]))
    except Exception as e:
        print(f, "unreadable:", e)
P
for f in gpurun_out/${TAG}_*.err; do echo $f; tail -2 $f; done
