#!/bin/bash
# Round-4 GPU visit K: where the STIR proof (the reference's default low-degree test at this height) spends its time
TAG=${1:-visit}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && TVMH_TRACE=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/bench.py --ldt stir --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>$R/gpurun_out/${TAG}_stir.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_stir.json
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_stir_kernels.txt
[ -n "$DB" ] && python tools/rocprof_gaps.py $DB > gpurun_out/${TAG}_stir_gaps.txt
[ -n "$DB" ] && python - "$DB" > gpurun_out/${TAG}_stir_tail.txt <<'P'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
# the last proof: from its last k_deep kernel to the end of the trace's last proof (before the profile pass): print the kernel sequence after k_deep with times
idx = [i for i, r in enumerate(rows) if "k_deep" in r[0]]
i0 = idx[2] if len(idx) > 2 else idx[-1]
t0 = rows[i0][1]
out = []
for name, s, e in rows[i0:i0 + 400]:
    out.append((round((s - t0) / 1e3, 1), round((e - s) / 1e3, 1), name[:60]))
    if "k_pad_main_table" in name: break
for o in out: print(*o)
P
rm -rf gpurun_out/${TAG}_prof
python - <<P
import json
d = json.load(open("gpurun_out/${TAG}_bench_stir.json")); print(d["ms_per_step"], json.dumps(d["stage_ms"]))
P
grep "tvmh" gpurun_out/${TAG}_stir.err | tail -9
head -40 gpurun_out/${TAG}_stir_kernels.txt | cut -c1-140
head -30 gpurun_out/${TAG}_stir_gaps.txt
wc -l gpurun_out/${TAG}_stir_tail.txt
