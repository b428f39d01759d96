#!/bin/bash
# round 4, visit T: the kernel sequence of one default (FRI) proof -- every launch outside the three big groups (row hashing, LDE passes,
# AIR parts) with its grid, to see which generic transforms and small kernels are left
TAG=${1:-r04_t}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>$R/gpurun_out/${TAG}.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench.json
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
python - "$DB" > gpurun_out/${TAG}_sequence.txt <<'P'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "k_pad_main_table" in r[0]]
i0, i1 = idx[2], idx[3] if len(idx) > 3 else len(rows)     # the third proof: warm, timed
t0 = rows[i0][1]
for name, s, e, gx, gy, wx in rows[i0:i1]:
    print(round((s - t0) / 1e3, 1), round((e - s) / 1e3, 1), gx // max(wx, 1), gy, wx, name[:70])
P
rm -rf gpurun_out/${TAG}_prof
wc -l gpurun_out/${TAG}_sequence.txt
python - <<P
import json
d = json.load(open("gpurun_out/${TAG}_bench.json")); print(d["ms_per_step"])
P
