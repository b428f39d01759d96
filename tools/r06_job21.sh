# round 6, job 21: P processes x K proving threads on one GPU (short traces): does a process per few threads lift the ceiling of one
# process's launch path?
export TMPDIR=/tmp
T=r06_w
mkdir -p gpurun_out
for L in 10 14; do for CFG in "1 8" "2 4" "4 2" "4 4" "8 2"; do set -- $CFG; P=$1; K=$2
  START=$(python -c "import time; print(time.time() + 25)")
  for i in $(seq 1 $P); do
    ( timeout 600 python tools/concurrent_provers.py $L 150 $K $START 2>/dev/null | tail -1 > gpurun_out/${T}_p${P}_k${K}_2p${L}_$i.json ) &
  done
  wait
  python - <<P2
import json, glob
runs = [json.load(open(f))["runs"][0] for f in sorted(glob.glob("gpurun_out/${T}_p${P}_k${K}_2p${L}_*.json"))]
lo = max(r["window"][0] for r in runs); hi = min(r["window"][1] for r in runs)
print(json.dumps({"log2_rows": $L, "processes": $P, "threads_per_process": $K, "proofs_per_s_sum": round(sum(r["proofs_per_s"] for r in runs), 1),
                  "common_window_share": round((hi - lo) / max(r["window"][1] - r["window"][0] for r in runs), 2)}))
P2
done; done | tee gpurun_out/${T}_processes_x_threads.txt
rm -f gpurun_out/${T}_p*_k*_2p*_*.json
