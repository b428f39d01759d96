export TMPDIR=/tmp; cd /tmp
for v in default fake; do
rocprofv3 --kernel-trace --stats -d /tmp/ap_$v -o r -- python $GRAFT_REPO_ROOT/tools/air_bench.py $v 2>&1 | grep air_ms
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/ap_$v -name "*.db" | head -1) | grep "k_air" | awk '{print $1, $3, $4}'
done
