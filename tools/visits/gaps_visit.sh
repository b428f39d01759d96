export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/gaps_prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
cd $R; DB=$(find gpurun_out/gaps_prof -name '*.db' | head -1); python tools/rocprof_gaps.py $DB > gpurun_out/r03_v_gaps.txt 2>&1; rm -rf gpurun_out/gaps_prof; cat gpurun_out/r03_v_gaps.txt | head -80
