#!/bin/bash
# first GPU visit: parity tests, smoke, stage timings, rocprof kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing|gfx9" | head -4 > gpurun_out/gpu_info.txt
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/pytest_gpu.log
( timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -5 ) > gpurun_out/smoke.log
( timeout 600 python tools/probe.py 20 379 91 3 2>&1 | tail -20 ) > gpurun_out/probe_2p20.log
( timeout 300 python tools/probe.py 16 379 91 2 2>&1 | tail -20 ) > gpurun_out/probe_2p16.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_probe -o probe -- python $GRAFT_REPO_ROOT/tools/probe.py 20 379 91 1 2>&1 | tail -5 ) > gpurun_out/rocprof.log
ls -R gpurun_out | head -40
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/probe_2p20.log gpurun_out/probe_2p16.log
