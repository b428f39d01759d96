# round 6, job 16: full GPU suite on the fork-lane tree + the host's own time per step at 2^10 rows
export TMPDIR=/tmp
T=r06_s
mkdir -p gpurun_out
( timeout 600 python bench.py --log2-rows 10 --steps 3 --warmup 2 --host-trace 2 --no-extras --no-cpu-baseline 2>&1 | grep tvmh | tail -40 ) > gpurun_out/${T}_host_trace_2p10.txt
( timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/${T}_pytest_gpu_full_suite.log
( timeout 600 python __graft_entry__.py --smoke 2>&1 | tail -2 ) > gpurun_out/${T}_smoke.log
cat gpurun_out/${T}_host_trace_2p10.txt gpurun_out/${T}_pytest_gpu_full_suite.log gpurun_out/${T}_smoke.log
