# round 6, job 18: K proving threads (one context each) on one GPU, short traces
export TMPDIR=/tmp
T=r06_u
mkdir -p gpurun_out
for L in 10 12 14; do
  ( timeout 900 python tools/concurrent_provers.py $L 40 1,2,4,8,16 2>gpurun_out/${T}_concurrent_2p$L.err | tail -1 ) > gpurun_out/${T}_concurrent_provers_2p$L.json
  cat gpurun_out/${T}_concurrent_provers_2p$L.json; tail -3 gpurun_out/${T}_concurrent_2p$L.err
done
