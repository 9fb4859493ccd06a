# round 6: where the device-side sweep loop beats the host-paced sweeps (batch sizes between the hand-over point and the full batch)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for kind in turn90 obstacles32 obstacles; do
  for b in ${CROSS_BATCHES:-1024 1536 2048 3072}; do
    timeout 600 python scripts/probe_loop.py $kind $b 3 2>&1 | grep -v "^$" | grep "ms per solve\|IDENTICAL\|DIFFERENT\|FAILED"
  done
done
} 2>&1 | tee gpurun_out/loop_crossover.log | cut -c1-300
