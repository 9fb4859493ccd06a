# round 6: the device-side sweep loop -- bits against the host-paced sweeps, timings, bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for spec in "turn90 4096 4" "turn90 700 2" "obstacles32 700 2" "obstacles 1024 2" ${LOOP_EXTRA:-}; do
  timeout 600 python scripts/probe_loop.py $spec 2>&1 | grep -v "^$" | tail -12
done
for c in ${BENCH_CONFIGS:-2}; do
  timeout 900 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1
  ALTRO_HIP_SWEEP_LOOP=0 timeout 900 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1
done
} 2>&1 | tee gpurun_out/loop.log | cut -c1-1500
