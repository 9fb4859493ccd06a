# round 3: more than four chains of sweeps (GPU_MAX_HW_QUEUES=8 is set by bench.py) x hand-over point
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in ${@:-3 2}; do
  for ch in 4 6 8; do
    for at in 256 512; do
    ALTRO_HIP_CHAINS=$ch ALTRO_HIP_PERSIST_AT=$at timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --no-latency 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; print('chains', $ch, 'persist_at', $at, 'config', $c, d['ms_per_step'], d['config']['sweeps'], 'concurrent', r.get('concurrent_chains'), 'wall', r['kernel_wall_ms'])
" | tee -a gpurun_out/chains8.log
    done
  done
done
