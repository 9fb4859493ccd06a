# Where the batched sweeps hand over to the persistent tail kernel: scripts/gpu_persist_at.sh [configs...]
cd $GRAFT_REPO_ROOT
for c in ${@:-2 3}; do
  for at in 256 384 512 768 1024; do
    ALTRO_HIP_PERSIST_AT=$at python bench.py --config $c --no-cpu-baseline 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('persist_at', $at, 'config', $c, d['ms_per_step'], d['config']['sweeps'], d['config']['max_iterations'], d['config']['solved_fraction'], d['config']['mean_iterations'], d['roofline']['kernel_ms'])
"
  done
done
