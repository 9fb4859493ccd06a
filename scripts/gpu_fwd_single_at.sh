cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for c in 2 3; do
  for at in 256 512 1024 2048; do
    ALTRO_HIP_FWD_SINGLE_AT=$at python bench.py --config $c --no-cpu-baseline 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('fwd_single_at', $at, 'config', $c, d['ms_per_step'], d['roofline']['kernel_wall_ms'])
"
  done
done
done
