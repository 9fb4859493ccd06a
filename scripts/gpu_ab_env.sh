# A/B of one environment switch on the same box: scripts/gpu_ab_env.sh "<ENV=VAL>" [configs...]
cd $GRAFT_REPO_ROOT
e=$1; shift
for rep in 1 2; do
for c in ${@:-2 3}; do
  for env in X=0 "$e"; do
    env $env python bench.py --config $c --no-cpu-baseline 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$env', 'config', $c, d['ms_per_step'], d['roofline']['kernel_ms'])
"
  done
done
done
