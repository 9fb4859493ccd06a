# A/B of two builds of the library on the same box: scripts/gpu_ab.sh <libA> <libB> [configs...]
cd $GRAFT_REPO_ROOT
a=$1; b=$2; shift 2
for rep in 1 2; do
for c in ${@:-2 3}; do
  for lib in $a $b; do
    ALTRO_HIP_LIB=$lib python bench.py --config $c --no-cpu-baseline 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$lib'.split('/')[-1], 'config', $c, d['ms_per_step'], d['roofline']['kernel_ms'])
"
  done
done
done
