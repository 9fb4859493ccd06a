cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
python bench.py --config 3 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('run', $i, d['ms_per_step'], d['config']['sweeps'], d['config']['max_iterations'], d['config']['solved_fraction'], d['config']['mean_iterations'], d['roofline']['kernel_ms'], d['roofline'].get('tail_iterations'))
"
done
