# config 4 bench lines (repeat) under the environment given on the command line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3; do
    timeout 300 python bench.py --config 4 --no-cpu-baseline --no-other-configs --no-latency 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; print('config 4', d['ms_per_step'], d['value'], 'kernel_ms', r['kernel_ms'], 'fwd avg us', r['avg_launch_us'], d['config']['solved_fraction'], d['config']['mean_iterations'])
" | tee -a gpurun_out/c4.log
done
