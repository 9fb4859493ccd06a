# round 3: whole GPU suite, then bench lines of configs 2 and 3 with the default settings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^E  " | tail -12 | tee gpurun_out/pytest_gpu.log
for c in 2 3; do
    timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --no-latency 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; print('default config', $c, d['ms_per_step'], d['value'], 'fused_ms', r['kernel_ms']['sweep_fused'], 'tail_iter_us', r.get('tail_iteration_us'), 'tail_iters', r['tail_iterations'])
" | tee -a gpurun_out/full_ab.log
done
