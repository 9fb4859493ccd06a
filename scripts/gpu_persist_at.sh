# Where the batched sweeps hand over to the persistent tail kernel: scripts/gpu_persist_at.sh [configs...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in ${@:-2 3}; do
  for at in ${PERSIST_AT_LIST:-256 512 768 1024 1536 2048}; do
    ALTRO_HIP_PERSIST_AT=$at timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --no-latency 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; print('persist_at', $at, 'config', $c, d['ms_per_step'], d['config']['sweeps'], d['config']['solved_fraction'], 'fused_ms', r['kernel_ms']['sweep_fused'], 'tail_iter_us', r.get('tail_iteration_us'), 'tail_iters', r['tail_iterations'])
" | tee -a gpurun_out/persist_at.log
  done
done
