# round 3: the persistent kernel with software-synchronised forward waves (ALTRO_HIP_SPECULATION=free) against the lock step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fused_gpu.py -q -m gpu -x -k "free or wave or NO_SPEC" 2>&1 | grep -v "^E  " | tail -15 | tee gpurun_out/pytest_free.log
for rep in 1 2; do
for c in 2 3; do
  for mode in wave free; do
    ALTRO_HIP_SPECULATION=$mode timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --no-latency 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; print('$mode', 'config', $c, d['ms_per_step'], 'fused_ms', r['kernel_ms']['sweep_fused'], 'tail_iter_us', r.get('tail_iteration_us'), 'tail_iters', r['tail_iterations'])
" | tee -a gpurun_out/free_ab.log
  done
done
done
timeout 600 python scripts/probe_spec_modes.py 2>&1 | tee gpurun_out/spec_modes.log
