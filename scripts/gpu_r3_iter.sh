# round 3: one visit for a change to the persistent kernel -- bitwise tests of the fused path, A/B of the speculation modes
# on configs 2 and 3, the phase stamps of the debug build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_gpu.py -q -m gpu 2>&1 | grep -v "^E  " | tail -8 | tee gpurun_out/pytest_fused.log
for c in 2 3; do
  for mode in wave free; do
    ALTRO_HIP_SPECULATION=$mode timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --no-latency 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; print('$mode', 'config', $c, d['ms_per_step'], 'fused_ms', r['kernel_ms']['sweep_fused'], 'tail_iter_us', r.get('tail_iteration_us'), 'tail_iters', r['tail_iterations'])
" | tee -a gpurun_out/iter_ab.log
  done
done
bash scripts/gpu_stamps.sh wave free 2>&1 | grep -A7 "turn90\|^==" | grep -v "^--$" | head -40 | tee gpurun_out/stamps.log
