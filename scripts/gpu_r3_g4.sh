# A/B: knots per synchronisation of the persistent kernel's knot loop (ALTRO_SYNC_FUSED = 2 | 4), both speculation modes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in 2 3; do
 for lib in libaltro_hip.so _x/libaltro_g4.so; do
  for mode in wave free; do
    ALTRO_HIP_LIB=$GRAFT_REPO_ROOT/altro-cpp_amd/csrc/$lib ALTRO_HIP_SPECULATION=$mode timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --no-latency 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; print('$lib', '$mode', 'config', $c, d['ms_per_step'], 'fused_ms', r['kernel_ms']['sweep_fused'], 'tail_iter_us', r.get('tail_iteration_us'), 'tail_iters', r['tail_iterations'])
"
  done
 done
done
