#!/bin/bash
# round 3: structural zeros in rk4_jacobian (12-state / 6-state / user models): parity tests of those models, bench lines
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_f32_gpu.py tests/test_backward_variants_gpu.py tests/test_user_model_gpu.py tests/test_knot_times_gpu.py tests/test_golden_fixtures.py -x -q -m gpu > gpurun_out/sz_tests.log 2>&1
echo "exit $?" >> gpurun_out/sz_tests.log
tail -8 gpurun_out/sz_tests.log
for c in 4 1 2 3; do
    timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --no-latency 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; print('config', $c, d['ms_per_step'], d['value'], 'kernel_ms', r['kernel_ms'], 'tail_iter_us', r.get('tail_iteration_us'))
" | tee -a gpurun_out/sz_bench.log
done
