#!/bin/bash
# round 3: k_al_init over a (instances, row chunks) grid: parity + warm-start tests, latency of small batches
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_options_gpu.py tests/test_user_types_gpu.py tests/test_fused_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/alinit_tests.log
bash scripts/gpu_latency_trace.sh 1 | head -12
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(d['ms_per_step'], d['value'], json.dumps(d['latency']['kTurn90']))
" | tee gpurun_out/alinit_latency.log
