# Round 2, GPU visit E: gains-from-global forward pass for the 12-state model.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
O=gpurun_out/r2e
timeout 1500 python -m pytest tests/test_backward_variants_gpu.py tests/test_parity_gpu.py tests/test_f32_gpu.py tests/test_golden_fixtures.py tests/test_options_gpu.py -q -m gpu -x 2>&1 | grep -E "^E  |^FAILED|^ERROR|passed|failed" | cut -c1-700 | head -40 | tee $O/pytest_gpu.log
for c in 4 1; do
  for e in "" "ALTRO_HIP_NO_KDG=1"; do
  env $e python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$e config', d['config']['baseline_config_index'], d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['avg_launch_us'])" | tee -a $O/bench.txt
  done
done
