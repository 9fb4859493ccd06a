# Round 2, GPU visit D: the 16x16x4 MFMA backward pass -- variant tests, configs 1 / 4 parity, bench lines.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
O=gpurun_out/r2d
timeout 1500 python -m pytest tests/test_backward_variants_gpu.py tests/test_parity_gpu.py tests/test_f32_gpu.py tests/test_golden_fixtures.py -q -m gpu -s -x 2>&1 | grep -E "^E  |^FAILED|^ERROR|passed|failed|config 5" | cut -c1-700 | head -40 | tee $O/pytest_gpu.log
for c in 4 1; do
  python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config', d['config']['baseline_config_index'], d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['avg_launch_us'])" | tee -a $O/bench.txt
  ALTRO_HIP_BACKWARD=coop python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('coop config', d['config']['baseline_config_index'], d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['avg_launch_us'])" | tee -a $O/bench.txt
done
