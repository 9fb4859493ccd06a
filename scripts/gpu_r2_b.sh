# Round 2, GPU visit B: the failed tests of visit A again + hand-over point of the persistent tail kernel.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
O=gpurun_out/r2b
timeout 1500 python -m pytest tests/test_f32_gpu.py tests/test_parity_gpu.py -q -m gpu -s 2>&1 | grep -E "^E  |^FAILED|^ERROR|passed|failed|solved|mismatch|config 5" | cut -c1-600 | head -60 | tee $O/pytest_gpu.log
for p in 128 256 384 512 768 1024 1536; do
  echo "PERSIST_AT=$p" | tee -a $O/persist.txt
  ALTRO_HIP_PERSIST_AT=$p python bench.py --config 2 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['tail_iterations'])" | tee -a $O/persist.txt
done
for p in 256 512 1024 2048; do
  echo "config3 PERSIST_AT=$p" | tee -a $O/persist.txt
  ALTRO_HIP_PERSIST_AT=$p python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['tail_iterations'])" | tee -a $O/persist.txt
done
