# Quick GPU check: full GPU test suite (+ C++ facade tests and perf drivers through pytest) + short bench lines.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -q -m gpu 2>&1 | grep -E "^E  |^FAILED|^ERROR|passed|failed" | cut -c1-500 | head -60 | tee gpurun_out/pytest_gpu.log
for c in ${BENCH_CONFIGS:-2}; do
  python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_quick_config$c.log
done
