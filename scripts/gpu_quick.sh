# Quick GPU check: full GPU test suite + a short bench line (no profiler).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-300 | head -30 | tee gpurun_out/pytest_gpu.log
python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_quick.log
