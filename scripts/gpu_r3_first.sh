# round 3, first visit: GPU tests, then the default bench line (with other_configs / latency / the new cpu_baseline)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
lscpu | head -25 > gpurun_out/lscpu.txt
python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^E  " | tail -12 | tee gpurun_out/pytest_gpu.log
( time python bench.py ) > gpurun_out/bench_default.log 2>&1
tail -5 gpurun_out/bench_default.log
