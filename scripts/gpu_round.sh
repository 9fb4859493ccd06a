# One GPU visit: parity tests, smoke, bench, rocprof kernel trace of the bench.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
python bench.py --steps 3 --warmup 1 2>&1 | tail -3 | tee gpurun_out/bench.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_run.log 2>&1
ls -R gpurun_out/prof | head -20
