# round 3, final state: whole GPU suite, smoke(), the default bench line (what the driver runs)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^E  " | tail -12 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/smoke.log
timeout 900 python bench.py 2>&1 | tail -3 | tee gpurun_out/bench_default.log
