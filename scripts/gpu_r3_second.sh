# round 3, second visit: per-knot steps / times tests first, the whole GPU suite, the CPU thread-scaling probe
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_knot_times_gpu.py tests/test_facade_gpu.py -q -m gpu -x 2>&1 | grep -v "^E  " | tail -30 | tee gpurun_out/pytest_knot.log
python -m pytest tests -q -m gpu 2>&1 | grep -v "^E  " | tail -15 | tee gpurun_out/pytest_gpu.log
python scripts/probe_cpu_scaling.py 4096 2>&1 | tee gpurun_out/cpu_scaling.log
