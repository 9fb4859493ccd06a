# round 3: tests of the handle / plugin changes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_handles_gpu.py tests/test_user_model_gpu.py tests/test_knot_times_gpu.py -q -m gpu 2>&1 | grep -v "^E  " | tail -40 | tee gpurun_out/pytest_misc.log
