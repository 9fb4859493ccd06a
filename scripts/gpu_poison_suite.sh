# The parity tests with poisoned LDS / candidate buffers (ALTRO_HIP_DEBUG_POISON): a kernel that reads memory the solve has
# not written computes with the pattern and fails its parity test.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for pat in "7ff80000,mix" "ffffffff"; do
  ALTRO_HIP_DEBUG_POISON=$pat timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_f32_gpu.py tests/test_user_types_gpu.py tests/test_user_model_gpu.py tests/test_knot_times_gpu.py tests/test_options_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee -a gpurun_out/poison_suite.log
done
