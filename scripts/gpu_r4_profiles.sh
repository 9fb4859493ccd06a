# Round 4: targeted tests of what changed since the last full run + the rocprofv3 passes of the four GPU configs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_user_discrete.py tests/test_user_model_gpu.py tests/test_user_types_gpu.py tests/test_knot_times_gpu.py tests/test_bench_launch.py tests/test_config3_shards_gpu.py -q -m gpu 2>&1 | grep -v "^E  " | tail -25 | tee gpurun_out/r4_pytest_subset.log
for c in 2 3 4 1; do
  timeout 900 bash scripts/gpu_profile.sh $c > gpurun_out/r4_profile_c$c.log 2>&1
  tail -4 gpurun_out/r4_profile_c$c.log
done
