# Round 5: the whole -m gpu suite, smoke, then the rocprofv3 passes of the four GPU configs (scripts/gpu_profile.sh) on the final build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r05_parity_errors.json
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^E  " | tail -30 | tee gpurun_out/r5_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r5_smoke.log
for c in 2 3 4 1; do
  timeout 900 bash scripts/gpu_profile.sh $c > gpurun_out/r5_profile_c$c.log 2>&1
  tail -3 gpurun_out/r5_profile_c$c.log
done
