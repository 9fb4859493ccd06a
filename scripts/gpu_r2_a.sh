# Round 2, GPU visit A: full GPU test suite, the 16x16x4 fp64 MFMA probe, one bench line per BASELINE config,
# kernel-trace stats of configs 3 and 4.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
O=gpurun_out/r2a
nproc > $O/nproc.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "^E  |^FAILED|^ERROR|passed|failed|solved|mismatch" | cut -c1-400 | head -80 | tee $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.log
./scripts/probes/mfma_f64_16x16_probe 2>&1 | tee $O/mfma16_probe.txt
for c in 2 1 3 4; do
  timeout 600 python bench.py --config $c --steps 3 --warmup 1 2>&1 | tail -1 | tee $O/bench_config$c.json
done
for c in 3 4; do
  rm -rf $O/trace$c
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$c -o bench -- python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $O/trace${c}_run.log 2>&1
  cp $O/trace$c/*/*kernel_stats.csv $O/kernel_stats_config$c.csv 2>/dev/null || cp $O/trace$c/*kernel_stats.csv $O/kernel_stats_config$c.csv 2>/dev/null
  rm -rf $O/trace$c
  head -12 $O/kernel_stats_config$c.csv | cut -c1-200
done
