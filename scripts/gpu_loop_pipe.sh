# The two-window pipeline of the device-side loop: bits against the sweeps, timings, phase log.  LOOP_SPECS: "<kind> <batch> <reps>" ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export ALTRO_HIP_SWEEP_LOOP=1 ALTRO_HIP_LOOP_LOG=1
{
IFS=';' read -ra SPECS <<< "${LOOP_SPECS:-turn90 700 2;turn90 4096 3}"
for spec in "${SPECS[@]}"; do
  timeout ${LOOP_TIMEOUT:-150} python scripts/probe_loop.py $spec 2>&1 | grep -v "^$" | tail -14 | cut -c1-700
  rc=${PIPESTATUS[0]}
  if [ "$rc" = "124" ]; then echo "TIMEOUT on $spec -- stopping"; break; fi
done
} 2>&1 | tee gpurun_out/loop_pipe.log
