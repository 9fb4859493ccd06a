# round 6: A/B of library builds on one box, loop kernel with its phase log: scripts/gpu_r6_ab.sh <lib> [<lib> ...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for rep in 1 2; do
for lib in "$@"; do
  for spec in ${AB_SPECS:-"turn90 4096 3"}; do :; done
  echo "== $lib"
  ALTRO_HIP_LIB=$lib ALTRO_HIP_LOOP_LOG=1 ALTRO_HIP_SWEEP_LOOP=1 timeout 600 python scripts/probe_loop.py --child ${AB_KIND:-turn90} ${AB_BATCH:-4096} 3 /tmp/x.npz 2>&1 | grep -v "^$" | tail -4
done
done
} 2>&1 | tee gpurun_out/loop_ab.log | cut -c1-1200
