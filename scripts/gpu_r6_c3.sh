cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for mode in 1 0; do
  echo "== ALTRO_HIP_SWEEP_LOOP=$mode obstacles32 4096"
  ALTRO_HIP_LOOP_LOG=1 ALTRO_HIP_SWEEP_LOOP=$mode timeout 600 python scripts/probe_loop.py --child obstacles32 4096 3 /tmp/x$mode.npz 2>&1 | grep -v "^$" | tail -3
done
python - <<'PY'
import numpy as np
a, b = np.load("/tmp/x1.npz"), np.load("/tmp/x0.npz")
bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
print("config 3 shard: loop vs sweeps+segments", "BIT-IDENTICAL" if not bad else "DIFFERENT " + str(bad))
PY
} 2>&1 | tee gpurun_out/r6_c3.log | cut -c1-900
