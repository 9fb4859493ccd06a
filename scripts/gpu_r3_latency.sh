# round 3: small-batch latency with and without the persistent kernel from the first sweep; bitwise check of the two
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for env in ALTRO_HIP_NO_FUSED_FIRST=1 X=0; do
  echo "== $env"
  env $env python - <<'PY'
import importlib, sys, os, time, hashlib
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
for name, fac, kw in (("turn90", P.batch_turn90, {}), ("obstacles", P.batch_three_obstacles, {"dtype": A.F64})):
    for B in (1, 8, 64, 200):
        s = fac(hm, batch=B, **kw)
        s.set_options(profiler_enable=0)
        s.solve()
        best = 1e9
        for _ in range(7):
            s.reset_trajectory()
            t0 = time.perf_counter(); s.solve(); best = min(best, time.perf_counter() - t0)
        st = s.get_stats(); X, U = s.get_trajectory(); K, d = s.get_gains()
        h = hashlib.sha1(X.tobytes() + U.tobytes() + K.tobytes() + s.get_duals().tobytes() + st["iterations_total"].tobytes() + st["cost"].tobytes()).hexdigest()[:12]
        print(f"  {name:9s} B={B:4d} {1e3 * best:7.3f} ms  iterations max {st['iterations_total'].max():3d}  state {h}", flush=True)
PY
done
