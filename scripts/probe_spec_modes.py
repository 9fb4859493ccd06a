"""Wall time of a small-batch AL solve under the speculation modes of the persistent kernel (ALTRO_HIP_SPECULATION)."""
import importlib, os, subprocess, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, root)
    import __graft_entry__ as g
    A = g.load_package()
    P = importlib.import_module("altro_cpp_amd.problems")
    hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
    B = int(sys.argv[2])
    for name, fac, kw in (("turn90", P.batch_turn90, {}), ("obstacles", P.batch_three_obstacles, {"dtype": A.F64})):
        s = fac(hm, batch=B, **kw)
        s.solve()
        best = 1e9
        for _ in range(5):
            s.reset_trajectory()
            t0 = time.perf_counter(); s.solve(); best = min(best, time.perf_counter() - t0)
        print(f"  B={B:4d} {name:9s} {1e3 * best:7.3f} ms  (longest chain {s.get_timing()['sweeps']})", flush=True)
    sys.exit(0)
for mode in ("off", "wave", "free", "helper"):
    print("ALTRO_HIP_SPECULATION=" + mode, flush=True)
    for B in (1, 32, 128, 250):
        subprocess.run([sys.executable, __file__, "child", str(B)], check=True, env=dict(os.environ, ALTRO_HIP_SPECULATION=mode))
