"""Bitwise comparison of a solve with and without one environment switch: python scripts/probe_env_bitwise.py ENV=VAL"""
import importlib, os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, root)
    import __graft_entry__ as g
    A = g.load_package()
    P = importlib.import_module("altro_cpp_amd.problems")
    hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
    which, dt, B = sys.argv[3], sys.argv[4], int(sys.argv[5])
    fac = P.batch_three_obstacles if which == "obstacles" else P.batch_turn90
    s = fac(hm, batch=B, dtype=A.F32 if dt == "f32" else A.F64)
    s.solve()
    X, U = s.get_trajectory(); st = s.get_stats()
    np.savez(sys.argv[2], X=X, U=U, it=st["iterations_total"], status=st["status"], cost=st["cost"], K=s.get_gains()[0], d=s.get_gains()[1], lam=s.get_duals())
    sys.exit(0)
kv = sys.argv[1].split("=", 1)
for which, dt, B in (("turn90", "f64", 2000), ("obstacles", "f64", 1000), ("obstacles", "f32", 1000), ("turn90", "f64", 70)):
    out = {}
    for tag, env in (("default", {}), ("switch", {kv[0]: kv[1]})):
        f = f"/tmp/bw_{tag}.npz"
        subprocess.run([sys.executable, __file__, "child", f, which, dt, str(B)], check=True, env=dict(os.environ, **env))
        out[tag] = np.load(f)
    diffs = {k: int((out["default"][k] != out["switch"][k]).sum()) for k in out["default"].files}
    print(which, dt, B, "differing elements:", diffs, "solved", float((out["default"]["status"] == 0).mean()), flush=True)
