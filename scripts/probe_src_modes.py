"""Bitwise comparison of the forward pass's input sources (ALTRO_HIP_FWD_SRC=lds vs default) on one batch."""
import importlib, os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, root)
    import __graft_entry__ as g
    A = g.load_package()
    P = importlib.import_module("altro_cpp_amd.problems")
    hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
    s = P.batch_three_obstacles(hm, batch=int(sys.argv[3]), dtype=A.F32 if sys.argv[4] == "f32" else A.F64)
    s.set_options(max_iterations_total=int(sys.argv[5]))
    s.solve()
    X, U = s.get_trajectory(); st = s.get_stats()
    np.savez(sys.argv[2], X=X, U=U, it=st["iterations_total"], status=st["status"], cost=st["cost"], K=s.get_gains()[0], lam=s.get_duals())
    sys.exit(0)
for dt in ("f64", "f32"):
    for iters in (1, 2, 3, 300):
        out = {}
        for tag, env in (("glb", {}), ("lds", {"ALTRO_HIP_FWD_SRC": "lds"})):
            f = f"/tmp/src_{tag}.npz"
            subprocess.run([sys.executable, __file__, "child", f, "1024", dt, str(iters)], check=True, env=dict(os.environ, ALTRO_HIP_NO_FUSED_SWEEP="1", **env))
            out[tag] = np.load(f)
        diffs = {k: int((out["glb"][k] != out["lds"][k]).sum()) for k in out["glb"].files}
        print(dt, "max_iterations_total", iters, "differing elements:", diffs, flush=True)
