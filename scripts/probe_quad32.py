import importlib, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
A = g.load_package(); P = importlib.import_module("altro_cpp_amd.problems")
mk = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
for env in ("", "1"):
    if env: os.environ["ALTRO_HIP_VALU_BACKWARD"] = "1"
    s = P.batch_quadrotor12(mk, batch=1024, dtype=A.F32)
    s.solve()
    st = s.get_stats()
    print("valu" if env else "coop", "status counts", np.bincount(st["status"], minlength=10), "iters hist", np.bincount(st["iterations_total"])[:14], "max it", st["iterations_total"].max())
