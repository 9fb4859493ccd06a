import ctypes, importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
lib = ctypes.CDLL(os.path.join(g.ROOT, "oracle/_build/liboracle.so"))
om = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d, _lib=lib, _prefix="oracle_")
hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
B = 64
res = {}
for name, mk, dt in (("o64", om, A.F64), ("o32", om, A.F32), ("g64", hm, A.F64), ("g32", hm, A.F32)):
    s = P.batch_three_obstacles(mk, batch=B, dtype=dt)
    s.solve()
    st = s.get_stats()
    res[name] = st
    print(name, "status hist", np.bincount(st["status"], minlength=10), "iters mean", st["iterations_total"].mean(), "max", st["iterations_total"].max())
for i in range(12):
    print(i, [(int(res[k]["status"][i]), int(res[k]["iterations_total"][i]), int(res[k]["iterations_outer"][i]), float(res[k]["violation"][i])) for k in ("o64", "o32", "g64", "g32")])
