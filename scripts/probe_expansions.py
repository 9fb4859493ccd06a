"""Time of one k_expansions launch on the obstacle batch (wall clock over repeated update_expansions calls)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
for name, fac, dt in (("obstacles r32", P.batch_three_obstacles, A.F32), ("turn90 f64", P.batch_turn90, A.F64)):
    s = fac(hm, batch=4096, dtype=dt)
    s.rollout(); s.update_expansions()
    t0 = time.perf_counter()
    for _ in range(200):
        s.update_expansions()
    s.cost()
    print(name, "update_expansions: %.1f us per call" % ((time.perf_counter() - t0) / 200 * 1e6), flush=True)
