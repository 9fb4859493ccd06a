"""Where the host-boundary step of bench.py spends its time (config 2): python scripts/probe_host_boundary.py"""
import importlib, os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
s = P.batch_turn90(hm, batch=4096, dtype=A.F64)
s.set_options(profiler_enable=0)
s.reset_trajectory()
X0, U0 = s.get_trajectory()
x0 = np.ascontiguousarray(X0[:, 0, :])
Xo = np.empty_like(X0); Uo = np.empty_like(U0)
import ctypes as C
for rep in range(4):
    t = [time.perf_counter()]
    s.set_initial_state(x0); t.append(time.perf_counter())
    s.set_trajectory(None, U0); t.append(time.perf_counter())
    s.solve(); t.append(time.perf_counter())
    X, U = s.get_trajectory(); t.append(time.perf_counter())
    s._call("get_trajectory", Xo.ctypes.data_as(C.POINTER(C.c_double)), Uo.ctypes.data_as(C.POINTER(C.c_double))); t.append(time.perf_counter())
    st = s.get_stats(); t.append(time.perf_counter())
    print("ms: set_initial_state %.2f  set_trajectory %.2f  solve %.2f  get_trajectory (fresh arrays) %.2f  get_trajectory (touched arrays) %.2f  get_stats %.2f" %
          tuple(1e3 * (b - a) for a, b in zip(t[:-1], t[1:])), flush=True)
