import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
A = g.load_package(); P = importlib.import_module("altro_cpp_amd.problems")
mk = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
for dt in (A.F32, A.F64):
    s = P.batch_quadrotor12(mk, batch=1024, dtype=dt)
    s.solve(); s.set_options(profiler_enable=1); s.reset_trajectory(); s.solve()
    print("quad12", "f32" if dt == A.F32 else "f64", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in s.get_timing().items()})
s = P.batch_triple_integrator(mk, batch=1024)
s.solve_ilqr(); s.set_options(profiler_enable=1); s.reset_trajectory(); s.solve_ilqr()
print("tripleint f64", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in s.get_timing().items()})
