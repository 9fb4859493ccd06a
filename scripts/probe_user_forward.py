"""Time of one solve of user models that need per-knot data (time-varying dynamics, a model per knot, per-knot steps of the
triple integrator): which forward kernel runs them (round 4: k_forward2 instead of the single-wave k_forward)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = lambda n: open(os.path.join(root, "tests", "models", n + ".hpp")).read()
B = 2048
goals = np.linspace(0.4, 1.5, B)
cases = []
kw = A.register_model_source("cartpole_wind", src("cartpole_wind"))
cases.append(("cartpole_wind (time-varying)", lambda: P.cartpole_move(make, kw, batch=B, goal=goals)))
ks = A.register_model_source("cartpole_steps", src("cartpole_steps"))
km = np.repeat([0, 1, 2], 20).astype(np.int32)
cases.append(("cartpole_steps (model per knot)", lambda: P.cartpole_steps(make, ks, km, batch=B, goal=goals)))
kc = A.register_model_source("cartpole", src("cartpole"))
cases.append(("cartpole (uniform, reference point)", lambda: P.cartpole_move(make, kc, batch=B, goal=goals)))
for name, build in cases:
    s = build()
    s.set_options(profiler_enable=1)
    best = 1e9
    for _ in range(4):
        s.reset_trajectory()
        t0 = time.perf_counter(); s.solve(); best = min(best, time.perf_counter() - t0)
    tm = s.get_timing()
    st = s.get_stats()
    print(f"{name}: {1e3 * best:.2f} ms per {B}-instance solve, sweeps {tm['sweeps']}, kernel sums E {tm['expansions_ms']:.2f} B {tm['backward_pass_ms']:.2f} "
          f"F {tm['forward_pass_ms']:.2f} ms, solved {np.mean(st['status'] == 0):.3f}")
