"""Per-kernel HIP-event timing of one config-3 solve (profiler_enable) + wall time of plain solves."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
s = P.batch_turn90(lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d), batch=4096, seed=P.SEED_BASE + 3)
s.solve()
best = 1e9
for rep in range(5):
    s.reset_trajectory()
    t0 = time.perf_counter(); s.solve(); best = min(best, time.perf_counter() - t0)
s.set_options(profiler_enable=1)
s.reset_trajectory(); s.solve()
tm = s.get_timing()
print("solve ms %.3f" % (best * 1e3), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()})
if tm["fused_sweeps"]:
    print("fused us/sweep %.1f" % (1e3 * tm["fused_ms"] / tm["fused_sweeps"]))
