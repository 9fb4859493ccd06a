"""Do the chains of sweeps keep their four hardware queues when other handles were created (and destroyed) before?
python scripts/probe_chain_streams.py <fresh|after_closed|beside_alive|after_two_closed>"""
import importlib, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
mode = sys.argv[1] if len(sys.argv) > 1 else "fresh"
keep = []
def small():
    s = P.unicycle_three_obstacles(hm)
    s.solve()
    return s
if mode == "after_closed":
    small().close()
elif mode == "after_two_closed":
    small().close(); small().close()
elif mode == "beside_alive":
    keep.append(small())
s = P.batch_turn90(hm, batch=4096, dtype=A.F64)
s.set_options(profiler_enable=1)
for rep in range(3):
    s.reset_trajectory()
    t0 = time.perf_counter(); s.solve(); dt = time.perf_counter() - t0
    tm = s.get_timing()
    print(mode, "rep", rep, "ms %.3f" % (1e3 * dt), "E %.2f B %.2f F %.2f fused %.2f" % (tm["expansions_ms"], tm["backward_pass_ms"], tm["forward_pass_ms"], tm["fused_ms"]), "sweep launches", tm["sweep_launches"], flush=True)
