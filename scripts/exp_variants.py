"""Times the sweep kernels of config 3 for experimental builds of the library (altro-cpp_amd/csrc/_x/libx*.so)."""
import glob, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [A.LIB_PATH] + sorted(glob.glob(os.path.join(root, "altro-cpp_amd", "csrc", "_x", "libx*.so")))
for path in libs:
    lib = A.load_library(path)
    s = P.batch_turn90(lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d, _lib=lib), batch=4096, seed=P.SEED_BASE + 3)
    s.solve()
    best = None
    for rep in range(3):
        s.reset_trajectory()
        t0 = time.perf_counter(); s.solve(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    s.set_options(profiler_enable=1)
    s.reset_trajectory(); s.solve()
    tm = s.get_timing()
    st = s.get_stats()
    print(os.path.basename(path), "solve ms %.3f" % (best * 1e3), "exp %.2f bwd %.2f fwd %.2f sweeps %d" % (
        tm["expansions_ms"], tm["backward_pass_ms"], tm["forward_pass_ms"], tm["sweeps"]),
        "iters", int(st["iterations_total"].sum()), flush=True)
