"""Feasibility probe (round 4): the first 8 sweeps of configs 2 and 3 -- all 4096 instances iterating -- with 20 lanes per
instance (3 per wavefront) against an experimental build with 10 lanes (ALTRO_LS_LANES=10: 6 per wavefront, only trials
0 .. 9 evaluated: NOT the reference's line search, a timing experiment).  usage: ALTRO_HIP_LIB=<lib> python scripts/probe_lanes10.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
for name, fac, dt, seed in (("config2", P.batch_turn90, A.F64, 3), ("config3", P.batch_three_obstacles, A.F32, 4)):
    s = fac(make, batch=4096, dtype=dt, seed=P.SEED_BASE + seed)
    s.set_options(max_iterations_total=8, max_iterations_inner=8, profiler_enable=1)
    for rep in range(3):
        s.reset_trajectory()
        t0 = time.perf_counter()
        s.solve()
        wall = time.perf_counter() - t0
    tm = s.get_timing()
    print(os.path.basename(os.environ.get("ALTRO_HIP_LIB", "default")), name, "8 sweeps: wall %.3f ms" % (1e3 * wall),
          "kernel sums E %.3f B %.3f F %.3f ms" % (tm["expansions_ms"], tm["backward_pass_ms"], tm["forward_pass_ms"]),
          "sweeps", tm["sweeps"], "launches", tm["sweep_launches"], "ls", s.get_options().line_search_max_iterations)
