"""Per-knot cost of the serial kernels: time(N=200) - time(N=100) isolates the loop from fixed overheads."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
for B in [int(x) for x in os.environ.get("PROBE_B", "3,64,4096").split(",")]:
    res = {}
    for N in (100, 200):
        s = P.batch_turn90(hm, batch=B, N=N)
        s.set_options(profiler_enable=1, max_iterations_inner=6, max_iterations_outer=1)
        s.solve()
        s.reset_trajectory(); s.solve()
        t = s.get_timing()
        res[N] = (t["expansions_ms"] / t["sweeps"], t["backward_pass_ms"] / t["sweeps"], t["forward_pass_ms"] / t["sweeps"], t["sweeps"])
    print(f"B={B}: N=100 exp/bwd/fwd us = {[round(1e3*x,1) for x in res[100][:3]]} sweeps {res[100][3]};  N=200: {[round(1e3*x,1) for x in res[200][:3]]};"
          f"  per-knot us: bwd {(res[200][1]-res[100][1])*10:.3f} fwd {(res[200][2]-res[100][2])*10:.3f}; fixed us: bwd {1e3*(2*res[100][1]-res[200][1]):.1f} fwd {1e3*(2*res[100][2]-res[200][2]):.1f}")
