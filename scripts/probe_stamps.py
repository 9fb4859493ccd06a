"""Temporary: phase stamps of the last k_forward2 launch (ALTRO_X_STAMPS=1)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
which = sys.argv[1] if len(sys.argv) > 1 else "turn90"
for B in [int(x) for x in os.environ.get("PROBE_B", "3,768,1536,4096").split(",")]:
    s = (P.batch_turn90 if which == "turn90" else P.batch_three_obstacles)(hm, batch=B)
    s.set_options(profiler_enable=1)
    s.solve()
    t = s.get_timing()
    print(f"B={B} fwd per sweep {1e3*t['forward_pass_ms']/t['sweeps']:.1f} us sweeps {t['sweeps']}", flush=True)
