"""Round 5: wall time of consecutive solves with twin workgroups (one process), per repetition."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
for name, fac in (("turn90_4096", lambda: P.batch_turn90(make, batch=4096, seed=P.SEED_BASE + 3)),
                  ("turn90_512", lambda: P.batch_turn90(make, batch=512, seed=P.SEED_BASE + 3))):
    s = fac()
    row = []
    for rep in range(8):
        s.reset_trajectory()
        t0 = time.perf_counter(); s.solve(); dt = time.perf_counter() - t0
        tm = s.get_timing()
        row.append((round(1e3 * dt, 2), round(tm["total_ms"], 2), tm["twin_claims"], tm["twin_handovers"]))
    print(name, os.environ.get("ALTRO_HIP_TWIN", "1"), row, flush=True)
    s.close()
