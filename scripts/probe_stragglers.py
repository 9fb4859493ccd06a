"""How the straggler instances of the bench workload spend their ~115 iterations (per-iteration history)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
s = P.batch_turn90(lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d), batch=512, seed=P.SEED_BASE + 3)
s.set_record_history(301)
s.solve()
st = s.get_stats()
bad = np.flatnonzero(st["status"] != 0)
print("stragglers:", len(bad), "of", len(st), "statuses", np.bincount(st["status"][bad], minlength=10))
for b in bad[:6]:
    dj = s.get_history(int(b), "cost_decrease")
    al = s.get_history(int(b), "alpha")
    reg = s.get_history(int(b), "regularization")
    rej = int((dj == 0).sum())
    print(f"instance {b}: {len(dj)} iterations, rejected line searches (dJ == 0): {rej}, alpha min {al[al > 0].min():.2e}, "
          f"median alpha {np.median(al):.3g}, max reg {reg.max():.2e}, outer {st['iterations_outer'][b]}")
b = int(bad[0])
for f in ("regularization", "cost", "gradient", "improvement_ratio", "alpha"):
    h = s.get_history(b, f)
    print(f, "last 6:", h[-6:], "distinct values in the last 90 iterations:", len(np.unique(h[-90:])))
