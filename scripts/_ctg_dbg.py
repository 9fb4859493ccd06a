import importlib, sys, numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
def first():
    h = P.batch_turn90(make, batch=4)
    h.set_record_history(64)
    h.solve()
    for inst in range(4):
        h.get_history(inst, "alpha")
if "--hist" in sys.argv:
    first()
g3 = P.batch_turn90(make, batch=2304, seed=P.SEED_BASE + 3)
g3.solve()
P3, p3 = g3.get_ctg()
g4 = P.batch_turn90(make, batch=2304, seed=P.SEED_BASE + 3)
g4.set_record_ctg(True)
g4.solve()
P4, p4 = g4.get_ctg()
s3, s4 = g3.get_stats(), g4.get_stats()
bad = np.flatnonzero((P3 != P4).reshape(2304, -1).any(axis=1) | (p3 != p4).reshape(2304, -1).any(axis=1))
keys = ("sweeps", "fused_sweeps", "sweep_launches", "twin_handovers", "twin_claims", "segment_columns", "loop_workgroups")
print("timing g3", {k: v for k, v in g3.get_timing().items() if k in keys})
print("timing g4", {k: v for k, v in g4.get_timing().items() if k in keys})
print("differing instances", len(bad), bad[:20])
for b in bad[:8]:
    print(b, "it", s3["iterations_total"][b], s4["iterations_total"][b], "status", s3["status"][b], s4["status"][b], "reg", s3["regularization"][b], s4["regularization"][b],
          "max dP", np.abs(P3[b] - P4[b]).max(), "knots differing", np.flatnonzero((P3[b] != P4[b]).reshape(P3.shape[1], -1).any(axis=1))[:5], "nan", np.isnan(P3[b]).any(), np.isnan(P4[b]).any())
print("stats identical", all(np.array_equal(s3[f], s4[f]) for f in s3.dtype.names), "X identical", np.array_equal(g3.get_trajectory()[0], g4.get_trajectory()[0]))
