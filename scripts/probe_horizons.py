"""History of one instance of the N = 1 problem, GPU vs oracle (debug aid of test_horizon_lengths_and_ragged_batches)."""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
lib = ctypes.CDLL(os.path.join(g.ROOT, "oracle", "_build", "liboracle.so"))
om = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d, _lib=lib, _prefix="oracle_")
hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
o = P.batch_turn90(om, batch=33, N=N); s = P.batch_turn90(hm, batch=33, N=N)
for x in (o, s):
    x.set_record_history(64); x.solve()
b = 4
np.set_printoptions(linewidth=250, precision=6)
for f in ("cost", "alpha", "gradient", "cost_decrease", "regularization", "violations", "max_penalty", "improvement_ratio"):
    ho, hg = o.get_history(b, f), s.get_history(b, f)
    print(f, "\n  oracle", ho, "\n  gpu   ", hg)
