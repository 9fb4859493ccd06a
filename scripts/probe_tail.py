import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
s = P.batch_turn90(lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d), batch=8)
s.solve(); s.reset_trajectory(); s.solve()
print(s.get_stats()[["status", "iterations_total", "alpha", "regularization"]])
