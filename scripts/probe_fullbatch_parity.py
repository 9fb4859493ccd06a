import importlib, sys, os, ctypes, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
A = g.load_package(); P = importlib.import_module("altro_cpp_amd.problems")
lib = ctypes.CDLL("oracle/_build/liboracle.so")
om = lambda n,m,N,b,d: A.BatchSolver(n,m,N,b,d,_lib=lib,_prefix="oracle_")
hm = lambda n,m,N,b,d: A.BatchSolver(n,m,N,b,d)
o = P.batch_turn90(om, batch=4096, seed=P.SEED_BASE+3); lib.oracle_set_threads(o._h, ctypes.c_int(200)); o.solve()
h = P.batch_turn90(hm, batch=4096, seed=P.SEED_BASE+3); h.solve()
so, sg = o.get_stats(), h.get_stats()
same = (so["iterations_total"]==sg["iterations_total"])&(so["status"]==sg["status"])
print("mismatching instances:", int((~same).sum()), "of 4096; stragglers:", int((so["status"]!=0).sum()))
Xo,_=o.get_trajectory(); Xg,_=h.get_trajectory()
ok=so["status"]==0
print("max |X_gpu - X_oracle| over solved:", np.abs(Xg[ok]-Xo[ok]).max(), " over matching stragglers:", np.abs(Xg[~ok & same]-Xo[~ok & same]).max())
