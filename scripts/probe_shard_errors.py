import sys, ctypes, importlib, os, numpy as np
sys.path.insert(0,'.')
import __graft_entry__ as g
A=g.load_package(); P=importlib.import_module('altro_cpp_amd.problems'); S=importlib.import_module('altro_cpp_amd.sharding')
lib=ctypes.CDLL('oracle/_build/liboracle.so')
mk=lambda n,m,N,b,d: A.BatchSolver(n,m,N,b,d,_lib=lib,_prefix='oracle_')
hm=lambda n,m,N,b,d: A.BatchSolver(n,m,N,b,d)
for r in (6,7):
    sh=S.shard_range(32768,8,r)
    o=P.batch_three_obstacles(mk,batch=32768,dtype=2,shard=sh); lib.oracle_set_threads(o._h, ctypes.c_int(len(os.sched_getaffinity(0)))); o.solve()
    gg=P.batch_three_obstacles(hm,batch=32768,dtype=A.F32,shard=sh); gg.solve()
    so,sg=o.get_stats(),gg.get_stats()
    same=(so['status']==sg['status'])&(so['iterations_total']==sg['iterations_total'])&(so['iterations_outer']==sg['iterations_outer'])
    ok=same&(so['status']==0)
    Xo,_=o.get_trajectory(); Xg,_=gg.get_trajectory()
    err=np.abs(Xg[ok]-Xo[ok]).max(axis=(1,2))
    idx=np.flatnonzero(ok)[np.argsort(err)[-5:]]
    print('shard',r,'ok',ok.sum(),'err>1e-5:',(err>1e-5).sum(),'max',err.max(),'top',np.sort(err)[-5:], 'iters', so['iterations_total'][idx], 'cost rel', np.abs(sg['cost'][idx]-so['cost'][idx])/so['cost'][idx])
