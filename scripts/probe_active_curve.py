"""Active-instance curve of the bench workloads (configs 2 and 3) from the CPU oracle: how many instances still iterate
after each sweep, and what a hand-over to the persistent kernel at 1x / 2x / 4x / 8x the CUs would leave.
    python scripts/probe_active_curve.py"""
import ctypes, importlib, sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
A = bench.graft.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
lib = bench.load_oracle()
omake = lambda n_, m_, N_, b_, d_: A.BatchSolver(n_, m_, N_, b_, d_, _lib=lib, _prefix="oracle_")
for ci in (2, 3):
    cfg = bench.CONFIGS[ci]
    o = getattr(P, cfg["factory"])(omake, batch=4096, dtype=A.F64 if ci == 2 else 2, seed=P.SEED_BASE + cfg["seed"])
    lib.oracle_set_threads(o._h, 8)
    o.solve()
    it = o.get_stats()["iterations_total"]
    act = [(it > s).sum() for s in range(0, it.max() + 1)]
    print("config", ci, "sum iters", it.sum(), "max", it.max())
    print(" active after sweep s:", [(s, int(a)) for s, a in enumerate(act) if s < 20 or s % 10 == 0])
    for thr in (256, 512, 1024, 2048):
        s0 = next(s for s, a in enumerate(act) if a <= thr)
        rem = sum(max(0, int(x) - s0) for x in it)
        print(f"  handover at <= {thr}: sweep {s0}, active {act[s0]}, remaining instance-iterations {rem}, longest {it.max() - s0}")
