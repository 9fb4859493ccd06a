"""CPU-side study behind the design of the ALTRO_F32 engine (runs on the oracle only; no GPU needed).

Solves the same seeded batch of BASELINE config 4 (kThreeObstacles, jittered obstacles) and config 5
(12-state model) with the oracle in three arithmetic modes:
  f64      the reference restatement in fp64
  f32      the same restatement with every scalar a float (what an all-fp32 port would compute)
  f64rec32 fp64 arithmetic, expansion and gain records rounded to fp32 (the product's ALTRO_F32 engine)
and prints solved fractions, iteration-count differences and state errors against the fp64 run.
usage: python scripts/cpu_fp32_study.py [batch4] [batch5]
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

A = graft.load_package()
import importlib  # noqa: E402

P = importlib.import_module("altro_cpp_amd.problems")
lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d, _lib=lib, _prefix="oracle_")  # noqa: E731
cores = len(os.sched_getaffinity(0))


def run(factory, batch, dtype, **kw):
    s = factory(make, batch=batch, dtype=dtype, **kw)
    lib.oracle_set_threads(s._h, ctypes.c_int(cores))
    s.solve()
    return s.get_stats(), s.get_trajectory()[0], s.get_gains()[0]


def report(name, factory, batch):
    ref, Xr, Kr = run(factory, batch, 0)
    print(f"== {name}: batch {batch}; fp64 solved {np.mean(ref['status'] == 0):.4f}, mean iterations "
          f"{ref['iterations_total'].mean():.1f}")
    for label, dt in (("f32", 1), ("f64rec32", 2)):
        st, X, K = run(factory, batch, dt)
        both = (ref["status"] == 0) & (st["status"] == 0)
        dit = st["iterations_total"].astype(int) - ref["iterations_total"].astype(int)
        err = np.abs(X - Xr).max(axis=(1, 2)) / np.maximum(1.0, np.abs(Xr).max(axis=(1, 2)))
        kerr = np.abs(K - Kr).max(axis=(1, 2, 3)) / np.maximum(1e-12, np.abs(Kr).max(axis=(1, 2, 3)))
        print(f"  {label:9s} solved {np.mean(st['status'] == 0):.4f}  solved-by-both {both.mean():.4f}  "
              f"same status {np.mean(st['status'] == ref['status']):.4f}")
        if both.any():
            d = dit[both]
            print(f"            |dIter|<=2: {np.mean(np.abs(d) <= 2):.4f}  ==0: {np.mean(d == 0):.4f}  "
                  f"max |dIter| {np.abs(d).max()}  state err/scale: median {np.median(err[both]):.2e} "
                  f"p99 {np.quantile(err[both], 0.99):.2e} max {err[both].max():.2e}  <=1e-3: {np.mean(err[both] <= 1e-3):.4f}  "
                  f"gain rel err: median {np.median(kerr[both]):.2e} <=1e-2: {np.mean(kerr[both] <= 1e-2):.4f}")


if __name__ == "__main__":
    b4 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    b5 = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    report("config 4 (three obstacles)", P.batch_three_obstacles, b4)
    report("config 3 (turn90)", P.batch_turn90, b4)
    if b5 > 0:
        report("config 5 (12-state model)", P.batch_quadrotor12, b5)
