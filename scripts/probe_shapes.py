#!/usr/bin/env python3
"""Stress of unusual shapes against the oracle (round 6, after the N > 126 restart bug): long horizons, odd batch sizes, fp32
records, Cholesky restarts on part of the batch.  Prints one line per case; exit code 1 if any schedule differs.

usage (GPU box): python scripts/probe_shapes.py"""
import ctypes
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
omake = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d, _lib=lib, _prefix="oracle_")
hmake = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)


def restart_mix(make, N, B, dtype, inner):
    s = make(3, 2, N, B, dtype)
    s.set_model(A.MODEL_UNICYCLE)
    s.set_uniform_step(np.float32(0.05))
    xf = np.tile(np.array([1.0, 0.5, 0.3]), (B, 1)) + np.linspace(0, 0.3, B)[:, None]
    R = np.diag([-2e-3, 1e-3])
    s.set_lqr_cost(0, N, np.eye(3) * 1e-3, R, xf, np.zeros(2))
    s.set_lqr_cost(N, N + 1, np.eye(3) * 10.0, R * 0, xf, np.zeros(2))
    s.add_control_bound(0, N, [-0.1, -0.1], [0.1, 0.1])
    s.set_initial_state(np.zeros(3))
    U = np.zeros((B, N, 2))
    U[0::3] = 0.05
    U[1::3] = 0.5
    U[2::3] = 0.08
    s.set_trajectory(None, U)
    s.set_options(max_iterations_inner=inner, max_iterations_outer=1)
    return s


def compare(name, o, g_, tight=True):
    o.solve(); g_.solve()
    so, sg = o.get_stats(), g_.get_stats()
    bad = int((so["iterations_total"] != sg["iterations_total"]).sum() + (so["status"] != sg["status"]).sum())
    Xo, _ = o.get_trajectory(); Xg, _ = g_.get_trajectory()
    ok = so["status"] == 0 if tight else np.ones(len(so), bool)
    dx = float(np.abs(Xo[ok] - Xg[ok]).max()) if ok.any() else 0.0
    Ko, _ = o.get_gains(); Kg, _ = g_.get_gains()
    dk = max((np.linalg.norm(Kg[b] - Ko[b]) / max(np.linalg.norm(Ko[b]), 1e-300) for b in np.flatnonzero(ok)), default=0.0)
    tm = g_.get_timing()
    print(f"{name:58s} schedule differences {bad:4d}  max|dX| {dx:8.1e}  max rel dK {dk:8.1e}  sweeps {tm['sweeps']:4d} fused {tm['fused_sweeps']:4d}", flush=True)
    return bad


def main():
    bad = 0
    for N in (127, 130, 200, 253):
        for B in (7, 640, 1100):
            for inner in (1, 3):
                bad += compare(f"restart mix N={N} B={B} inner={inner} f64", restart_mix(omake, N, B, A.F64, inner), restart_mix(hmake, N, B, A.F64, inner), tight=False) if inner == 1 else 0
    for N, B, dt in ((127, 700, A.F64), (200, 600, A.F64), (160, 900, A.F32), (300, 520, A.F64)):
        o = P.batch_three_obstacles(omake, batch=B, N=N, dtype=A.F64 if dt == A.F64 else 2)
        h = P.batch_three_obstacles(hmake, batch=B, N=N, dtype=dt)
        bad += compare(f"obstacles N={N} B={B} {'f64' if dt == A.F64 else 'f32 records'}", o, h)
    for N, B in ((150, 1300), (255, 530)):
        bad += compare(f"turn90 N={N} B={B} f64", P.batch_turn90(omake, batch=B, N=N), P.batch_turn90(hmake, batch=B, N=N))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
