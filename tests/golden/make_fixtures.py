#!/usr/bin/env python3
"""Generates tests/golden/fixtures.npz with the CPU oracle (oracle/altro_oracle.cpp).

The reference itself cannot be built in this image (Eigen/fmt absent), so the fixtures are outputs of
the oracle, which is pinned to the reference's known-answer tests by
tests/test_oracle_reference_constants.py.  Small batches (<= 8 instances) of BASELINE configs 1-5 with
the seeded synthetic inputs of altro-cpp_amd/problems.py; instance 0 is always the reference problem.

    python tests/golden/make_fixtures.py
"""
import ctypes
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

CASES = {
    # name: (factory name, kwargs, solve mode[, kwargs the HIP product takes instead])
    "c1_three_obstacles": ("unicycle_three_obstacles", dict(batch=1), "al"),
    "c2_triple_integrator": ("batch_triple_integrator", dict(batch=8), "ilqr"),
    "c3_turn90": ("batch_turn90", dict(batch=8), "al"),
    "c4_three_obstacles_batch": ("batch_three_obstacles", dict(batch=4, dtype=0), "al"),
    "c5_quadrotor12": ("batch_quadrotor12", dict(batch=4, dtype=0), "al"),
    # ALTRO_F32 (the dtype BASELINE configs[3] and [4] name): fp64 arithmetic with fp32 expansion / gain records.  The
    # fixture is the RECORD-ROUNDING oracle (oracle dtype 2); the product is created with ALTRO_F32 (dtype 1).
    "c4_three_obstacles_f32": ("batch_three_obstacles", dict(batch=4, dtype=2), "al", dict(dtype=1)),
    "c5_quadrotor12_f32": ("batch_quadrotor12", dict(batch=4, dtype=2), "al", dict(dtype=1)),
}


def solve_case(P, make, name, product=False):
    fac, kw, mode = CASES[name][:3]
    if product and len(CASES[name]) > 3:
        kw = dict(kw, **CASES[name][3])
    s = getattr(P, fac)(make, **kw)
    (s.solve if mode == "al" else s.solve_ilqr)()
    st = s.get_stats()
    X, U = s.get_trajectory()
    K, d = s.get_gains()
    out = {"X": X, "U": U, "K": K, "d": d, "status": st["status"], "iterations_total": st["iterations_total"],
           "iterations_outer": st["iterations_outer"], "cost": st["cost"], "violation": st["violation"],
           "alpha": st["alpha"], "max_penalty": st["max_penalty"]}
    if s.num_constraints() > 0:
        out["duals"] = s.get_duals()
    return out


def main():
    A = g.load_package()
    P = importlib.import_module("altro_cpp_amd.problems")
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
    make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d, _lib=lib, _prefix="oracle_")  # noqa: E731
    blob = {}
    for name in CASES:
        for k, v in solve_case(P, make, name).items():
            blob[f"{name}/{k}"] = v
        print(name, "iterations", blob[f"{name}/iterations_total"], "status", blob[f"{name}/status"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fixtures.npz"), **blob)


if __name__ == "__main__":
    main()
