"""CPU checks of bench.py's cpu_baseline leg (the oracle's pinned thread team; test infrastructure timing itself)."""
import ctypes
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _oracle(A):
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
    lib.oracle_bench_seconds.restype = ctypes.c_double
    lib.oracle_bench_busy.restype = ctypes.c_double
    return lib, (lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d, _lib=lib, _prefix="oracle_"))


def test_bench_team_matches_plain_solves(A, P):
    """oracle_bench_al (pinned team, repetition-major tasks, instances prepared outside) leaves every instance in the
    state a plain oracle_solve_al leaves it: the timed baseline solves the workload it claims to solve."""
    lib, omake = _oracle(A)
    ref = P.batch_turn90(omake, batch=24, seed=P.SEED_BASE + 3)
    ref.solve()
    sr = ref.get_stats()
    Xr, Ur = ref.get_trajectory()
    for nt in (1, 3):
        o = P.batch_turn90(omake, batch=24, seed=P.SEED_BASE + 3)
        assert lib.oracle_prepare(o._h) == 0
        lib.oracle_set_threads(o._h, nt)
        assert lib.oracle_bench_al(o._h, 3) == 0
        assert lib.oracle_bench_seconds(o._h) > 0.0
        assert lib.oracle_bench_threads(o._h) == nt
        assert lib.oracle_bench_busy(o._h, 1) >= lib.oracle_bench_busy(o._h, 0) > 0.0
        so = o.get_stats()
        for f in ("status", "iterations_total", "iterations_outer", "cost"):
            assert (so[f] == sr[f]).all(), f
        Xo, Uo = o.get_trajectory()
        assert (Xo == Xr).all() and (Uo == Ur).all()
    # every 8th instance only: the others stay untouched
    o = P.batch_turn90(omake, batch=24, seed=P.SEED_BASE + 3)
    lib.oracle_prepare(o._h)
    lib.oracle_set_ilqr_mode(o._h, 0)
    assert lib.oracle_bench_subset(o._h, 8, 2) == 0
    so = o.get_stats()
    assert (so["iterations_total"][::8] == sr["iterations_total"][::8]).all()
    assert (np.delete(so["iterations_total"], np.arange(0, 24, 8)) == 0).all()


def test_host_topology_is_sane():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
    hw, phys = lib.oracle_host_threads(), lib.oracle_host_physical_cores()
    assert 1 <= phys <= hw == len(os.sched_getaffinity(0))


def test_cpu_baseline_block(A, P):
    """The block bench.py prints: fields the judge reads, and an efficiency that is a fraction."""
    bench = importlib.import_module("bench")
    cfg = bench.CONFIGS[2]
    r = bench.cpu_baseline(A, P, cfg, 64, P.SEED_BASE + cfg["seed"], budget_cpu_s=1.0)
    assert r["kind"] == "port" and r["value"] > 0 and r["cores"] in (r["physical_cores"], r["hardware_threads"])
    assert 0.05 < r["parallel_efficiency"] < 1.5
    assert r["single_thread_value"] > 0 and r["teams"]
