"""Several large handles in one process: the chains of sweeps are a per-device resource (four streams
that must land on hardware queues of their own), booked by the first large engine of a device and released with it; a
second large handle takes the one-chain path -- same bits, reported through altro_get_timing."""
import numpy as np
import pytest


def _chains(s):
    tm = s.get_timing()
    return round(tm["sweep_launches"] / max(1, tm["sweeps"] - tm["fused_sweeps"]))


def _snapshot(s):
    st = s.get_stats()
    X, U = s.get_trajectory()
    return st, X, U


def _same(a, b):
    for f in a[0].dtype.names:
        assert np.array_equal(a[0][f], b[0][f]), f
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.gpu
def test_two_large_handles_share_a_device(A, P, hip_make, monkeypatch):
    monkeypatch.delenv("ALTRO_HIP_CHAINS", raising=False)
    B = 4096
    s1 = P.batch_turn90(hip_make, batch=B)
    s2 = P.batch_turn90(hip_make, batch=B)
    s1.solve()
    ref = _snapshot(s1)
    s2.solve()
    assert _chains(s1) == 4 and _chains(s2) == 1  # the second handle's fall-back is visible in its timing
    _same(ref, _snapshot(s2))  # ... and changes no bit
    # both in flight at once (altro_solve_al_async: bench.py --pipeline 2)
    for s in (s1, s2):
        s.reset_trajectory()
    s1.solve_async(); s2.solve_async()
    s1.wait(); s2.wait()
    _same(ref, _snapshot(s1))
    _same(ref, _snapshot(s2))
    # the book is per device and released with the handle: the next large handle gets the chains again
    s1.close()
    s3 = P.batch_turn90(hip_make, batch=B)
    s3.solve()
    assert _chains(s3) == 4
    _same(ref, _snapshot(s3))
    s2.reset_trajectory(); s2.solve()
    assert _chains(s2) == 1
    _same(ref, _snapshot(s2))


@pytest.mark.gpu
def test_plugin_engines_book_in_the_same_ledger(A, P, hip_make, monkeypatch):
    """A user-model plugin carries its own copy of the engine code: the library hands it its counter at load time, so a
    large plugin engine beside a large built-in engine takes the one-chain path too."""
    import os
    monkeypatch.delenv("ALTRO_HIP_CHAINS", raising=False)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kind = A.register_model_source("cartpole", open(os.path.join(root, "tests", "models", "cartpole.hpp")).read())
    big = P.batch_turn90(hip_make, batch=2048)
    big.solve()
    assert _chains(big) == 4
    cp = P.cartpole_move(hip_make, kind, batch=2048, goal=np.linspace(0.4, 1.6, 2048))
    cp.solve()
    assert _chains(cp) == 1
    big.close()
    cp2 = P.cartpole_move(hip_make, kind, batch=2048, goal=np.linspace(0.4, 1.6, 2048))
    cp2.solve()
    assert _chains(cp2) == 4
    ok = cp.get_stats()["status"] == 0
    assert np.array_equal(cp.get_stats()["iterations_total"], cp2.get_stats()["iterations_total"])
    assert np.array_equal(cp.get_trajectory()[0][ok], cp2.get_trajectory()[0][ok])
