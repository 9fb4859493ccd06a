"""User-defined dynamics: a model the library has never seen -- a cart-pole,
tests/models/cartpole.hpp -- registered as SOURCE through altro_register_model_source, compiled by the library into
a plugin that carries the whole solver for it, checked by the device-side FunctionBase::CheckJacobian, and solved;
parity against the oracle built with the very same source compiled for the host (oracle/Makefile)."""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CARTPOLE = open(os.path.join(ROOT, "tests", "models", "cartpole.hpp")).read()


@pytest.fixture(scope="module")
def cartpole_oracle(A):
    path = os.path.join(ROOT, "oracle", "_build", "liboracle_cartpole.so")
    if not os.path.exists(path):
        import __graft_entry__ as graft
        graft.build_oracle()
    lib = ctypes.CDLL(path)
    return lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d, _lib=lib, _prefix="oracle_")


def test_registration_compiles_loads_and_caches(A):
    """No GPU needed: hipcc cross-compiles the plugin, the library loads it, the second registration is a cache hit;
    the Jacobian check is deferred to the first handle when no device is present."""
    import time
    os.environ.setdefault("ALTRO_HIP_ARCH", "gfx950")
    k1 = A.register_model_source("cartpole", CARTPOLE)
    t0 = time.perf_counter()
    k2 = A.register_model_source("cartpole", CARTPOLE)
    assert k1 == k2 >= A.MODEL_USER_BASE and time.perf_counter() - t0 < 0.5
    # a model with the wrong dimensions for the handle is refused before anything is launched
    s = A.BatchSolver(3, 2, 10, 1, A.F64)
    s.set_model(k1)
    with pytest.raises(A.AltroError, match="dimensions do not match the user model"):
        s.rollout()  # the first compute call creates the device state


def test_source_that_does_not_compile_is_reported(A):
    with pytest.raises(A.AltroError, match="compiling the user model 'broken' failed"):
        A.register_model_source("broken", "struct UserModel { static constexpr int n = 2, m = 1; this is not C++ };")


@pytest.mark.gpu
def test_wrong_jacobian_is_rejected(A):
    """FunctionBase::CheckJacobian (functionbase.cpp:35-73) on the device: a sign error in one entry is caught."""
    bad = CARTPOLE.replace("J[2 + 4 * n] = T(1) / D;", "J[2 + 4 * n] = -T(1) / D;")
    assert bad != CARTPOLE
    with pytest.raises(A.AltroError, match="does not match finite differences"):
        A.register_model_source("cartpole_bad_jacobian", bad)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype_name", ["F64", "F32"])
def test_cartpole_solves_and_matches_the_oracle(A, P, hip_make, cartpole_oracle, dtype_name):
    kind = A.register_model_source("cartpole", CARTPOLE)  # runs the Jacobian check on the device
    B = 48
    goals = np.linspace(0.4, 1.6, B)
    dt = getattr(A, dtype_name)
    g = P.cartpole_move(hip_make, kind, batch=B, goal=goals, dtype=dt)
    o = P.cartpole_move(cartpole_oracle, kind, batch=B, goal=goals, dtype=A.F64 if dtype_name == "F64" else 2)
    g.solve(); o.solve()
    so, sg = o.get_stats(), g.get_stats()
    print("cart-pole", dtype_name, "iterations", np.unique(so["iterations_total"], return_counts=True), "solved", (so["status"] == 0).mean())
    for f in ("status", "iterations_total", "iterations_outer"):
        assert (so[f] == sg[f]).all(), (f, so[f], sg[f])
    assert (so["status"] == 0).mean() > 0.9
    ok = so["status"] == 0
    (Xo, Uo), (Xg, Ug) = o.get_trajectory(), g.get_trajectory()
    tol = 1e-7 if dtype_name == "F64" else 1e-5
    assert np.allclose(Xg[ok], Xo[ok], rtol=tol, atol=tol), np.abs(Xg[ok] - Xo[ok]).max()
    assert np.allclose(Ug[ok], Uo[ok], rtol=10 * tol, atol=10 * tol), np.abs(Ug[ok] - Uo[ok]).max()
    assert np.allclose(sg["cost"][ok], so["cost"][ok], rtol=1e-7)
    # the cart arrived and stands still, the force respected its bound
    assert (np.abs(Xg[ok][:, -1, 0] - goals[ok]) < 1e-3).all() and (np.abs(Xg[ok][:, -1, 2:]) < 1e-3).all()
    assert np.abs(Ug[ok]).max() <= 3.0 + 1e-3 and np.abs(Ug[ok]).max() > 2.9  # the bound is active
    # step level: the RK4 Jacobian the device built from the user's f / jac
    g2 = P.cartpole_move(hip_make, kind, batch=4, goal=goals[:4], dtype=A.F64)
    o2 = P.cartpole_move(cartpole_oracle, kind, batch=4, goal=goals[:4], dtype=A.F64)
    for s in (g2, o2):
        s.set_trajectory(None, np.full((60, 1), 0.7)); s.rollout(); s.update_expansions()
    for k in (0, 30, 59):
        eo, eg = o2.get_expansion(k), g2.get_expansion(k)
        assert np.allclose(eg["A"], eo["A"], rtol=1e-12, atol=1e-14) and np.allclose(eg["B"], eo["B"], rtol=1e-12, atol=1e-14)
