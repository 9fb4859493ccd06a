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
TRACK = open(os.path.join(ROOT, "tests", "models", "cartpole_track.hpp")).read()  # + UserCost + UserConstraint


@pytest.fixture(scope="module")
def cartpole_oracle(A):
    path = os.path.join(ROOT, "oracle", "_build", "liboracle_cartpole.so")
    if not os.path.exists(path):
        import __graft_entry__ as graft
        graft.build_oracle()
    lib = ctypes.CDLL(path)
    return lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d, _lib=lib, _prefix="oracle_")


@pytest.fixture(scope="module")
def track_oracle(A):
    path = os.path.join(ROOT, "oracle", "_build", "liboracle_cartpole_track.so")
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
    tol = 1e-10  # (both engines against their own oracle -- fp64 / record-rounding: measured X 5e-14, U 5e-13, duals 1e-11 abs)
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


# ---- user-defined cost function and constraint (the other two plug-in classes of the reference) ------------------
def test_source_with_cost_and_constraint_compiles(A):
    """No GPU needed: the plugin of a source that also defines UserCost / UserConstraint cross-compiles and loads."""
    os.environ.setdefault("ALTRO_HIP_ARCH", "gfx950")
    assert A.register_model_source("cartpole_track", TRACK) >= A.MODEL_USER_BASE


def test_user_functors_on_the_oracle(A, P, track_oracle):
    """CPU: the oracle compiled with the same source solves the sway-limited move; the constraint is active."""
    goals = np.linspace(0.4, 1.4, 6)
    o = P.cartpole_track(track_oracle, A.MODEL_USER_BASE, batch=6, goal=goals)
    o.solve()
    st = o.get_stats()
    assert (st["status"] == 0).all(), st["status"]
    X, _ = o.get_trajectory()
    sway = np.abs(0.5 * np.sin(X[:, :, 1])).max(axis=1)
    assert sway[0] < 0.03 and (np.abs(sway[3:] - 0.04) < 2e-4).all(), sway  # inactive for the short move, active for the long ones


@pytest.mark.gpu
def test_user_cost_and_constraint_need_a_model_that_defines_them(A, P, hip_make):
    kind = A.register_model_source("cartpole", CARTPOLE)  # dynamics only
    s = hip_make(4, 1, 20, 2, A.F64)
    s.set_model(kind); s.set_uniform_step(np.float32(0.05))
    s.set_user_cost(0, 21, np.zeros(6))
    s.set_initial_state(np.zeros(4)); s.set_trajectory(None, np.zeros((20, 1)))
    with pytest.raises(A.AltroError, match="defines no UserCost"):
        s.rollout()
    s = hip_make(3, 2, 20, 2, A.F64)  # a built-in model
    s.set_model(A.MODEL_UNICYCLE); s.set_uniform_step(np.float32(0.05))
    s.set_lqr_cost(0, 21, np.eye(3), np.eye(2), np.zeros(3), np.zeros(2))
    s.add_user_constraint(0, 20, np.zeros(2))
    s.set_initial_state(np.zeros(3)); s.set_trajectory(None, np.zeros((20, 2)))
    with pytest.raises(A.AltroError, match="defines no UserConstraint"):
        s.rollout()


@pytest.mark.gpu
def test_wrong_cost_gradient_hessian_and_constraint_jacobian_are_rejected(A):
    """ScalarFunction::CheckGradient, FunctionBase::CheckHessian / CheckJacobian (functionbase.cpp:42-125) on the device."""
    cases = [("dx[1] = par[2] * sin(x[1]);", "dx[1] = -par[2] * sin(x[1]);", "UserCost::gradient"),
             ("dxdx[1 + 1 * 4] = par[2] * cos(x[1]);", "dxdx[1 + 1 * 4] = par[2] * sin(x[1]);", "UserCost::hessian"),
             ("J[1 + 1 * 2] = -dt;", "J[1 + 1 * 2] = dt;", "UserConstraint::jacobian")]
    for i, (good, bad, what) in enumerate(cases):
        src = TRACK.replace(good, bad)
        assert src != TRACK
        with pytest.raises(A.AltroError, match=what + r"\(\) does not match finite differences"):
            A.register_model_source(f"cartpole_track_bad{i}", src)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype_name", ["F64", "F32"])
def test_user_cost_and_constraint_match_the_oracle(A, P, hip_make, track_oracle, dtype_name):
    kind = A.register_model_source("cartpole_track", TRACK)  # runs the derivative checks on the device
    B = 40
    goals = np.linspace(0.4, 1.4, B)
    dt = getattr(A, dtype_name)
    g = P.cartpole_track(hip_make, kind, batch=B, goal=goals, dtype=dt)
    o = P.cartpole_track(track_oracle, kind, batch=B, goal=goals, dtype=A.F64 if dtype_name == "F64" else 2)
    # step level first: cost, expansion with the user's gradient / Hessian and the user's constraint terms
    g2 = P.cartpole_track(hip_make, kind, batch=4, goal=goals[-4:], dtype=A.F64)
    o2 = P.cartpole_track(track_oracle, kind, batch=4, goal=goals[-4:], dtype=A.F64)
    for s in (g2, o2):
        s.set_trajectory(None, np.full((60, 1), 2.5)); s.rollout(); s.set_penalty(7.0); s.update_expansions()
    assert np.allclose(g2.cost(), o2.cost(), rtol=1e-12)
    Xr, _ = o2.get_trajectory()
    assert np.abs(0.5 * np.sin(Xr[:, :, 1])).max() > 0.05  # the rollout violates the sway limit: the AL terms are live
    for k in (1, 30, 59, 60):
        eo, eg = o2.get_expansion(k), g2.get_expansion(k)
        for f in (("lx", "lu", "lxx", "lxu", "luu") if k < 60 else ("lx", "lxx")):  # (the terminal knot has no control)
            assert np.allclose(eg[f], eo[f], rtol=1e-11, atol=1e-12), (k, f, np.abs(eg[f] - eo[f]).max())
    assert np.allclose(g2.get_constraint_values(), o2.get_constraint_values(), rtol=1e-12, atol=1e-14)
    # the solves
    g.solve(); o.solve()
    so, sg = o.get_stats(), g.get_stats()
    print("cart-pole track", dtype_name, "iterations", np.unique(so["iterations_total"], return_counts=True), "solved", (so["status"] == 0).mean())
    for f in ("status", "iterations_total", "iterations_outer"):
        assert (so[f] == sg[f]).all(), (f, so[f], sg[f])
    assert (so["status"] == 0).mean() > 0.9
    ok = so["status"] == 0
    (Xo, Uo), (Xg, Ug) = o.get_trajectory(), g.get_trajectory()
    tol = 1e-10  # (both engines against their own oracle -- fp64 / record-rounding: measured X 5e-14, U 5e-13, duals 1e-11 abs)
    assert np.allclose(Xg[ok], Xo[ok], rtol=tol, atol=tol), np.abs(Xg[ok] - Xo[ok]).max()
    assert np.allclose(Ug[ok], Uo[ok], rtol=10 * tol, atol=10 * tol), np.abs(Ug[ok] - Uo[ok]).max()
    assert np.allclose(sg["cost"][ok], so["cost"][ok], rtol=1e-7)
    assert np.allclose(g.get_duals()[ok], o.get_duals()[ok], rtol=1e2 * tol, atol=1e2 * tol)
    # the sway limit holds (to the constraint tolerance) and is active for the long moves
    sway = np.abs(0.5 * np.sin(Xg[:, :, 1])).max(axis=1)
    assert (sway[ok] < 0.04 + 1e-3).all() and (sway[ok][-5:] > 0.04 - 1e-3).all(), sway


@pytest.mark.gpu
def test_plugin_is_compiled_on_this_machine_and_a_stale_cache_entry_is_rebuilt(A, P, hip_make, cartpole_oracle, tmp_path):
    """The run-time compile itself: a source with a nonce cannot be in any cache, so hipcc really
    runs HERE, for the architecture of the device that is present; the plugin it produces solves like the oracle.  Then
    the cache entry is overwritten with ANOTHER plugin under the same file name: the next process notices (the generated
    translation unit embeds its hash) and rebuilds instead of loading foreign code."""
    import shutil
    import subprocess
    import sys
    import time
    nonce = time.time_ns()
    script = f"""
import sys, time, os
sys.path.insert(0, {ROOT!r})
import __graft_entry__ as g
A = g.load_package()
src = open({os.path.join(ROOT, 'tests', 'models', 'cartpole.hpp')!r}).read() + "\\n// nonce {nonce}\\n"
t0 = time.perf_counter()
kind = A.register_model_source("cartpole nonce/\\"x", src)   # (a name that needs sanitising)
print("SECONDS", time.perf_counter() - t0)
print("PATH", A.user_model_path(kind))
import importlib, numpy as np
P = importlib.import_module("altro_cpp_amd.problems")
s = P.cartpole_move(lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d), kind, batch=6, goal=np.linspace(0.5, 1.5, 6))
s.solve()
print("ITERS", s.get_stats()["iterations_total"].tolist())
"""
    env = dict(os.environ, ALTRO_HIP_CACHE_DIR=str(tmp_path))
    env.pop("ALTRO_HIP_ARCH", None)  # the architecture of the device that is here

    def run():
        r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out = dict(l.split(" ", 1) for l in r.stdout.splitlines() if l.split(" ", 1)[0] in ("SECONDS", "PATH", "ITERS"))
        return float(out["SECONDS"]), out["PATH"], out["ITERS"]

    sec1, path1, iters1 = run()
    assert sec1 > 3.0 and os.path.dirname(path1) == str(tmp_path)  # compiled, here
    o = P.cartpole_move(cartpole_oracle, A.MODEL_USER_BASE, batch=6, goal=np.linspace(0.5, 1.5, 6))
    o.solve()
    assert iters1 == str(o.get_stats()["iterations_total"].tolist())
    sec2, path2, iters2 = run()
    assert sec2 < 2.0 and path2 == path1 and iters2 == iters1  # second process: the cache entry is taken
    # foreign code under the cached name: the plugin of another source (any other .so of the in-tree cache)
    other = A.user_model_path(A.register_model_source("cartpole", CARTPOLE))
    shutil.copyfile(other, path1)
    sec3, path3, iters3 = run()
    assert sec3 > 3.0 and path3 == path1 and iters3 == iters1  # noticed and rebuilt


@pytest.mark.gpu
def test_wrong_entry_does_not_hide_behind_a_large_jacobian(A):
    """the derivative check is per ENTRY, |J_fd - J|_ij / max(1, |J_ij|) < 1e-4.  A norm-wise relative error let a
    wrong entry of order one through once the Jacobian as a whole was large: here ||J|| ~ 1000 and one entry is off by 1e-2
    (1e-5 of the norm)."""
    good = open(os.path.join(ROOT, "tests", "models", "stiff_pair.hpp")).read()
    assert A.register_model_source("stiff_pair", good) >= A.MODEL_USER_BASE
    bad = good.replace("J[5] = T(1);", "J[5] = T(1.01);")
    assert bad != good
    with pytest.raises(A.AltroError, match="does not match finite differences"):
        A.register_model_source("stiff_pair_bad", bad)
