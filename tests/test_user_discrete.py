"""The other DiscreteDynamics a knot of the reference's Problem may carry (VERDICT r3 missing #1 / #2):
DiscretizedModel<Model, ExplicitEuler> (altro/problem/integration.hpp:87-104), the caller's own problem::DiscreteDynamics
subclass (altro/problem/dynamics.hpp:148-187: Evaluate(x, u, t, h, xnext), Jacobian(x, u, t, h, jac)), and a DIFFERENT model
on every knot (Problem::SetDynamics(model, k), problem.hpp:155-166, 187-191) -- all three as user SOURCE
(include/altro_hip.h), the oracle compiled from the same text.

CPU part: the oracle against the reference's own known answer for ExplicitEuler (test/problem/triple_integrator_test.cpp:
135-156) and against the built-in paths; the plugins cross-compile.  GPU part: the HIP path against the oracle."""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = {name: open(os.path.join(ROOT, "tests", "models", name + ".hpp")).read()
       for name in ("cartpole_steps", "tripleint_euler", "pendulum_discrete")}


def _oracle(A, name):
    path = os.path.join(ROOT, "oracle", "_build", f"liboracle_{name}.so")
    if not os.path.exists(path):
        import __graft_entry__ as graft
        graft.build_oracle()
    lib = ctypes.CDLL(path)
    return lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d, _lib=lib, _prefix="oracle_")


@pytest.fixture(scope="module")
def steps_oracle(A):
    return _oracle(A, "cartpole_steps")


@pytest.fixture(scope="module")
def tieuler_oracle(A):
    return _oracle(A, "tripleint_euler")


@pytest.fixture(scope="module")
def pend_oracle(A):
    return _oracle(A, "pendulum_discrete")


def _triple(P, make, kind, knot_models, N=10, batch=1):
    """The ilqr_test.cpp fixture of the triple integrator (P.triple_integrator) on a USER model of the same dynamics."""
    s = P.triple_integrator(lambda n, m, N_, b, d: _with_model(make(n, m, N_, b, d), kind, knot_models), batch=batch, N=N)
    return s


def _with_model(s, kind, knot_models):
    # P.triple_integrator calls set_model(MODEL_TRIPLE_INTEGRATOR, [dof]) on what `make` returns: redirect it to the user kind
    orig = s.set_model
    s.set_model = lambda *_a, **_k: (orig(kind), s.set_knot_models(knot_models) if knot_models is not None else None)
    return s


def _euler_answer(h):
    """jac_ans of TripleIntegratorTest.EulerIntegration (tests/golden/reference_constants.json, K26)."""
    import json
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_constants.json")))["K26_explicit_euler_jacobian"]
    assert float(np.float32(k["h"])) == h
    num = lambda rows: np.array([[h if v == "h" else float(v) for v in r] for r in rows])  # noqa: E731
    return num(k["A_rows"]), num(k["B_rows"])


# ---- CPU: the oracle -----------------------------------------------------------------------------------------------
def test_explicit_euler_known_answer_on_the_oracle(A, P, tieuler_oracle):
    """test/problem/triple_integrator_test.cpp:135-156 (TripleIntegratorTest.EulerIntegration): xnext = x + f(x, u) h and
    jac = [1 0 h 0 0 0 0 0; 0 1 0 h ...; ...; 0 0 0 0 0 1 0 h] for h = 0.1f -- isApprox (1e-12 relative) in the reference."""
    N = 10
    o = _triple(P, tieuler_oracle, A.MODEL_USER_BASE, np.ones(N, dtype=np.int32), N=N)
    rng = np.random.default_rng(0)
    x0 = rng.uniform(-1, 1, 6)
    U = rng.uniform(-1, 1, (N, 2))
    o.set_initial_state(x0)
    o.set_trajectory(None, U)
    o.rollout()
    o.update_expansions()
    X, _ = o.get_trajectory()
    h = float(np.float32(0.1))
    e = o.get_expansion(0)
    Aans, Bans = _euler_answer(h)
    assert np.allclose(e["A"][0], Aans, rtol=1e-12, atol=0) and np.allclose(e["B"][0], Bans, rtol=1e-12, atol=0)
    f0 = np.concatenate([x0[2:4], x0[4:6], U[0]])
    assert np.allclose(X[0, 1], x0 + f0 * h, rtol=1e-12, atol=1e-15)


def test_model_list_index_zero_is_the_builtin_triple_integrator(A, P, tieuler_oracle, oracle_make):
    """Index 0 of the list is the same RK4-discretised triple integrator the built-in model is: same solve, bit for bit
    (the K6 fixture: 2 iterations); index 1 (Euler) on every knot is a different discretisation and a different solve."""
    ref = P.triple_integrator(oracle_make)
    rk4 = _triple(P, tieuler_oracle, A.MODEL_USER_BASE, np.zeros(10, dtype=np.int32))
    eul = _triple(P, tieuler_oracle, A.MODEL_USER_BASE, np.ones(10, dtype=np.int32))
    for s in (ref, rk4, eul):
        s.solve_ilqr()
    sr, s0, s1 = ref.get_stats(), rk4.get_stats(), eul.get_stats()
    assert sr["iterations_total"][0] == 2 and s0["iterations_total"][0] == 2 and sr["status"][0] == s0["status"][0] == 0
    assert np.array_equal(ref.get_trajectory()[0], rk4.get_trajectory()[0])
    assert np.array_equal(ref.get_gains()[0], rk4.get_gains()[0])
    assert s1["status"][0] == 0 and not np.allclose(eul.get_trajectory()[0], rk4.get_trajectory()[0], atol=1e-6)


def test_mixed_knots_on_the_oracle(A, P, steps_oracle):
    """RK4 on the first third, explicit Euler on the second, the caller's symplectic map (with its time-dependent gust)
    on the last: the solve converges, and each knot's successor state obeys ITS model's map."""
    N = 60
    km = np.repeat([0, 1, 2], N // 3).astype(np.int32)
    o = P.cartpole_steps(steps_oracle, A.MODEL_USER_BASE, km, batch=3, goal=[0.5, 0.8, 1.1])
    o.solve()
    st = o.get_stats()
    assert (st["status"] == 0).all(), st
    X, U = o.get_trajectory()
    h = float(np.float32(0.05))

    def f(x, F):
        mc, mp, l, g = 1.0, 0.2, 0.5, 9.81
        s, c, q = np.sin(x[1]), np.cos(x[1]), x[3]
        D = mc + mp * s * s
        return np.array([x[2], q, (F + mp * s * (l * q * q + g * c)) / D,
                         (-F * c - mp * l * q * q * c * s - (mc + mp) * g * s) / (l * D)])
    k = 25  # an Euler knot
    assert np.allclose(X[0, k + 1], X[0, k] + f(X[0, k], U[0, k, 0]) * h, rtol=1e-12, atol=1e-14)
    k = 45  # a symplectic knot: t_k = float(k) * h (Trajectory::SetUniformStep)
    t = float(np.float32(k) * np.float32(0.05))
    xd = f(X[0, k], U[0, k, 0] + 0.3 * np.sin(1.7 * t))
    v = X[0, k, 2:] + xd[2:] * h
    assert np.allclose(X[0, k + 1], np.concatenate([X[0, k, :2] + v * h, v]), rtol=1e-12, atol=1e-14)
    # ... and all-RK4 knots of the same source reproduce the plain cart-pole of tests/models/cartpole.hpp bit for bit
    plain = P.cartpole_move(_oracle(A, "cartpole"), A.MODEL_USER_BASE, batch=3, goal=[0.5, 0.8, 1.1])
    allrk = P.cartpole_steps(steps_oracle, A.MODEL_USER_BASE, None, batch=3, goal=[0.5, 0.8, 1.1])
    plain.solve(); allrk.solve()
    assert np.array_equal(plain.get_trajectory()[0], allrk.get_trajectory()[0])
    assert np.array_equal(plain.get_stats()["iterations_total"], allrk.get_stats()["iterations_total"])


def test_plugins_of_the_new_model_kinds_compile(A):
    """No GPU needed: the three sources cross-compile into plugins (every kernel of the solver instantiated for a model
    list / an Euler model / a discrete-only model) and load."""
    os.environ.setdefault("ALTRO_HIP_ARCH", "gfx950")
    for name, src in SRC.items():
        assert A.register_model_source(name, src) >= A.MODEL_USER_BASE


def test_knot_model_indices_are_validated(A):
    s = A.BatchSolver(4, 1, 10, 1, A.F64)
    with pytest.raises(A.AltroError):
        s.set_knot_models(np.zeros(9, dtype=np.int32))  # N indices expected
    with pytest.raises(A.AltroError):
        s.set_knot_models(-np.ones(10, dtype=np.int32))


# ---- GPU: the HIP path against the oracle ------------------------------------------------------------------------------
def _parity(o, g, xtol=1e-10, min_solved=0.9):  # (measured X 7e-13, U 7e-12 abs over all three models and both engines)
    so, sg = o.get_stats(), g.get_stats()
    for f in ("status", "iterations_total", "iterations_outer"):
        assert (so[f] == sg[f]).all(), (f, so[f], sg[f])
    ok = so["status"] == 0
    assert ok.mean() > min_solved
    (Xo, Uo), (Xg, Ug) = o.get_trajectory(), g.get_trajectory()
    assert np.allclose(Xg[ok], Xo[ok], rtol=xtol, atol=xtol), np.abs(Xg[ok] - Xo[ok]).max()
    assert np.allclose(Ug[ok], Uo[ok], rtol=10 * xtol, atol=10 * xtol), np.abs(Ug[ok] - Uo[ok]).max()
    assert np.allclose(sg["cost"][ok], so["cost"][ok], rtol=1e-7)
    return so


@pytest.mark.gpu
def test_explicit_euler_known_answer_on_the_gpu(A, P, hip_make):
    kind = A.register_model_source("tripleint_euler", SRC["tripleint_euler"])  # (checks every model's Jacobian on the device)
    N = 10
    g = _triple(P, hip_make, kind, np.ones(N, dtype=np.int32), N=N)
    rng = np.random.default_rng(0)
    x0 = rng.uniform(-1, 1, 6)
    U = rng.uniform(-1, 1, (N, 2))
    g.set_initial_state(x0)
    g.set_trajectory(None, U)
    g.rollout()
    g.update_expansions()
    h = float(np.float32(0.1))
    e = g.get_expansion(3)
    Aans, Bans = _euler_answer(h)
    assert np.allclose(e["A"][0], Aans, rtol=1e-12, atol=0) and np.allclose(e["B"][0], Bans, rtol=1e-12, atol=0)
    X, _ = g.get_trajectory()
    assert np.allclose(X[0, 1], x0 + np.concatenate([x0[2:4], x0[4:6], U[0]]) * h, rtol=1e-12, atol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype_name", ["F64", "F32"])
def test_model_per_knot_matches_the_oracle(A, P, hip_make, steps_oracle, dtype_name):
    kind = A.register_model_source("cartpole_steps", SRC["cartpole_steps"])
    B, N = 24, 60
    goals = np.linspace(0.4, 1.5, B)
    rng = np.random.default_rng(5)
    for km in (np.repeat([0, 1, 2], N // 3), np.repeat([2, 0, 1], N // 3), rng.integers(0, 3, N)):
        km = km.astype(np.int32)
        g = P.cartpole_steps(hip_make, kind, km, batch=B, goal=goals, dtype=getattr(A, dtype_name))
        o = P.cartpole_steps(steps_oracle, kind, km, batch=B, goal=goals, dtype=A.F64 if dtype_name == "F64" else 2)
        g.solve(); o.solve()
        so = _parity(o, g)
        print("model per knot", dtype_name, km[:6], "...", "iterations", np.unique(so["iterations_total"]))
        g.close()
    # step level: every knot's [A | B] is its own model's
    g = P.cartpole_steps(hip_make, kind, km, batch=2, goal=goals[:2])
    o = P.cartpole_steps(steps_oracle, kind, km, batch=2, goal=goals[:2])
    for s in (g, o):
        s.set_trajectory(None, np.full((N, 1), 0.7)); s.rollout(); s.update_expansions()
    for k in range(0, N, 7):
        eo, eg = o.get_expansion(k), g.get_expansion(k)
        assert np.allclose(eg["A"], eo["A"], rtol=1e-12, atol=1e-14) and np.allclose(eg["B"], eo["B"], rtol=1e-12, atol=1e-14), k
    # an index outside the list is refused at the first compute call
    bad = P.cartpole_steps(hip_make, kind, np.full(N, 3, dtype=np.int32), batch=1)
    with pytest.raises(A.AltroError, match="lists 3 models"):
        bad.rollout()


@pytest.mark.gpu
def test_discrete_only_model_matches_the_oracle(A, P, hip_make, pend_oracle):
    kind = A.register_model_source("pendulum_discrete", SRC["pendulum_discrete"])  # step_jac checked against step
    B = 32
    goals = np.linspace(0.3, 1.2, B)
    g = P.pendulum_swing(hip_make, kind, batch=B, goal=goals)
    o = P.pendulum_swing(pend_oracle, kind, batch=B, goal=goals)
    g.solve(); o.solve()
    so = _parity(o, g, min_solved=0.8)  # (4 of the 32 swings run into max_iterations_inner on both sides, same schedule)
    print("discrete-only pendulum: iterations", np.unique(so["iterations_total"], return_counts=True))
    Xg, _ = g.get_trajectory()
    ok = so["status"] == 0
    assert (np.abs(Xg[ok][:, -1, 0] - goals[ok]) < 1e-3).all()
    # a wrong step_jac is caught by the device-side check at registration
    bad = SRC["pendulum_discrete"].replace("J[1 + 2 * 2] = o2;", "J[1 + 2 * 2] = -o2;")
    assert bad != SRC["pendulum_discrete"]
    with pytest.raises(A.AltroError, match="does not match finite differences"):
        A.register_model_source("pendulum_bad", bad)


@pytest.mark.gpu
def test_single_model_handle_rejects_other_indices(A, P, hip_make):
    s = P.unicycle_turn90(lambda n, m, N, b, d: _with_idx(hip_make(n, m, N, b, d)), batch=1)
    with pytest.raises(A.AltroError, match="is a single model"):
        s.rollout()


def _with_idx(s):
    orig = s.set_model
    s.set_model = lambda *a, **k: (orig(*a, **k), s.set_knot_models(np.ones(s.N, dtype=np.int32)))
    return s
