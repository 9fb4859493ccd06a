"""Per-knot steps and times (Trajectory::SetStep / SetTime, /root/reference/altro/common/trajectory.hpp:119-120; float h, t
per knot in knotpoint.hpp:179-180) and time-varying dynamics (ContinuousDynamics::Evaluate(x, u, t, xdot),
altro/problem/dynamics.hpp:59-95; RungeKutta4's stage times, integration.hpp:123-150): the C-ABI's altro_set_steps /
altro_set_times against the oracle."""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WIND = open(os.path.join(ROOT, "tests", "models", "cartpole_wind.hpp")).read()


def _geometric_steps(N, tf, ratio):
    """h_k = h_0 * ratio^k with sum = tf, as 32-bit floats (a fine grid where the manoeuvre starts)."""
    w = ratio ** np.arange(N)
    return (tf * w / w.sum()).astype(np.float32)


def _same_schedule(o, g):
    so, sg = o.get_stats(), g.get_stats()
    for f in ("status", "iterations_total", "iterations_outer", "iterations_inner"):
        assert (so[f] == sg[f]).all(), (f, so[f], sg[f])
    assert np.array_equal(so["alpha"], sg["alpha"])
    return so, sg


# ---- CPU: the setters themselves (host-side bookkeeping of the C-ABI and of the oracle) -------------------------------
def test_steps_and_times_round_trip_without_a_gpu(A):
    s = A.BatchSolver(3, 2, 10, 1, A.F64)
    s.set_uniform_step(0.1)
    hk, tk = s.get_steps()
    assert np.array_equal(hk, np.full(10, np.float32(0.1)))
    assert np.array_equal(tk[:10], np.arange(10, dtype=np.float32) * np.float32(0.1)) and tk[10] == np.float32(0.1) * 10
    steps = _geometric_steps(10, 1.0, 1.2)
    s.set_steps(steps)
    hk, tk2 = s.get_steps()
    assert np.array_equal(hk, steps) and np.array_equal(tk2, tk)  # SetStep leaves the times alone
    times = np.concatenate([[0.0], np.cumsum(steps)]).astype(np.float32)
    s.set_times(times)
    assert np.array_equal(s.get_steps()[1], times)
    with pytest.raises(A.AltroError, match="expected N = 10 steps"):
        s.set_steps(steps[:5])
    with pytest.raises(A.AltroError, match="not positive"):
        s.set_steps(np.zeros(10))
    s.set_uniform_step(0.2)  # SetUniformStep overwrites both again
    hk, tk = s.get_steps()
    assert np.array_equal(hk, np.full(10, np.float32(0.2))) and tk[3] == np.float32(3) * np.float32(0.2)


def test_oracle_equal_steps_equal_the_uniform_step(P, oracle_make):
    """The oracle's own consistency: SetStep(k, h) with the same h on every knot is SetUniformStep(h), bit for bit."""
    a = P.batch_turn90(oracle_make, batch=3)
    b = P.batch_turn90(oracle_make, batch=3)
    h = a.get_steps()[0]
    b.set_steps(h)
    a.solve(); b.solve()
    assert (a.get_stats()["iterations_total"] == b.get_stats()["iterations_total"]).all()
    assert np.array_equal(a.get_trajectory()[0], b.get_trajectory()[0])


# ---- GPU ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype_name", ["F64", "F32"])
def test_geometric_step_schedule_matches_the_oracle(A, P, oracle_make, hip_make, dtype_name):
    """AL-iLQR on the unicycle with a geometric step schedule (h_99 / h_0 = 2.7): exact schedule, X to 1e-7."""
    B = 24
    dt = getattr(A, dtype_name)
    g = P.batch_turn90(hip_make, batch=B, dtype=dt)
    o = P.batch_turn90(oracle_make, batch=B, dtype=A.F64 if dtype_name == "F64" else 2)
    steps = _geometric_steps(100, 3.0, 1.01)
    for s in (g, o):
        s.set_steps(steps)
        s.set_times(np.concatenate([[0.0], np.cumsum(steps)]).astype(np.float32))
        s.solve()
    so, _ = _same_schedule(o, g)
    assert (so["status"] == 0).mean() > 0.8
    ok = so["status"] == 0
    Xo, Uo = o.get_trajectory()
    Xg, Ug = g.get_trajectory()
    assert np.abs(Xg[ok] - Xo[ok]).max() < 1e-7 and np.abs(Ug[ok] - Uo[ok]).max() < 1e-6
    # the steps really are in the dynamics: the uniform-step solution differs
    u = P.batch_turn90(hip_make, batch=B, dtype=dt)
    u.solve()
    assert np.abs(u.get_trajectory()[0][ok] - Xg[ok]).max() > 1e-3


@pytest.mark.gpu
def test_step_level_calls_with_per_knot_steps(A, P, oracle_make, hip_make):
    """Rollout, expansions (RK4 Jacobians with h[k]), backward and forward pass one by one."""
    g = P.batch_three_obstacles(hip_make, batch=5, dtype=A.F64)  # (the factory's default is the config's ALTRO_F32)
    o = P.batch_three_obstacles(oracle_make, batch=5, dtype=A.F64)
    steps = _geometric_steps(100, 5.0, 0.985)
    for s in (g, o):
        s.set_steps(steps)
        s.al_init(); s.solve_setup(); s.rollout()
        s.update_expansions(); s.backward_pass(); s.forward_pass()
    for k in (0, 37, 99):
        eo, eg = o.get_expansion(k), g.get_expansion(k)
        assert np.allclose(eg["A"], eo["A"], rtol=1e-10, atol=1e-12) and np.allclose(eg["B"], eo["B"], rtol=1e-10, atol=1e-12)
        assert np.abs(eg["B"]).max() > 0
    Ko, do = o.get_gains()
    Kg, dg = g.get_gains()
    assert np.allclose(Kg, Ko, rtol=1e-9, atol=1e-11) and np.allclose(dg, do, rtol=1e-9, atol=1e-11)  # (measured 1.1e-14)
    assert np.array_equal(o.get_stats()["alpha"], g.get_stats()["alpha"])
    assert np.allclose(g.get_trajectory()[0], o.get_trajectory()[0], rtol=1e-9, atol=1e-11)


@pytest.mark.gpu
def test_equal_steps_take_the_general_kernels_to_the_same_solution(A, P, hip_make):
    """SetStep with one value on every knot runs the general path (k_forward, no persistent kernel); the uniform step
    runs the fused one: same schedule, same trajectories to rounding (the two forward kernels order the RK4 combination
    differently)."""
    a = P.batch_turn90(hip_make, batch=40)
    b = P.batch_turn90(hip_make, batch=40)
    b.set_steps(a.get_steps()[0])
    a.solve(); b.solve()
    sa, sb = a.get_stats(), b.get_stats()
    assert (sa["iterations_total"] == sb["iterations_total"]).all() and (sa["status"] == sb["status"]).all()
    ok = sa["status"] == 0
    assert np.abs(a.get_trajectory()[0][ok] - b.get_trajectory()[0][ok]).max() < 1e-9
    # ... and back: SetUniformStep afterwards returns the handle to the uniform-step kernels, bit for bit
    b.set_uniform_step(a.get_steps()[0][0])
    b.reset_trajectory(); b.solve()
    a.reset_trajectory(); a.solve()
    assert np.array_equal(a.get_trajectory()[0], b.get_trajectory()[0])


@pytest.fixture(scope="module")
def wind_oracle(A):
    path = os.path.join(ROOT, "oracle", "_build", "liboracle_cartpole_wind.so")
    if not os.path.exists(path):
        import __graft_entry__ as graft
        graft.build_oracle()
    lib = ctypes.CDLL(path)
    return lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d, _lib=lib, _prefix="oracle_")


@pytest.mark.gpu
@pytest.mark.parametrize("t0", [0.0, 1.25])
def test_time_varying_user_model_matches_the_oracle(A, P, hip_make, wind_oracle, t0):
    """A cart-pole in a gusting wind, f(x, u, t, xdot): the knot times reach the dynamics (stage times t, t + h/2,
    t + h/2, t + h; Jacobian times t, t/2, t/2, t as the reference has them), also when SetTime moves them."""
    kind = A.register_model_source("cartpole_wind", WIND)  # runs the device-side Jacobian check at several times
    B = 16
    goals = np.linspace(0.5, 1.4, B)
    g = P.cartpole_move(hip_make, kind, batch=B, goal=goals)
    o = P.cartpole_move(wind_oracle, kind, batch=B, goal=goals)
    if t0:
        hk, tk = g.get_steps()
        for s in (g, o):
            s.set_times(tk + np.float32(t0))
    g.solve(); o.solve()
    so, _ = _same_schedule(o, g)
    ok = so["status"] == 0
    assert ok.mean() > 0.8
    Xo, Uo = o.get_trajectory()
    Xg, Ug = g.get_trajectory()
    assert np.abs(Xg[ok] - Xo[ok]).max() < 1e-7 and np.abs(Ug[ok] - Uo[ok]).max() < 1e-6
    if t0 == 0.0:
        test_time_varying_user_model_matches_the_oracle.ref = Xg.copy()
    else:  # another phase of the gust: another trajectory
        ref = getattr(test_time_varying_user_model_matches_the_oracle, "ref", None)
        assert ref is None or np.abs(ref - Xg).max() > 1e-3


@pytest.mark.gpu
def test_time_varying_model_uploaded_before_its_step_is_known(A, hip_make):
    """the facade builds the solver from the problem BEFORE the trajectory (and with it the step) is handed over.
    For a time-varying model that upload used to fail with "the integration step of knot 0 is not set" and left the handle
    unusable for good.  Now the knot times stay unset until a step arrives, integrating calls are refused meanwhile, and
    the handle works once altro_set_uniform_step has been called."""
    kind = A.register_model_source("cartpole_wind", WIND)
    N, B = 60, 4
    s = hip_make(4, 1, N, B, A.F64)
    s.set_model(kind)
    xf = np.zeros((B, 4)); xf[:, 0] = np.linspace(0.5, 1.1, B)
    hd = float(np.float32(0.05))
    s.set_lqr_cost(0, N, np.eye(4) * (1e-1 * hd), np.eye(1) * (1e-2 * hd), xf, np.zeros(1))
    s.set_lqr_cost(N, N + 1, np.eye(4) * 100.0, np.zeros((1, 1)), xf, np.zeros(1))
    s.add_control_bound(0, N, [-3.0], [3.0])
    s.add_constraint(A.CON_GOAL, N, N + 1, xf)
    s.set_initial_state(np.zeros(4))
    s.set_trajectory(None, np.zeros((N, 1)))
    assert s.num_constraints() == 2 * N + 4      # uploads the problem: no step known yet
    with pytest.raises(A.AltroError, match="integration step is not set"):
        s.rollout()
    s.set_uniform_step(np.float32(0.05))          # Trajectory::SetUniformStep arrives: steps AND times t_k = float(k) * h
    s.solve()
    st = s.get_stats()
    assert (st["status"] == 0).all(), st
    hk, tk = s.get_steps()
    assert np.all(hk == np.float32(0.05)) and tk[3] == np.float32(3) * np.float32(0.05)
