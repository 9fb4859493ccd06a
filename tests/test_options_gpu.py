"""GPU vs oracle under non-default SolverOptions and on the failure paths of the reference algorithm:
regularisation restarts (Cholesky failure), exhausted line searches, state/control limits, warm starts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(o, g, traj_tol=1e-9, only_solved=False):  # (bars pinned to profiles/r05_parity_errors.json: measured X 2.3e-12, U 8.2e-11 abs)
    so, sg = o.get_stats(), g.get_stats()
    for f in ("status", "status_ilqr", "iterations_total", "iterations_outer", "iterations_inner"):
        assert (so[f] == sg[f]).all(), (f, so[f], sg[f])
    Xo, Uo = o.get_trajectory()
    Xg, Ug = g.get_trajectory()
    ok = (so["status"] == 0) if only_solved else np.isfinite(Xo).all(axis=(1, 2))
    assert np.allclose(Xg[ok], Xo[ok], rtol=traj_tol, atol=traj_tol)
    assert np.allclose(Ug[ok], Uo[ok], rtol=traj_tol, atol=traj_tol)
    assert np.allclose(sg["regularization"], so["regularization"], rtol=1e-12)
    return so


@pytest.mark.parametrize("kw", [
    dict(line_search_max_iterations=10),
    dict(line_search_max_iterations=25, line_search_decrease_factor=1.5),  # > 20 lanes: single-wave fallback kernel
    dict(line_search_max_iterations=4, max_iterations_inner=30),
    dict(constraint_tolerance=1e-6),
    dict(initial_penalty=10.0),
    dict(check_forwardpass_bounds=0),
    dict(bp_reg_initial=1e-3),
    dict(max_iterations_inner=5, max_iterations_outer=4),
    dict(max_iterations_total=7),
    dict(cost_tolerance=1e-6, gradient_tolerance=1e-4),
    dict(line_search_decrease_factor=3.0),
])
def test_option_variations(P, oracle_make, hip_make, kw):
    o = P.batch_turn90(oracle_make, batch=12)
    g = P.batch_turn90(hip_make, batch=12)
    o.set_options(**kw); g.set_options(**kw)
    o.solve(); g.solve()
    _same(o, g, only_solved=True)


def test_penalty_scaling_and_set_penalty(P, oracle_make, hip_make):
    o = P.batch_turn90(oracle_make, batch=6)
    g = P.batch_turn90(hip_make, batch=6)
    for s in (o, g):
        s.set_penalty_scaling(4.0)
        s.set_options(initial_penalty=0.0)  # keep the penalties that SetPenalty installs (quirk Q8)
        s.set_penalty(3.0)
        s.solve()
    so = _same(o, g, only_solved=True)
    assert np.allclose(g.get_penalties(), o.get_penalties())
    assert (so["max_penalty"] % 3.0 == 0).all()


def test_warm_start_keeps_duals(P, oracle_make, hip_make):
    # MPC pattern (al_solver.hpp:292-297): second solve with reset_duals = false, initial_penalty = 0
    o = P.batch_turn90(oracle_make, batch=6)
    g = P.batch_turn90(hip_make, batch=6)
    for s in (o, g):
        s.solve()
        s.set_options(reset_duals=0, initial_penalty=0.0)
        s.solve()
    so = _same(o, g, only_solved=True)
    assert (so["iterations_total"] <= 3).all()  # already converged: the warm start needs almost nothing


def test_cholesky_restart_path(P, A, oracle_make, hip_make):
    """Indefinite R makes Quu + rho I fail its Cholesky factorisation until the regularisation has grown:
    exercises IncreaseRegularization + sweep restart (ilqr.hpp:409-427) on both backward kernels."""
    def build(make, dtype):
        s = make(3, 2, 40, 6, dtype)
        h = np.float32(0.05)
        s.set_model(A.MODEL_UNICYCLE)
        s.set_uniform_step(h)
        xf = np.tile(np.array([1.0, 0.5, 0.3]), (6, 1)) + np.linspace(0, 0.3, 6)[:, None]
        R = np.diag([-2e-3, 1e-3])
        s.set_lqr_cost(0, 40, np.eye(3) * 1e-3, R, xf, np.zeros(2))
        s.set_lqr_cost(40, 41, np.eye(3) * 10.0, R * 0, xf, np.zeros(2))
        s.set_initial_state(np.zeros(3))
        s.set_trajectory(None, np.full((40, 2), 0.05))
        return s
    o, g = build(oracle_make, A.F64), build(hip_make, A.F64)
    # the cost is unbounded below, so only the first iterations are compared (the iterates are chaotic later)
    o.set_options(max_iterations_inner=4); g.set_options(max_iterations_inner=4)
    o.solve_ilqr(); g.solve_ilqr()
    so, sg = o.get_stats(), g.get_stats()
    assert (so["regularization"] > 1e-8).any()  # the restart path was really taken
    for f in ("status", "iterations_total"):
        assert (so[f] == sg[f]).all(), (f, so[f], sg[f])
    assert np.allclose(sg["regularization"], so["regularization"], rtol=1e-12)
    Xo, _ = o.get_trajectory()
    Xg, _ = g.get_trajectory()
    assert np.allclose(Xg, Xo, rtol=1e-9, atol=1e-11)  # (measured 1e-15)


def test_cholesky_restart_beside_finished_instances(A, oracle_make, hip_make):
    """A restarted sweep beside instances that do not need one, on a horizon of more than one gain chunk (N = 160 > 126).
    Even instances start inside their control bounds: R is indefinite, Quu + rho I fails its factorisation and the sweep
    restarts until the regularisation has grown.  Odd instances start beyond both bounds: the active AL terms add the penalty
    to Quu's diagonal and their first sweep goes through.  The four instances of a wavefront of k_backward_mfma are two of
    each kind, so the restarted sweeps run with finished neighbours -- whose LDS slots the bulk store used to write out as
    well (after the first chunk they hold other knots' gains; round 6).  Gains of the one backward pass against the oracle."""
    def build(make):
        N, B = 160, 640
        s = make(3, 2, N, B, A.F64)
        s.set_model(A.MODEL_UNICYCLE)
        s.set_uniform_step(np.float32(0.05))
        xf = np.tile(np.array([1.0, 0.5, 0.3]), (B, 1)) + np.linspace(0, 0.3, B)[:, None]
        R = np.diag([-2e-3, 1e-3])
        s.set_lqr_cost(0, N, np.eye(3) * 1e-3, R, xf, np.zeros(2))
        s.set_lqr_cost(N, N + 1, np.eye(3) * 10.0, R * 0, xf, np.zeros(2))
        s.add_control_bound(0, N, [-0.1, -0.1], [0.1, 0.1])
        s.set_initial_state(np.zeros(3))
        U = np.zeros((B, N, 2))
        U[0::2] = 0.05
        U[1::2] = 0.5
        s.set_trajectory(None, U)
        s.set_options(max_iterations_inner=1, max_iterations_outer=1)
        return s
    o, g = build(oracle_make), build(hip_make)
    o.solve(); g.solve()
    so, sg = o.get_stats(), g.get_stats()
    assert (so["regularization"][0::2] > 0.1).all() and (so["regularization"][1::2] == 0).all()  # restarts on the even ones only
    assert g.get_timing()["fused_sweeps"] == 0  # (the batched kernels ran: the backward pass with the chunked store)
    assert (so["status"] == sg["status"]).all() and (so["iterations_total"] == sg["iterations_total"]).all()
    assert np.allclose(sg["regularization"], so["regularization"], rtol=1e-12)
    Ko, do = o.get_gains()
    Kg, dg = g.get_gains()
    for b in range(len(Ko)):
        assert np.linalg.norm(Kg[b] - Ko[b]) <= 1e-9 * np.linalg.norm(Ko[b]), b
    assert np.allclose(dg, do, rtol=1e-8, atol=1e-10)


def test_state_limit_and_rejected_line_search(P, oracle_make, hip_make):
    """Huge feedforward steps: rollouts blow through state_max (kStateLimit) for large alpha and the line
    search has to back off; with a tiny state_max every trial fails and the status must say so."""
    o = P.batch_turn90(oracle_make, batch=4)
    g = P.batch_turn90(hip_make, batch=4)
    for s in (o, g):
        s.set_options(state_max=2.5, control_max=50.0, max_iterations_inner=8, max_iterations_outer=2)
        s.solve()
    so = _same(o, g, traj_tol=1e-9)
    assert set(np.unique(so["status"])) <= {0, 2, 3, 5, 6, 7}
    o2 = P.batch_turn90(oracle_make, batch=4)
    g2 = P.batch_turn90(hip_make, batch=4)
    for s in (o2, g2):
        s.set_options(state_max=1e-3, max_iterations_inner=3, max_iterations_outer=1)
        s.solve()
    so2 = _same(o2, g2)
    assert (so2["status"] == 2).all()  # kStateLimit: every trial of the last line search left the box


def test_valu_backward_matches_mfma(P, hip_make, monkeypatch):
    """The one-lane-per-instance VALU backward pass and the MFMA one agree (same schedule, ~1e-12)."""
    g1 = P.batch_turn90(hip_make, batch=16)
    g1.solve()
    monkeypatch.setenv("ALTRO_HIP_BACKWARD", "valu")
    g2 = P.batch_turn90(hip_make, batch=16)
    g2.solve()
    s1, s2 = g1.get_stats(), g2.get_stats()
    assert (s1["iterations_total"] == s2["iterations_total"]).all() and (s1["status"] == s2["status"]).all()
    ok = s1["status"] == 0
    assert np.allclose(g1.get_trajectory()[0][ok], g2.get_trajectory()[0][ok], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("seed", list(range(8)))
def test_randomised_problems_and_options(P, A, oracle_make, hip_make, seed):
    """Seeded random draws of goals, initial states, obstacle positions and solver options: the GPU must walk
    the same schedule as the oracle (iteration counts, statuses, penalties) and land on the same trajectories."""
    rng = np.random.default_rng(20260927 + seed)
    B = 10
    circles = np.tile(P.THREE_OBSTACLE_CIRCLES, (B, 1, 1))
    circles[:, :, :2] += rng.uniform(-0.15, 0.15, (B, 3, 2))
    opts = dict(
        max_iterations_inner=int(rng.integers(20, 120)),
        max_iterations_outer=int(rng.integers(3, 12)),
        cost_tolerance=float(10.0 ** rng.uniform(-6, -3)),
        gradient_tolerance=float(10.0 ** rng.uniform(-4, -1)),
        constraint_tolerance=float(10.0 ** rng.uniform(-6, -3)),
        initial_penalty=float(rng.choice([0.1, 1.0, 10.0, 100.0])),
        line_search_max_iterations=int(rng.integers(6, 21)),
        line_search_decrease_factor=float(rng.choice([1.5, 2.0, 3.0])),
        bp_reg_initial=float(rng.choice([0.0, 0.0, 1e-4])),
    )
    x0 = np.zeros((B, 3))
    x0[:, :2] = rng.uniform(-0.2, 0.2, (B, 2))
    x0[:, 2] = rng.uniform(-0.3, 0.3, B)
    solvers = []
    for make in (oracle_make, hip_make):
        s = P.unicycle_three_obstacles(make, batch=B, dtype=A.F64, circles=circles)
        s.set_initial_state(x0)
        s.set_penalty_scaling(float([2.0, 10.0, 10.0][seed % 3]))
        s.set_options(**opts)
        s.solve()
        solvers.append(s)
    o, g = solvers
    so, sg = o.get_stats(), g.get_stats()
    for f in ("status", "status_ilqr", "iterations_total", "iterations_outer", "iterations_inner"):
        assert (so[f] == sg[f]).all(), (f, opts, so[f], sg[f])
    assert np.allclose(sg["max_penalty"], so["max_penalty"], rtol=1e-12)
    ok = so["status"] == 0
    Xo, Uo = o.get_trajectory()
    Xg, Ug = g.get_trajectory()
    assert np.allclose(Xg[ok], Xo[ok], rtol=1e-9, atol=1e-11)  # (measured: X 5e-14, U 4.3e-13 abs)
    assert np.allclose(Ug[ok], Uo[ok], rtol=1e-9, atol=1e-11)


def test_async_solve_matches_blocking(P, A, hip_make):
    """altro_solve_al_async / altro_solve_poll / altro_wait: same result as the blocking call."""
    g1 = P.batch_turn90(hip_make, batch=32)
    g1.solve()
    g2 = P.batch_turn90(hip_make, batch=32)
    with pytest.raises(A.AltroError):
        g2.wait()  # nothing pending
    g2.solve_async()
    with pytest.raises(A.AltroError):
        g2.solve_async()  # one at a time
    polls = 0
    while not g2.poll():
        polls += 1
    g2.wait()
    assert g2.poll()
    s1, s2 = g1.get_stats(), g2.get_stats()
    assert (s1["iterations_total"] == s2["iterations_total"]).all() and (s1["status"] == s2["status"]).all()
    assert np.array_equal(g1.get_trajectory()[0], g2.get_trajectory()[0])
