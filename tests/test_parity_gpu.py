"""GPU parity: libaltro_hip.so (through the C-ABI) against the CPU oracle on identical inputs.

Tolerances:
  fp64 GPU vs fp64 oracle: iteration counts, step lengths and statuses EXACT; trajectories, gains,
  duals within rtol 1e-7 (atol 1e-9) -- the two sides differ only in FMA contraction and libm
  sin/cos rounding, which the iterations amplify by a few orders of magnitude.
  ALTRO_F32 engine (fp32 records, fp64 state and arithmetic): tests/test_f32_gpu.py.
"""
import inspect
import os

import numpy as np
import pytest

import _ledger

pytestmark = pytest.mark.gpu

RT, AT = 1e-7, 1e-9


def close(a, b, rtol=RT, atol=AT, label=None):
    """Elementwise bar; the measured maxima go to the ledger (tests/_ledger.py -> profiles/r05_parity_errors.json)."""
    if label is None:
        label = f"line {inspect.currentframe().f_back.f_lineno}"
    _ledger.close(a, b, rtol, atol, label)


def both(P, factory, oracle_make, hip_make, **kw):
    return factory(oracle_make, **kw), factory(hip_make, **kw)


def test_step_level_unicycle(P, oracle_make, hip_make):
    """Mirror of UnicycleiLQRTest (test/ilqr/unicycle_ilqr_test.cpp:32-88) on a jittered batch."""
    o, g = both(P, P.batch_turn90, oracle_make, hip_make, batch=8)
    for s in (o, g):
        s.set_record_ctg(True)
        s.set_record_history(8)
        s.rollout()
    close(g.cost(), o.cost(), 1e-12)
    close(g.get_trajectory()[0], o.get_trajectory()[0], 1e-12, 1e-13)
    for it in range(2):
        for s in (o, g):
            s.update_expansions()
        for k in (0, 1, 50, 99, 100):
            eo, eg = o.get_expansion(k), g.get_expansion(k)
            for key in ("lxx", "lx") + (("A", "B", "lxu", "luu", "lu") if k < 100 else ()):
                close(eg[key], eo[key], 1e-10, 1e-12)
        close(g.get_knot_costs(), o.get_knot_costs(), 1e-11, 1e-13)
        close(g.get_constraint_values(), o.get_constraint_values(), 1e-10, 1e-11)
        for s in (o, g):
            s.backward_pass()
        Ko, do = o.get_gains()
        Kg, dg = g.get_gains()
        close(Kg, Ko, 1e-8, 1e-10)
        close(dg, do, 1e-8, 1e-10)
        Po, po = o.get_ctg()
        Pg, pg = g.get_ctg()
        close(Pg, Po, 1e-8, 1e-10)
        close(pg, po, 1e-8, 1e-10)
        for s in (o, g):
            s.forward_pass()
        so, sg = o.get_stats(), g.get_stats()
        assert (so["alpha"] == sg["alpha"]).all()
        close(sg["cost"], so["cost"], 1e-9)
        close(sg["improvement_ratio"], so["improvement_ratio"], 1e-7)
        Xo, Uo = o.get_trajectory()
        Xg, Ug = g.get_trajectory()
        close(Xg, Xo, 1e-8, 1e-10)
        close(Ug, Uo, 1e-8, 1e-10)
        close(g.get_constraint_values(), o.get_constraint_values(), 1e-8, 1e-10)
    assert g.get_stats()["alpha"][0] == 0.0625 or it  # K11 on instance 0 checked below


def test_step_level_quadrotor12(P, A, oracle_make, hip_make):
    """The n = 12 path at STEP level against the oracle: one expansions / backward / forward sweep of
    batch_quadrotor12 in fp64 -- [A|B], lxx ... lu of k_expansions (RK4 Jacobian chain with its structural zeros), K, d and
    P, p of k_backward_mfma16 (the fp64 16x16x4 MFMA Riccati step), the line search of k_forward2 -- each against an
    absolute bar.  Two sweeps: the second runs on the trajectory the first one accepted (non-trivial gains and states).

    The second sweep's Riccati recursion is ill-conditioned (Qf / Q = 5e5 over 200 knots): the ORACLE's own K moves by
    6e-4 norm-wise when its input trajectory is moved by one unit in the last place (measured on CPU; 1e-16 and 1e-12
    relative perturbations give the same 1e-3: the amplification saturates).  So that sweep is judged against a second
    oracle instance `o2` that gets the same trajectory with ~1-ulp noise: the GPU may differ from the oracle by no more
    than 10 x what the oracle differs from itself (and never less than the 1e-8 bar); both figures go to the ledger."""
    N = 200
    o, g = both(P, P.batch_quadrotor12, oracle_make, hip_make, batch=6, N=N, dtype=A.F64)
    o2 = P.batch_quadrotor12(oracle_make, batch=6, N=N, dtype=A.F64)
    for s in (o, g, o2):
        s.set_record_ctg(True)
        s.rollout()
    close(g.cost(), o.cost(), 1e-12, 0.0, label="initial cost")
    close(g.get_trajectory()[0], o.get_trajectory()[0], 1e-12, 1e-13, label="X rollout")
    def spread(a, b):
        ax = tuple(range(1, a.ndim))
        return float((np.abs(a - b).max(axis=ax) / np.maximum(np.abs(b).max(axis=ax), 1e-12)).max())

    for it in range(2):
        for s in (o, g, o2):
            s.update_expansions()
        for k in (0, 1, 100, N - 1, N):
            eo, eg = o.get_expansion(k), g.get_expansion(k)
            for key in ("lxx", "lx") + (("A", "B", "lxu", "luu", "lu") if k < N else ()):
                close(eg[key], eo[key], 1e-10, 1e-12, label=f"expansion {key} (sweep {it})")
        # (sweep 0 starts from identical inputs: expansions and costs agree to the last bit or two; sweep 1 starts from
        #  trajectories that differ by the 3e-11 of the first line search, which the bars of that sweep carry)
        close(g.get_knot_costs(), o.get_knot_costs(), 1e-11, 1e-13 if it == 0 else 2e-12, label=f"knot costs (sweep {it})")
        close(g.get_constraint_values(), o.get_constraint_values(), 1e-10, 1e-11 if it == 0 else 1e-9, label=f"constraint values (sweep {it})")
        for s in (o, g, o2):
            s.backward_pass()
        assert (o.get_stats()["regularization"] == g.get_stats()["regularization"]).all()
        # (measured, sweep 0: K 1.9e-11, d 6.5e-12, P 6.7e-11, p 2.9e-11 norm-wise.  Sweep 1 runs from non-trivial gains and
        #  states -- handed to BOTH sides bit for bit, see the end of the loop -- so that its bars measure the kernels, not
        #  the 1e10-fold sensitivity of this model to its inputs, which test_config5_* measures)
        Ko, do = o.get_gains()
        Kg, dg = g.get_gains()
        Po, po = o.get_ctg()
        Pg, pg = g.get_ctg()
        nwK = nwd = nwP = nwp = 1e-9
        if it == 1:
            K2, d2 = o2.get_gains()
            P2, p2 = o2.get_ctg()
            own = [spread(K2.reshape(-1, 4, 12), Ko.reshape(-1, 4, 12)), spread(d2.reshape(-1, 4), do.reshape(-1, 4)),
                   spread(P2.reshape(-1, 12, 12), Po.reshape(-1, 12, 12)), spread(p2.reshape(-1, 12), po.reshape(-1, 12))]
            for name, v in zip("KdPp", own):
                _ledger.record(f"ORACLE vs ORACLE + 1 ulp of input: {name} per knot, normwise (sweep 1)", v, v, 0.0, 0.0, 0.0)
            nwK, nwd, nwP, nwp = (max(1e-8, 10.0 * v) for v in own)
        # norm-wise per knot block: an m x n gain block / n x n cost-to-go block has entries 1e-8 of its largest one
        # (decoupled axes), for which an element-wise relative bar is meaningless; SURVEY 8(c): rel 1e-9
        close_normwise(Kg.reshape(-1, 4, 12), Ko.reshape(-1, 4, 12), nwK, label=f"K per knot, normwise (sweep {it})")
        close_normwise(dg.reshape(-1, 4), do.reshape(-1, 4), nwd, label=f"d per knot, normwise (sweep {it})")
        close_normwise(Pg.reshape(-1, 12, 12), Po.reshape(-1, 12, 12), nwP, label=f"P per knot, normwise (sweep {it})")
        close_normwise(pg.reshape(-1, 12), po.reshape(-1, 12), nwp, label=f"p per knot, normwise (sweep {it})")
        if it == 1:
            break  # (the line search of a sweep whose gains are only defined to 1e-3 has nothing to compare)
        for s in (o, g, o2):
            s.forward_pass()
        so, sg = o.get_stats(), g.get_stats()
        assert (so["alpha"] == sg["alpha"]).all(), (so["alpha"], sg["alpha"])
        close(sg["cost"], so["cost"], 1e-9, 0.0, label=f"cost after the line search (sweep {it})")
        close(sg["improvement_ratio"], so["improvement_ratio"], 1e-7, 0.0, label=f"z (sweep {it})")
        Xo, Uo = o.get_trajectory()
        Xg, Ug = g.get_trajectory()
        # (measured in sweep 0: X 1.2e-11, U 3.3e-11 abs -- entries of 1e-4 next to entries of 1)
        close(Xg, Xo, 1e-9, 2e-10, label=f"X after the line search (sweep {it})")
        close(Ug, Uo, 1e-9, 2e-10, label=f"U after the line search (sweep {it})")
        # the next sweep starts from ONE trajectory on both sides (the GPU's); the second oracle gets it with noise of one
        # unit in the last place
        for s in (o, g):
            s.set_trajectory(Xg, Ug)
        rng = np.random.default_rng(5)
        o2.set_trajectory(Xg * (1.0 + 1e-16 * rng.standard_normal(Xg.shape)), Ug * (1.0 + 1e-16 * rng.standard_normal(Ug.shape)))


def test_reference_constants_on_gpu(P, hip_make):
    """The reference's own known-answer tests, run against the HIP path (K9-K14, K18, K19, K21, K22)."""
    s = P.unicycle_turn90(hip_make, constraints=False)
    s.set_record_ctg(True)
    s.rollout()
    assert abs(s.cost()[0] - 259.27636137767087) < 1e-5
    s.update_expansions(); s.backward_pass()
    _, p = s.get_ctg()
    _, d = s.get_gains()
    assert np.allclose(p[0, 0], [0.024904637422419617, -0.46496022574032614, -0.0573096310550007], rtol=1e-5)
    assert np.allclose(d[0, 0], [-2.565783457444465, 5.514158930898376], rtol=1e-5)
    s.forward_pass()
    assert s.get_stats()["alpha"][0] == 0.0625
    s.update_expansions(); s.backward_pass()
    _, p = s.get_ctg()
    _, d = s.get_gains()
    assert np.allclose(p[0, 0], [-0.0015143873973949232, -0.07854630832127288, -0.017945283678268698], rtol=1e-5)
    assert np.allclose(d[0, 0], [0.21887571453613042, 1.3097976615154625], rtol=1e-5)
    s.forward_pass()
    assert s.cost()[0] - 62.773696055304384 < 1e-5
    # K13
    s = P.unicycle_turn90(hip_make, constraints=False)
    s.rollout(); s.solve_ilqr()
    st = s.get_stats()[0]
    assert st["iterations_inner"] == 9 and st["status"] == 0
    assert abs(s.cost()[0] - 0.0387016567) < 1e-5
    # K14 + K18
    s = P.unicycle_turn90(hip_make, constraints=True)
    s.rollout(); s.solve_ilqr()
    assert s.get_stats()[0]["iterations_inner"] == 10
    assert abs(s.cost()[0] - 0.03893427133384412) / 0.03893427133384412 < 1e-6
    assert abs(s.get_max_violation()[0] - 0.00017691645708972636) / 0.00017691645708972636 < 1e-6
    s.update_duals(); s.update_penalties(); s.solve_ilqr()
    assert s.get_stats()[0]["iterations_inner"] == 1
    assert abs(s.max_violation()[0] - 6.26e-5) / 6.26e-5 < 0.1
    # K19 (+ SolveTwice)
    s = P.unicycle_turn90(hip_make, constraints=True)
    s.set_options(constraint_tolerance=1e-6)
    for _ in range(2):
        s.set_trajectory(None, np.full((100, 2), 0.1))
        s.solve()
        st = s.get_stats()[0]
        assert (st["iterations_total"], st["iterations_outer"], st["status"]) == (14, 5, 0)
        assert abs(s.cost()[0] - 0.03893465058924039) < 1e-12
        assert s.get_max_violation()[0] < 1e-6
    # K21 / K22 / K23
    s = P.unicycle_three_obstacles(hip_make, constraints=True)
    s.rollout()
    assert abs(s.cost()[0] - 141.9639680271223) < 1e-6
    s.set_penalty(10.0)
    assert abs(s.cost()[0] - 221.6032851439234) < 1e-6
    s.solve_ilqr(); s.update_duals(); s.update_penalties()
    lamN = s.get_duals()[0][-3:]
    assert np.allclose(lamN, [-0.43555910438329626, 0.5998598475208317, -0.0044282251970790935], rtol=1e-6)
    s = P.unicycle_three_obstacles(hip_make, constraints=True)
    s.set_penalty(10.0); s.solve()
    st = s.get_stats()[0]
    assert (st["status"], st["iterations_total"], st["iterations_outer"]) == (0, 50, 5)
    assert s.max_violation()[0] < 1e-4


def close_normwise(a, b, rtol, label="normwise"):
    """max-norm error relative to the max-norm of the reference block (per instance)."""
    _ledger.close_normwise(a, b, rtol, label)


def test_history_matches_oracle(P, oracle_make, hip_make):
    """SolverStats vectors (solver_stats.hpp:56-63): per-iteration cost / alpha / dJ / grad histories."""
    o, g = both(P, P.batch_turn90, oracle_make, hip_make, batch=4)
    g.set_record_history(64)
    o.solve(); g.solve()
    for inst in range(4):
        for field, tol in (("alpha", 0), ("cost", 1e-9), ("cost_decrease", 1e-6), ("gradient", 1e-7),
                           ("max_penalty", 0), ("regularization", 0)):
            ho = o.get_history(inst, field)
            hg = g.get_history(inst, field)
            # the reference vectors end with the row opened by the last NewIteration (a copy)
            n = min(len(ho), len(hg))
            assert n >= o.get_stats()[inst]["iterations_total"]
            if tol == 0:
                assert (ho[:n] == hg[:n]).all(), (field, ho[:n], hg[:n])
            else:
                assert np.allclose(hg[:n], ho[:n], rtol=tol, atol=1e-9), field
    assert g.get_history(0, "alpha")[0] == 0.0625  # K11


# Bars of the full-solve comparisons, pinned to what is MEASURED (profiles/r05_parity_errors.json; GPU and oracle are both
# deterministic, so the measured maxima reproduce to the bit) and to SURVEY 8(c) (fp64: rel 1e-9, abs 1e-12 scale):
#   X      measured abs <= 3.9e-12 over every config           bar rel 1e-9 + abs 2e-11
#   U      measured abs <= 6.3e-11 (|u| = 100: rel 8e-13)       bar rel 1e-9 + abs 5e-10
#   K      measured norm-wise rel <= 3.4e-10                    bar norm-wise 5e-9
#   d      measured abs <= 1.0e-10 (d -> 0 at convergence)      bar rel 1e-9 + abs 1e-9
#   duals  lambda - rho c: an error of 1e-12 in c shows up as rho 1e-12; measured 8.8e-8 abs at max penalty 1e5
#                                                               bar rel 1e-9 + abs 1e-11 x the instance's max penalty
XTOL, UTOL, KTOL, DTOL = (1e-9, 2e-11), (1e-9, 5e-10), 5e-9, (1e-9, 1e-9)


def _compare_full(o, g, solved_only_tight=True, xtol=XTOL, utol=UTOL, gtol=KTOL):
    so, sg = o.get_stats(), g.get_stats()
    assert (so["status"] == sg["status"]).all(), (so["status"], sg["status"])
    assert (so["iterations_total"] == sg["iterations_total"]).all(), np.flatnonzero(so["iterations_total"] != sg["iterations_total"])
    assert (so["iterations_outer"] == sg["iterations_outer"]).all()
    assert (so["iterations_inner"] == sg["iterations_inner"]).all()
    ok = so["status"] == 0 if solved_only_tight else np.ones(len(so), bool)
    Xo, Uo = o.get_trajectory()
    Xg, Ug = g.get_trajectory()
    close(Xg[ok], Xo[ok], *xtol, label="X")
    close(Ug[ok], Uo[ok], *utol, label="U")
    Ko, do = o.get_gains()
    Kg, dg = g.get_gains()
    close_normwise(Kg[ok], Ko[ok], gtol, label="K normwise")
    close(dg[ok], do[ok], *DTOL, label="d")  # d -> 0 at convergence: absolute floor
    if o.num_constraints() > 0:
        # a dual is lambda - rho*c with rho up to 1e4..1e8: 1e-12 in c shows up as rho*1e-12 -> the bar scales with the penalty
        lo, lg = o.get_duals()[ok], g.get_duals()[ok]
        pen = np.maximum(so["max_penalty"][ok], 1.0)[:, None]
        close(lg / pen, lo / pen, 1e-9, 1e-10, label="duals / max penalty")  # (measured 8.8e-8 abs at max penalty >= 1e4)
        close(g.get_penalties(), o.get_penalties(), 0, 0, label="penalties")
    for f in ("max_penalty", "alpha", "regularization"):
        close(sg[f][ok], so[f][ok], 0, 0, label="stat " + f)  # exact
    close(sg["cost"][ok], so["cost"][ok], 1e-10, 0.0, label="stat cost")            # measured rel 5.9e-12
    close(sg["violation"][ok], so["violation"][ok], 1e-7, 1e-12, label="stat violation")  # measured abs 1.8e-14 (rel 5e-9 of 3e-6)
    return so, sg


def test_cost_to_go_is_readable_after_a_default_solve(P, A, oracle_make, hip_make):
    """KnotPointFunctions::GetCostToGoHessian / Gradient after Solve() (knot_point_function_type.hpp:243-268): the reference
    keeps P, p of the last backward pass; the persistent kernel keeps them in registers, so altro_get_ctg behind a solve that
    did not record them runs that backward pass once more with the records on (Engine::ReplayCtg) -- the oracle's values, the
    bits of a solve that records all the way, and a solver state that the replay leaves alone."""
    o, g = both(P, P.batch_turn90, oracle_make, hip_make, batch=96)
    o.solve(); g.solve()
    assert g.get_timing()["fused_sweeps"] > 0  # (the path that does not store them)
    before = g.get_stats().copy()
    lam0, K0 = g.get_duals(), g.get_gains()[0]
    Pg, pg = g.get_ctg()
    Po, po = o.get_ctg()
    ok = o.get_stats()["status"] == 0
    close_normwise(Pg[ok].reshape(-1, 3, 3), Po[ok].reshape(-1, 3, 3), 1e-8, label="P per knot after Solve, normwise")
    close_normwise(pg[ok].reshape(-1, 3), po[ok].reshape(-1, 3), 1e-7, label="p per knot after Solve, normwise")
    after = g.get_stats()
    for f in before.dtype.names:
        assert np.array_equal(before[f], after[f]), f
    assert np.array_equal(lam0, g.get_duals()) and np.array_equal(K0, g.get_gains()[0])
    g2 = P.batch_turn90(hip_make, batch=96)
    g2.set_record_ctg(True)
    g2.solve()
    assert g2.get_timing()["fused_sweeps"] == 0
    P2, p2 = g2.get_ctg()
    assert np.array_equal(Pg, P2) and np.array_equal(pg, p2)
    # a large batch: chains of sweeps, twin workgroups in the tail -- still the recorded solve's bits
    g3 = P.batch_turn90(hip_make, batch=2304, seed=P.SEED_BASE + 3)
    g3.solve()
    P3, p3 = g3.get_ctg()
    g4 = P.batch_turn90(hip_make, batch=2304, seed=P.SEED_BASE + 3)
    g4.set_record_ctg(True)
    g4.solve()
    P4, p4 = g4.get_ctg()
    bad = np.flatnonzero((P3 != P4).reshape(len(P3), -1).any(axis=1) | (p3 != p4).reshape(len(p3), -1).any(axis=1))
    s3, s4 = g3.get_stats(), g4.get_stats()
    assert len(bad) == 0, (bad[:10], [(int(s3["iterations_total"][b]), int(s4["iterations_total"][b]), int(s3["status"][b]), float(np.abs(P3[b] - P4[b]).max()),
                                       np.flatnonzero((P3[b] != P4[b]).reshape(P3.shape[1], -1).any(axis=1))[:4].tolist()) for b in bad[:6]],
                           {k: g3.get_timing()[k] for k in ("sweep_launches", "fused_sweeps", "twin_handovers")}, {k: g4.get_timing()[k] for k in ("sweep_launches", "fused_sweeps")})
    # (round 6 regression: with SEVERAL chains of sweeps the recording backward pass of one chain must not touch what another
    #  chain's forward pass reads -- its junk sink used to be instance 0's line-search candidates, and instance 0 of g4 took 119
    #  iterations instead of 11 whenever g4 was the handle that got the four chains)
    assert np.array_equal(s3["iterations_total"], s4["iterations_total"]) and np.array_equal(g3.get_trajectory()[0], g4.get_trajectory()[0])
    # once the expansions have changed, the backward pass of the last iteration cannot be run again: nothing to read
    g5 = P.batch_turn90(hip_make, batch=8)
    g5.solve()
    g5.update_expansions()
    with pytest.raises(A.AltroError):
        g5.get_ctg()


def test_config1_three_obstacles_single(P, oracle_make, hip_make):
    o, g = both(P, P.unicycle_three_obstacles, oracle_make, hip_make)
    o.solve(); g.solve()
    so, _ = _compare_full(o, g)
    assert so["iterations_total"][0] == 50


def test_config2_triple_integrator_ilqr(P, oracle_make, hip_make):
    o, g = both(P, P.batch_triple_integrator, oracle_make, hip_make, batch=96)
    o.solve_ilqr(); g.solve_ilqr()
    so, _ = _compare_full(o, g)
    assert (so["status"] == 0).all() and (so["iterations_total"] == 2).all()


def test_config3_turn90_al_batch(P, oracle_make, hip_make):
    o, g = both(P, P.batch_turn90, oracle_make, hip_make, batch=200)
    o.solve(); g.solve()
    so, _ = _compare_full(o, g)
    assert (so["iterations_total"][0], so["iterations_outer"][0]) == (11, 2)


def test_config4_three_obstacles_batch_f64(P, A, oracle_make, hip_make):
    o, g = both(P, P.batch_three_obstacles, oracle_make, hip_make, batch=48, dtype=A.F64)
    o.solve(); g.solve()
    _compare_full(o, g)


def test_triple_integrator_constrained(P, oracle_make, hip_make):
    o, g = both(P, P.triple_integrator, oracle_make, hip_make, constraints=True)
    o.solve(); g.solve()
    _compare_full(o, g)
    _, U = g.get_trajectory()
    assert np.allclose(U[0, 0], [100, 200]) and np.allclose(U[0, -1], [100, 200])


def _config5_sensitivity(P, A, oracle_make, oracle_lib, batch):
    """How far the fp64 ORACLE itself moves when the goals of the config-5 batch are perturbed by one ulp:
    the 12-state model has Qf/Q = 5e5 over 200 knots and stops after ~4 iterations at the default (loose)
    cost tolerance, so rounding-level differences in the first iterate are amplified ~1e10-fold (dU ~ 1e-5,
    dK ~ 3e-3 norm-wise, dX ~ 5e-7).  Returns the oracle solve and the max-norm spread of X, U, K, d over
    three perturbed solves: the yardstick the GPU is compared against, instead of an asserted tolerance."""
    import ctypes
    pos = P.batch_quadrotor12_goals(batch)

    def run(xf_pos):
        s = P.quadrotor12(oracle_make, batch=batch, dtype=A.F64, xf_pos=xf_pos)
        oracle_lib.oracle_set_threads(s._h, ctypes.c_int(len(os.sched_getaffinity(0))))
        s.solve()
        return s, s.get_trajectory(), s.get_gains()
    o, (X0, U0), (K0, d0) = run(pos)
    spread = dict(X=0.0, U=0.0, K=0.0, d=0.0)
    prng = np.random.default_rng(1)
    for _ in range(3):
        _, (X1, U1), (K1, d1) = run(pos * (1.0 + prng.choice([-1.0, 1.0], pos.shape) * 2.2e-16))
        spread["X"] = max(spread["X"], np.abs(X1 - X0).max())
        spread["U"] = max(spread["U"], np.abs(U1 - U0).max())
        spread["K"] = max(spread["K"], (np.abs(K1 - K0).max(axis=(1, 2, 3)) / np.abs(K0).max(axis=(1, 2, 3))).max())
        spread["d"] = max(spread["d"], np.abs(d1 - d0).max())
    return o, spread


def _config5_against_oracle(P, A, oracle_make, hip_make, oracle_lib, batch):
    o, spread = _config5_sensitivity(P, A, oracle_make, oracle_lib, batch)
    g = P.batch_quadrotor12(hip_make, batch=batch, dtype=A.F64)
    g.solve()
    so, sg = o.get_stats(), g.get_stats()
    for f in ("status", "iterations_total", "iterations_outer", "iterations_inner"):
        assert (so[f] == sg[f]).all(), f   # the schedule is exact
    assert (so["status"] == 0).mean() > 0.99
    (Xo, Uo), (Xg, Ug) = o.get_trajectory(), g.get_trajectory()
    (Ko, do), (Kg, dg) = o.get_gains(), g.get_gains()
    err = dict(X=np.abs(Xg - Xo).max(), U=np.abs(Ug - Uo).max(), d=np.abs(dg - do).max(),
               K=(np.abs(Kg - Ko).max(axis=(1, 2, 3)) / np.abs(Ko).max(axis=(1, 2, 3))).max())
    print("config 5 fp64: GPU-vs-oracle error", {k: f"{v:.2e}" for k, v in err.items()},
          "oracle's own 1-ulp sensitivity", {k: f"{v:.2e}" for k, v in spread.items()})
    for k in err:  # the GPU differs from the oracle by no more than the oracle differs from itself
        _ledger.record(f"config5 {k} (GPU vs oracle; bar = 4 x the oracle's own 1-ulp sensitivity)", err[k], err[k],
                       err[k] / (4.0 * spread[k] + 1e-12), note=f"oracle 1-ulp spread {spread[k]:.3e}")
        assert err[k] <= 4.0 * spread[k] + 1e-12, (k, err[k], spread[k])
    assert np.allclose(sg["cost"], so["cost"], rtol=1e-7)


def test_config5_quadrotor_fp64(P, A, oracle_make, hip_make, oracle_lib):
    _config5_against_oracle(P, A, oracle_make, hip_make, oracle_lib, 16)


def test_config2_full_batch_against_oracle(P, A, oracle_make, hip_make):
    """BASELINE configs[1] at full size: 1024 triple integrators (51 knots), unconstrained iLQR, fp64."""
    o, g = both(P, P.batch_triple_integrator, oracle_make, hip_make, batch=1024)
    o.solve_ilqr(); g.solve_ilqr()
    so, _ = _compare_full(o, g)
    assert (so["status"] == 0).all() and (so["iterations_total"] == 2).all()


def test_config5_full_batch_fp64_against_oracle(P, A, oracle_make, hip_make, oracle_lib):
    """BASELINE configs[4] at full size in fp64: 1024 x 12-state model, 201 knots, full AL loop."""
    _config5_against_oracle(P, A, oracle_make, hip_make, oracle_lib, 1024)


def test_full_batch_properties(P, A, hip_make):
    """BASELINE config 3 at full size: determinism, batch independence, convergence invariants."""
    B = 4096
    g = P.batch_turn90(hip_make, batch=B)
    g.solve()
    s1 = g.get_stats().copy()
    X1, U1 = g.get_trajectory()
    # same handle, same inputs again -> bitwise identical (SolveTwice, auglag_test.cpp:353-380)
    g.set_trajectory(None, np.full((100, 2), 0.1))
    g.solve()
    s2 = g.get_stats()
    X2, U2 = g.get_trajectory()
    assert (X1 == X2).all() and (U1 == U2).all()
    for f in s1.dtype.names:
        assert (s1[f] == s2[f]).all(), f
    # instance 0 is the reference problem: identical to a batch of one
    g1 = P.unicycle_turn90(hip_make)
    g1.solve()
    Xa, Ua = g1.get_trajectory()
    assert (Xa[0] == X1[0]).all() and (Ua[0] == U1[0]).all()
    assert g1.get_stats()[0]["iterations_total"] == s1[0]["iterations_total"] == 11
    # every solved instance satisfies the termination criteria
    ok = s1["status"] == 0
    assert ok.mean() > 0.95
    assert (s1["violation"][ok] < 1e-4).all()
    assert (s1["cost_decrease"][ok] < 1e-4).all() and (s1["gradient"][ok] < 1e-2).all()
    # dynamic feasibility: the returned X is the rollout of the returned U
    g.rollout()
    Xr, _ = g.get_trajectory()
    assert np.abs(Xr - X1).max() < 1e-12
    # control bounds and goal
    assert (np.abs(U1[ok]).max(axis=(1, 2)) <= 1.5 + 1e-4).all()


def test_config3_full_batch_against_oracle(P, A, oracle_make, hip_make, oracle_lib):
    """BASELINE config 3 at full size (the bench workload): every one of the 4096 instances walks the same
    schedule on the GPU as in the CPU oracle -- including the ~2 % stragglers that run into the iteration
    limit -- and the solved ones end on the same trajectories."""
    import ctypes
    B = 4096
    o = P.batch_turn90(oracle_make, batch=B, seed=P.SEED_BASE + 3)
    oracle_lib.oracle_set_threads(o._h, ctypes.c_int(len(os.sched_getaffinity(0))))
    o.solve()
    g = P.batch_turn90(hip_make, batch=B, seed=P.SEED_BASE + 3)
    g.solve()
    so, sg = o.get_stats(), g.get_stats()
    same = (so["iterations_total"] == sg["iterations_total"]) & (so["status"] == sg["status"]) & \
           (so["iterations_outer"] == sg["iterations_outer"])
    solved = so["status"] == 0
    # solved instances: exact schedule.  The 99 stragglers ride a line search down to alpha = 2^-19 for ~100
    # iterations; today all 4096 instances match (max |dX| 6e-13) -- two may flip before this fails
    assert same[solved].all(), np.flatnonzero(~same & solved)[:10]
    # explicit counter of tolerated schedule flips (stragglers only; printed with their instance ids, kept in the ledger)
    _ledger.count("schedule flips among the unsolved stragglers (4096 instances)", (~same).sum(), 2, np.flatnonzero(~same))
    assert (~same).sum() <= 2, (~same).sum()
    Xo, Uo = o.get_trajectory()
    Xg, Ug = g.get_trajectory()
    close(Xg[solved], Xo[solved], *XTOL, label="X solved instances")          # measured 1.7e-12
    close(Ug[solved], Uo[solved], *UTOL, label="U solved instances")          # measured 2.5e-12
    close(Xg[~solved & same], Xo[~solved & same], *XTOL, label="X stragglers (same schedule)")  # measured 2.6e-13
    close(sg["cost"][solved], so["cost"][solved], 1e-10, 0.0, label="cost solved instances")


@pytest.mark.parametrize("no_fused", [False, True])
def test_horizon_lengths_and_ragged_batches(P, A, oracle_make, hip_make, no_fused, monkeypatch):
    """Edge cases of the knot loops: the shortest horizons, odd and even N (the forward pass publishes two knots per
    barrier and the persistent kernel's auxiliary wave takes one of them per half-wave: N odd / even end the last pair
    differently), batches that fill neither a wavefront nor a workgroup -- with the persistent tail kernel (a batch
    below the CU count goes there directly) and with the three-kernel sweeps."""
    if no_fused:
        monkeypatch.setenv("ALTRO_HIP_NO_FUSED_SWEEP", "1")  # read when the handle is created
    for N in (1, 2, 3, 7, 51):
        o, g = both(P, P.batch_turn90, oracle_make, hip_make, batch=33, N=N)
        o.solve(); g.solve()
        if N == 1:
            # one step cannot turn the unicycle: the goal constraint is infeasible, the penalty climbs to 1e9 and the
            # line-search test compares cost differences of 1e-10 on costs of 1e4 -- rounding noise decides how many
            # iterations the last outer loops take (scripts/probe_horizons.py).  Statuses and outer counts still agree.
            so, sg = o.get_stats(), g.get_stats()
            assert (so["status"] == sg["status"]).all() and (so["iterations_outer"] == sg["iterations_outer"]).all()
            assert (np.abs(so["iterations_total"] - sg["iterations_total"]) <= 2).all()
        else:
            _compare_full(o, g)  # (round 5: the default bars -- measured X 3.2e-13, U 5e-13, K 3.4e-10 on these horizons)
        assert g.get_timing()["fused_sweeps"] == 0 if no_fused else g.get_timing()["fused_sweeps"] > 0
    for N, batch in ((5, 5), (51, 5), (50, 1)):
        o, g = both(P, P.batch_three_obstacles, oracle_make, hip_make, batch=batch, N=N, dtype=A.F64)
        o.solve(); g.solve()
        _compare_full(o, g)


def test_long_horizon_with_cholesky_restarts(P, A, oracle_make, hip_make):
    """N > 126: the MFMA recursion (k_backward_mfma) buffers 126 knots of gains in LDS between two bulk stores, so a pass of
    160 knots is two chunks.  The obstacle batch has real Cholesky restarts (ilqr.hpp:409-427): a wavefront then runs another
    sweep for the instance that failed while the three others of the wavefront sit it out -- and round 6 found the bulk store
    writing THEIR slots too, which after the first chunk hold other knots' gains (one-chunk passes rewrote the same values).
    Exact schedules against the oracle for every instance."""
    o, g = both(P, P.batch_three_obstacles, oracle_make, hip_make, batch=768, N=160, dtype=A.F64)
    o.solve(); g.solve()
    assert g.get_timing()["sweeps"] > g.get_timing()["fused_sweeps"]  # (batched sweeps ran: the kernel with the chunks)
    _compare_full(o, g)


def test_setters_are_ordered_against_the_solver_stream(P, A, hip_make):
    """The engine's stream does not synchronise with the null stream: a trajectory handed over with SetTrajectory must be
    the one ResetTrajectory restores and the one the next solve starts from, however soon those calls follow (a
    device-to-device copy left on the null stream used to race with them: a rare solve from a half-copied guess)."""
    B, N = 4096, 100
    s = P.batch_turn90(hip_make, batch=B, dtype=A.F64)
    rng = np.random.default_rng(7)
    for rep in range(12):
        U = rng.uniform(-0.5, 0.5, size=(B, N, 2))
        s.set_trajectory(None, U)
        s.reset_trajectory()
        _, Ug = s.get_trajectory()
        assert np.array_equal(Ug, U)
    # the first solve of a fresh solver, issued right behind the setters, equals a later solve from the same guess
    s = P.batch_turn90(hip_make, batch=B, dtype=A.F64)
    s.solve()
    first = (s.get_trajectory()[0], s.get_stats()["iterations_total"].copy())
    s.reset_trajectory()
    s.solve()
    assert np.array_equal(first[1], s.get_stats()["iterations_total"])
    assert np.array_equal(first[0], s.get_trajectory()[0])


@pytest.mark.parametrize("opts", [dict(max_iterations_total=7), dict(max_iterations_inner=4, max_iterations_outer=3), dict()],
                         ids=["total-cap", "inner-outer-caps", "defaults"])
def test_chains_of_sweeps_against_oracle(P, A, oracle_make, hip_make, opts, monkeypatch):
    """A batch of >= 2048 instances is swept as four chains on streams of their own, with one joint persistent launch
    (Engine::Chain): ragged chains (2304 + 37 instances: 640, 640, 640, 421), chains that run into the iteration caps
    before the persistent kernel takes over (the caps bound the sweeps of every chain), and the default schedule."""
    B = 2341
    monkeypatch.setenv("ALTRO_HIP_CHAINS", "4")  # (the default of the first large engine of a process: forced, other tests' handles may be alive)
    o, g = both(P, P.batch_turn90, oracle_make, hip_make, batch=B)
    for s in (o, g):
        s.set_options(**opts)
    o.solve(); g.solve()
    so, sg = _compare_full(o, g)
    if "max_iterations_total" in opts:
        assert so["iterations_total"].max() <= 7 and (so["status"] != 0).any()
    # a second solve from the same guess through the same handle (the chains' counters, lists and events are reused)
    g.reset_trajectory(); g.solve()
    assert np.array_equal(g.get_stats()["iterations_total"], sg["iterations_total"])
    assert np.array_equal(g.get_stats()["status"], sg["status"])
