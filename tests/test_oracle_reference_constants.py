"""Pins the CPU oracle (oracle/altro_oracle.cpp) to the reference's own known-answer tests.

Every test below replays one gtest of /root/reference through the oracle's C API and compares with
the constant that gtest asserts (tests/golden/reference_constants.json holds the values with their
reference test file:line).  These run on CPU (`-m "not gpu"`).
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_constants.json")) as f:
    K = json.load(f)


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


# ---- triple integrator (test/ilqr/ilqr_test.cpp) -----------------------------------------------------
def _rollout_zero_controls(s):
    x0 = np.array([-1.0, -2.0, 0, 0, 0, 0])
    s.set_trajectory(np.tile(x0, (11, 1)), np.zeros((10, 2)))


def test_K1_K2_triple_rk4_jacobian(P, oracle_make):
    # test/ilqr/ilqr_test.cpp:159-180 and test/problem/triple_integrator_test.cpp:87-133
    s = P.triple_integrator(oracle_make)
    _rollout_zero_controls(s)
    s.update_expansions()
    h = float(np.float32(0.1))
    Ac = np.zeros((6, 6))
    Bc = np.zeros((6, 2))
    for i in range(2):
        Ac[i, i + 2] = 1
        Ac[i + 2, i + 4] = 1
        Bc[i + 4, i] = 1
    # closed-form power series of the discretised linear system (RK4 is exact to 4th order here)
    Ad = np.eye(6) + Ac * h + Ac @ Ac * h**2 / 2 + Ac @ Ac @ Ac * h**3 / 6
    Bd = (np.eye(6) * h + Ac * h**2 / 2 + Ac @ Ac * h**3 / 6 + Ac @ Ac @ Ac * h**4 / 24) @ Bc
    for k in (0, 5, 9):
        e = s.get_expansion(k)
        assert rel(e["A"][0], Ad) < 1e-12
        assert rel(e["B"][0], Bd) < 1e-12
    A_test = np.eye(6)
    for i in range(2):
        A_test[i, i + 2] = 0.1
        A_test[i + 2, i + 4] = 0.1
        A_test[i, i + 4] = 0.005
    assert rel(s.get_expansion(0)["A"][0], A_test) < 1e-6  # isApprox 1e-6


def test_K3_triple_backward_pass(P, oracle_make):
    s = P.triple_integrator(oracle_make)
    _rollout_zero_controls(s)
    s.update_expansions()
    s.backward_pass()
    _, p = s.get_ctg()
    _, d = s.get_gains()
    assert rel(p[0, 0], K["K3_triple_bp_p0"]["value"]) < 1e-4
    assert rel(d[0, 0], K["K3_triple_bp_d0"]["value"]) < 1e-4


def test_K4_K5_K6_triple_steps(P, oracle_make):
    s = P.triple_integrator(oracle_make)
    s.rollout()
    assert s.cost()[0] == K["K4_triple_initial_cost"]["value"]  # EXPECT_DOUBLE_EQ(100 + 1e6)
    s.update_expansions(); s.backward_pass(); s.forward_pass()
    J1 = s.cost()[0]
    assert abs(J1 - K["K5_triple_fp_cost"]["value"]) < 1e-3
    s.update_expansions(); s.backward_pass(); s.forward_pass()
    J2 = s.cost()[0]
    assert J1 - J2 < 1e-10
    Kg, d = s.get_gains()
    assert rel(Kg[0, 0], K["K6_triple_K0"]["value"]) < 1e-4
    assert np.abs(d[0]).max() < 1e-8
    s2 = P.triple_integrator(oracle_make)
    s2.solve_ilqr()
    st = s2.get_stats()[0]
    assert st["status"] == 0 and st["iterations_inner"] == 2
    assert rel(s2.get_gains()[0][0, 0], K["K6_triple_K0"]["value"]) < 1e-3


def test_K7_K8_triple_auglag(P, oracle_make):
    s = P.triple_integrator(oracle_make, goal_only=True)
    s.rollout()
    assert s.cost()[0] == K["K7_triple_al_initial_cost"]["value"]
    s.update_expansions(); s.backward_pass()
    _, p = s.get_ctg()
    _, d = s.get_gains()
    assert rel(p[0, 0], K["K8_triple_al_bp_p0"]["value"]) < 1e-4
    assert rel(d[0, 0], K["K8_triple_al_bp_d0"]["value"]) < 1e-4
    s.forward_pass()
    assert abs(s.cost()[0] - K["K8_triple_al_fp_cost"]["value"]) < 1e-3
    s.update_expansions(); s.backward_pass(); s.forward_pass()
    X, _ = s.get_trajectory()
    assert np.abs(X[0, -1] - np.array([1, 2, 0, 0, 0, 0])).max() < 0.01


def test_K7_auglag_cost_expansion(P, oracle_make):
    # test/ilqr/ilqr_test.cpp:348-379: rho = 123, lambda = 1.5 on the goal constraint
    s = P.triple_integrator(oracle_make, goal_only=True)
    s.rollout()
    s.set_penalty(123.0)
    s.set_duals(np.full((1, 6), 1.5))
    s.update_expansions()
    e = s.get_expansion(10)
    x0, xf = np.array([-1.0, -2, 0, 0, 0, 0]), np.array([1.0, 2, 0, 0, 0, 0])
    lam_bar = 1.5 - 123.0 * (x0 - xf)
    assert rel(e["lxx"][0], np.eye(6) * 1e5 + np.eye(6) * 123.0) < 1e-14
    assert rel(e["lx"][0], 1e5 * (x0 - xf) - lam_bar) < 1e-14


def test_K24_triple_problem(P, oracle_make):
    s = P.triple_integrator(oracle_make)
    s.rollout(); s.solve_ilqr()
    st = s.get_stats()[0]
    assert st["status"] == 0 and st["iterations_total"] == 2
    assert st["cost_decrease"] < 1e-4 and st["gradient"] < 1e-2
    s = P.triple_integrator(oracle_make, constraints=True)
    assert s.num_constraints(0) == 4 and s.num_constraints(10) == 6
    s.solve()
    st = s.get_stats()[0]
    X, U = s.get_trajectory()
    assert st["status"] == 0 and st["violation"] < 1e-4
    assert np.abs(X[0, -1] - np.array([1, 2, 0, 0, 0, 0])).max() < 1e-4
    assert rel(U[0, 0], [100, 200]) < 1e-8 and rel(U[0, -1], [100, 200]) < 1e-8


# ---- unicycle (test/ilqr/unicycle_ilqr_test.cpp, test/augmented_lagrangian/auglag_test.cpp) --------------
def test_K9_to_K12_unicycle_steps(P, oracle_make):
    s = P.unicycle_turn90(oracle_make, constraints=False)
    s.rollout()
    assert abs(s.cost()[0] - K["K9_unicycle_initial_cost"]["value"]) < 1e-5
    s.update_expansions(); s.backward_pass()
    _, p = s.get_ctg()
    _, d = s.get_gains()
    assert rel(p[0, 0], K["K10_unicycle_bp_p0"]["value"]) < 1e-5
    assert rel(d[0, 0], K["K10_unicycle_bp_d0"]["value"]) < 1e-5
    J0 = s.cost()[0]
    s.forward_pass()
    assert s.cost()[0] < J0
    assert s.get_history(0, "alpha")[0] == K["K11_unicycle_alpha0"]["value"]
    s.update_expansions(); s.backward_pass()
    _, p = s.get_ctg()
    _, d = s.get_gains()
    assert rel(p[0, 0], K["K12_unicycle_bp2_p0"]["value"]) < 1e-5
    assert rel(d[0, 0], K["K12_unicycle_bp2_d0"]["value"]) < 1e-5
    s.forward_pass()
    assert s.cost()[0] - K["K12_unicycle_fp2_cost_upper"]["value"] < 1e-5


def test_K11_auglag_forward_pass_alpha(P, oracle_make):
    s = P.unicycle_turn90(oracle_make, constraints=True)
    s.rollout(); s.update_expansions(); s.backward_pass()
    J0 = s.cost()[0]
    s.forward_pass()
    assert s.cost()[0] < J0
    assert s.get_history(0, "alpha")[0] == 0.0625


def test_K13_unicycle_full_ilqr(P, oracle_make):
    s = P.unicycle_turn90(oracle_make, constraints=False)
    s.rollout(); s.solve_ilqr()
    st = s.get_stats()[0]
    assert st["iterations_inner"] == K["K13_unicycle_ilqr"]["iterations"]
    assert st["status"] == 0
    assert abs(s.cost()[0] - K["K13_unicycle_ilqr"]["cost"]) < 1e-5
    assert st["gradient"] < 1e-2


def test_K14_K18_auglag_ilqr_and_two_solves(P, oracle_make):
    s = P.unicycle_turn90(oracle_make, constraints=True)
    s.rollout()
    J0, v0 = s.cost()[0], s.get_max_violation()[0]
    s.solve_ilqr()
    J, viol = s.cost()[0], s.get_max_violation()[0]
    _, U = s.get_trajectory()
    k14 = K["K14_unicycle_al_ilqr"]
    assert abs(J - k14["cost"]) / k14["cost"] < 1e-6
    assert abs(viol - k14["violation"]) / k14["violation"] < 1e-6
    assert abs((np.abs(U[0]).max() - 1.5) - k14["violation"]) / k14["violation"] < 1e-6
    assert s.get_stats()[0]["iterations_inner"] == k14["iterations"]
    assert J < J0 and viol < v0
    s.update_duals(); s.update_penalties()
    assert s.cost()[0] > J
    s.solve_ilqr()
    viol = s.max_violation()[0]
    assert abs(viol - K["K18_two_solves"]["violation"]) / K["K18_two_solves"]["violation"] < 0.1
    assert s.get_stats()[0]["iterations_inner"] == K["K18_two_solves"]["iterations"]


def test_K19_full_al_solve_and_solve_twice(P, oracle_make):
    k = K["K19_full_al"]
    s = P.unicycle_turn90(oracle_make, constraints=True)
    s.set_options(constraint_tolerance=k["constraint_tolerance"])
    for _ in range(2):  # SolveTwice, auglag_test.cpp:353-380: same result after resetting the guess
        s.set_trajectory(None, np.full((100, 2), 0.1))
        s.solve()
        st = s.get_stats()[0]
        assert st["iterations_total"] == k["iterations_total"]
        assert st["iterations_outer"] == k["iterations_outer"]
        assert st["status"] == 0
        assert abs(s.cost()[0] - k["cost"]) < k["atol"]
        assert s.get_max_violation()[0] < k["constraint_tolerance"]


def test_K20_constraint_layout(P, oracle_make):
    # auglag_test.cpp:382-399: first constraint is the control bound at k=0, the goal sits at k=N
    s = P.unicycle_turn90(oracle_make, constraints=True)
    assert s.num_constraints(0) == 4 and s.num_constraints(100) == 3
    assert s.num_constraints() == 4 * 100 + 3
    s.solve()
    c = s.get_constraint_values()[0]
    viol_goal = np.abs(c[-3:]).max()
    viol_bounds = np.maximum(c[:-3], 0).max()
    assert viol_goal > viol_bounds  # sorted-by-violation puts the goal constraint first


def test_K21_K22_K23_three_obstacles(P, oracle_make):
    k = K["K21_three_obstacles_costs"]
    s = P.unicycle_three_obstacles(oracle_make, constraints=False)
    s.rollout()
    assert abs(s.cost()[0] - k["plain"]) < 1e-6
    s = P.unicycle_three_obstacles(oracle_make, constraints=True)
    s.rollout()
    assert abs(s.cost()[0] - k["al_rho1"]) < 1e-6
    s.set_penalty(10.0)
    assert abs(s.cost()[0] - k["al_rho10"]) < 1e-6
    # SolveOneStep (example_unicycle_test.cpp:52-67)
    s.solve_ilqr(); s.update_duals(); s.update_penalties()
    lamN = s.get_duals()[0][-3:]
    assert rel(lamN, K["K22_three_obstacles_lambdaN"]["value"]) < 1e-6
    # SolveConstrained (example_unicycle_test.cpp:69-89); SetPenalty(10) is a no-op (quirk Q8)
    s = P.unicycle_three_obstacles(oracle_make, constraints=True)
    s.rollout(); s.set_penalty(10.0); s.solve()
    st = s.get_stats()[0]
    X, _ = s.get_trajectory()
    for cx, cy, r in P.THREE_OBSTACLE_CIRCLES:
        dist = np.hypot(X[0, :, 0] - cx, X[0, :, 1] - cy) - r
        assert dist.min() > -1e-3
    assert st["status"] == 0
    assert s.max_violation()[0] < 1e-4
    assert st["cost_decrease"] < 1e-4 and st["gradient"] < 1e-2
    assert st["iterations_total"] == 50 and st["iterations_outer"] == 5


def test_K15_K16_K17_al_cost_terms(P, oracle_make):
    # auglag_test.cpp:48-93,150-196: value of the AL term and FD check of gradient / Hessian
    s = P.unicycle_turn90(oracle_make, constraints=True, N=4)
    rho = 1.1
    s.set_penalty(rho)
    X = np.zeros((5, 3))
    U = np.zeros((4, 2))
    U[0] = [2.0, 0.1]  # v violates the upper bound 1.5 by 0.5
    s.set_trajectory(X, U)
    J_nom = (P.unicycle_turn90(oracle_make, constraints=False, N=4))
    J_nom.set_trajectory(X, U)
    base = J_nom.cost()[0]
    assert np.isclose(s.cost()[0] - base - 0.5 * rho * (X[4] - [1.5, 1.5, np.pi / 2]) @ (X[4] - [1.5, 1.5, np.pi / 2]),
                      0.5 * rho * 0.5**2, rtol=1e-13)
    assert np.allclose(s.get_penalties(), rho)
    s.set_penalty_scaling(3.0)
    s.update_penalties()
    assert np.allclose(s.get_penalties(), 3.0 * rho)
    # gradient / Gauss-Newton Hessian of knot 0 vs finite differences of the knot cost
    s.update_expansions()
    e = s.get_expansion(0)
    eps = 1e-6
    g = np.zeros(2)
    for j in range(2):
        Up, Um = U.copy(), U.copy()
        Up[0, j] += eps
        Um[0, j] -= eps
        s.set_trajectory(X, Up); s.cost(); cp = s.get_knot_costs()[0, 0]
        s.set_trajectory(X, Um); s.cost(); cm = s.get_knot_costs()[0, 0]
        g[j] = (cp - cm) / (2 * eps)
    assert rel(e["lu"][0], g) < 1e-4
