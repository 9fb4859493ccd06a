"""CPU: what the twin workgroups and the segments of the GPU path rely on, pinned on the oracle (= the reference's semantics).

An iteration whose line search rejects every trial changes NOTHING of an instance but its counters and its regularisation,
and the regularisation follows a rule known in advance: the backward pass ends with DecreaseRegularization (ilqr.hpp:440), the
failed line search with IncreaseRegularization (ilqr.hpp:550, both ilqr.hpp:770-786).  So the state entering iteration j + L
of a rejection streak is known at iteration j -- which is what lets the persistent kernel hand the second half of a streak to
a twin workgroup and the batched sweeps run a streak as four segments side by side; on the GPU every
hand-over is verified bit for bit, here the premise itself is checked on the stragglers of BASELINE configs[2]."""
import numpy as np


def _increase(o, rho, drho):  # iLQR::IncreaseRegularization, ilqr.hpp:770-777
    drho = max(drho * o.bp_reg_increase_factor, o.bp_reg_increase_factor)
    rho = min(max(rho * drho, o.bp_reg_min), o.bp_reg_max)
    return rho, drho


def _decrease(o, rho, drho):  # iLQR::DecreaseRegularization, ilqr.hpp:779-786
    drho = min(drho / o.bp_reg_increase_factor, 1.0 / o.bp_reg_increase_factor)
    rho = min(max(rho * drho, o.bp_reg_min), o.bp_reg_max)
    return rho, drho


def test_a_rejection_streak_changes_only_counters_and_regularisation(P, oracle_make):
    s = P.batch_turn90(oracle_make, batch=512, seed=P.SEED_BASE + 3)  # (the batch of tests/test_fused_gpu.py's twin test)
    s.set_record_history(301)
    s.solve()
    o = s.get_options()
    st = s.get_stats()
    stragglers = np.nonzero(st["iterations_total"] >= 100)[0]
    assert 8 <= len(stragglers) <= 24, len(stragglers)  # ~2 % of the batch
    assert (st["status"][stragglers] != 0).all()        # none of them ends kSolved
    for b in stragglers:
        alpha = s.get_history(int(b), "alpha")
        cost = s.get_history(int(b), "cost")
        dec = s.get_history(int(b), "cost_decrease")
        reg = s.get_history(int(b), "regularization")
        assert len(alpha) == st["iterations_total"][b] + 1  # (row 0: the initial cost)
        # the streak: the trailing run of iterations without an accepted step (their rows repeat the step length and the
        # improvement ratio of the last accepted one: nothing logs a rejection) ...
        rejected = dec == 0.0
        first = len(alpha) - 1
        while first > 1 and rejected[first - 1]:
            first -= 1
        streak = np.arange(first, len(alpha))
        # ... lasts until max_iterations_inner ends the inner solve (ilqr.hpp:600-611): that is where the time of a batch goes
        assert len(streak) >= o.max_iterations_inner - 5, (b, len(streak))
        # nothing moves: the cost is the same bits in every row, the decrease is exactly zero
        assert (cost[streak] == cost[streak[0]]).all() and (alpha[streak] == alpha[streak[0]]).all()
        # the logged regularisation (after the failed line search raised it) follows the rule from its second row on: every
        # iteration lowers it behind its backward pass and raises it again behind its line search
        rho, drho = reg[streak[1]], o.bp_reg_increase_factor  # (one rejection behind a successful backward pass: drho = the factor)
        for j in streak[2:]:
            rho, drho = _decrease(o, rho, drho)
            rho, drho = _increase(o, rho, drho)
            assert rho == reg[j], (b, j, rho, reg[j])


def test_the_plateau_of_the_obstacle_batch_is_made_of_such_streaks(P, A, oracle_make):
    """BASELINE configs[3] (jittered obstacles; here with fp64 records: the history is logged in the record type): a quarter of the instances spend ~100 consecutive iterations in
    one streak somewhere INSIDE their solve (the plateau the segments pack), and the rule holds there."""
    s = P.batch_three_obstacles(oracle_make, batch=256, dtype=A.F64)
    s.set_record_history(301)
    s.solve()
    o = s.get_options()
    st = s.get_stats()
    long_streaks = 0
    for b in range(256):
        dec = s.get_history(b, "cost_decrease")
        reg = s.get_history(b, "regularization")
        cost = s.get_history(b, "cost")
        # longest run of rows without a cost decrease
        best, cur, end = 0, 0, 0
        for j in range(1, len(dec)):
            cur = cur + 1 if dec[j] == 0.0 else 0
            if cur > best:
                best, end = cur, j
        if best < o.max_iterations_inner - 5:
            continue
        long_streaks += 1
        streak = np.arange(end - best + 1, end + 1)
        assert (cost[streak] == cost[streak[0]]).all()
        rho, drho = reg[streak[1]], o.bp_reg_increase_factor
        for j in streak[2:]:
            rho, drho = _decrease(o, rho, drho)
            rho, drho = _increase(o, rho, drho)
            assert rho == reg[j], (b, j, rho, reg[j])
    assert 0.15 * 256 <= long_streaks <= 0.40 * 256, long_streaks
    assert abs((st["status"] == 0).mean() - 0.74) < 0.08
