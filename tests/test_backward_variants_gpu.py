"""The backward-pass kernels -- MFMA 4x4x4 (n = 3, m = 2), MFMA 16x16x4 (n = 6 and n = 12: the default there),
cooperative one-instance-per-wavefront on the vector ALUs (ALTRO_HIP_BACKWARD=coop), one-lane-per-instance
VALU (ALTRO_HIP_BACKWARD=valu) -- against each other and against the oracle, including the
restart-on-Cholesky-failure schedule of ilqr.hpp:409-427 on every one of them."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Problems whose Quu + rho I is indefinite until the regularisation has grown (negative entries in R, weak
# terminal weight): the first backward passes fail their Cholesky factorisation and restart.
_BUILDERS = r'''
import numpy as np
def restart_unicycle(A, make, B=6, N=40):
    s = make(3, 2, N, B, A.F64)
    s.set_model(A.MODEL_UNICYCLE)
    s.set_uniform_step(np.float32(0.05))
    xf = np.tile(np.array([1.0, 0.5, 0.3]), (B, 1)) + np.linspace(0, 0.3, B)[:, None]
    R = np.diag([-2e-3, 1e-3])
    s.set_lqr_cost(0, N, np.eye(3) * 1e-3, R, xf, np.zeros(2))
    s.set_lqr_cost(N, N + 1, np.eye(3) * 10.0, R * 0, xf, np.zeros(2))
    s.set_initial_state(np.zeros(3))
    s.set_trajectory(None, np.full((N, 2), 0.05))
    return s
def restart_triple_integrator(A, make, B=6, N=30):
    s = make(6, 2, N, B, A.F64)
    s.set_model(A.MODEL_TRIPLE_INTEGRATOR, [2])
    s.set_uniform_step(np.float32(0.1))
    xf = np.zeros((B, 6)); xf[:, 0] = 1.0 + 0.1 * np.arange(B); xf[:, 1] = 2.0
    R = np.diag([-2e-3, 1e-3])
    s.set_lqr_cost(0, N, np.eye(6) * 1e-3, R, xf, np.zeros(2))
    s.set_lqr_cost(N, N + 1, np.eye(6) * 1.0, R * 0, xf, np.zeros(2))
    s.set_initial_state(-xf)
    s.set_trajectory(None, np.zeros((N, 2)))
    return s
def restart_quadrotor(A, make, B=4, N=40):
    s = make(12, 4, N, B, A.F64)
    s.set_model(A.MODEL_QUADROTOR12)
    s.set_uniform_step(np.float32(0.02))
    xf = np.zeros((B, 12)); xf[:, 0] = 0.5 + 0.1 * np.arange(B); xf[:, 2] = 0.3
    R = np.diag([-1e-4, 1e-4, 1e-4, -5e-5])
    s.set_lqr_cost(0, N, np.eye(12) * 1e-4, R, xf, np.zeros(4))
    s.set_lqr_cost(N, N + 1, np.eye(12) * 1.0, R * 0, xf, np.zeros(4))
    s.set_initial_state(np.zeros(12))
    s.set_trajectory(None, np.zeros((N, 4)))
    return s
RESTART = {"unicycle": restart_unicycle, "triple_integrator": restart_triple_integrator, "quadrotor12": restart_quadrotor}
'''
_ns = {}
exec(_BUILDERS, _ns)
RESTART = _ns["RESTART"]

# Child process: the backward kernel is chosen when the engine is created (environment variable), so each
# variant runs in its own interpreter and dumps everything the comparison needs.
_CHILD = r'''
import importlib, sys
sys.path.insert(0, %(root)r)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
''' + _BUILDERS + r'''
out = {}
def dump(tag, s):
    st = s.get_stats()
    X, U = s.get_trajectory()
    K, d = s.get_gains()
    out[tag + "_X"], out[tag + "_U"], out[tag + "_K"], out[tag + "_d"] = X, U, K, d
    for f in ("status", "iterations_total", "iterations_outer", "cost", "regularization"):
        out[tag + "_" + f] = st[f]
# whole solves of the two models the cooperative kernel serves (configs 2 and 5, plus the constrained variants)
s = P.batch_triple_integrator(make, batch=48); s.solve_ilqr(); dump("ti_ilqr", s)
s = P.triple_integrator(make, batch=1, constraints=True); s.solve(); dump("ti_al", s)
s = P.batch_quadrotor12(make, batch=12, dtype=A.F64); s.solve(); dump("quad_al", s)
# one backward pass with the cost-to-go recorded
s = P.batch_quadrotor12(make, batch=4, dtype=A.F64)
s.set_record_ctg(True); s.rollout(); s.update_expansions(); s.backward_pass()
Pm, pv = s.get_ctg(); K, d = s.get_gains()
out["quad_step_P"], out["quad_step_p"], out["quad_step_K"], out["quad_step_d"] = Pm, pv, K, d
# the restart schedule
for name, build in RESTART.items():
    s = build(A, make); s.set_options(max_iterations_inner=4); s.solve_ilqr(); dump("restart_" + name, s)
np.savez(sys.argv[1], **out)
'''


def _child(tmp_path, tag, env_extra):
    out = str(tmp_path / f"{tag}.npz")
    subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}, out], check=True,
                   env=dict(os.environ, **env_extra), timeout=900)
    return np.load(out)


@pytest.fixture(scope="module")
def variants(tmp_path_factory):
    d = tmp_path_factory.mktemp("bwd")
    return (_child(d, "default", {}), _child(d, "coop", {"ALTRO_HIP_BACKWARD": "coop"}),
            _child(d, "valu", {"ALTRO_HIP_BACKWARD": "valu"}))


def test_coop_backward_is_bitwise_the_valu_backward(variants):
    """k_backward_coop performs the operations of riccati_q / riccati_gains in the same order and type
    (altro_kernels.hpp, header of k_backward_coop): n = 6 and n = 12 engines must return the same bits
    whichever of the two kernels ran."""
    _, coop, valu = variants
    for k in coop.files:
        if k.startswith(("ti_", "quad_", "restart_triple", "restart_quad")):
            assert np.array_equal(coop[k], valu[k]), k


def test_mfma_backward_agrees_with_valu(variants):
    """The matrix-core kernels (4x4x4 for the unicycle, 16x16x4 for n = 6 / 12) associate the products differently
    from the vector-ALU kernels: same schedule (iteration counts, statuses, regularisation), values to rounding."""
    mfma, _, valu = variants
    for tag in ("restart_unicycle", "restart_triple_integrator", "restart_quadrotor12", "ti_ilqr", "ti_al", "quad_al"):
        for f in ("status", "iterations_total", "regularization"):
            assert np.array_equal(mfma[f"{tag}_{f}"], valu[f"{tag}_{f}"]), (tag, f)
    for tag in ("restart_unicycle", "restart_triple_integrator", "restart_quadrotor12", "ti_ilqr", "ti_al"):
        assert np.allclose(mfma[tag + "_X"], valu[tag + "_X"], rtol=1e-9, atol=1e-10), tag  # (measured 4.1e-12 abs)
    # one backward pass of the 12-state model from the same expansions: gains and cost-to-go of all 200 knots
    for key, tol in (("quad_step_K", 1e-9), ("quad_step_d", 1e-9), ("quad_step_P", 1e-10), ("quad_step_p", 1e-10)):
        a, b = mfma[key], valu[key]
        assert np.abs(a - b).max() <= tol * np.abs(b).max(), (key, np.abs(a - b).max(), np.abs(b).max())


@pytest.mark.parametrize("name", list(RESTART))
def test_cholesky_restart_against_oracle(A, oracle_make, variants, name):
    """ilqr.hpp:409-427 on every backward kernel: default (MFMA 4x4x4 for the unicycle, MFMA 16x16x4 for
    n = 6 / 12), cooperative and VALU, against the oracle in fp64."""
    o = RESTART[name](A, oracle_make)
    o.set_options(max_iterations_inner=4)
    o.solve_ilqr()
    so = o.get_stats()
    assert (so["regularization"] > 1e-8).all()  # the restart path was really taken
    Xo, Uo = o.get_trajectory()
    for v in variants:
        for f in ("status", "iterations_total"):
            assert np.array_equal(v[f"restart_{name}_{f}"], so[f]), f
        assert np.allclose(v[f"restart_{name}_regularization"], so["regularization"], rtol=1e-12)
        assert np.allclose(v[f"restart_{name}_X"], Xo, rtol=1e-9, atol=1e-11)  # (measured: X 1.8e-15, U 6.4e-14 abs)
        assert np.allclose(v[f"restart_{name}_U"], Uo, rtol=1e-9, atol=1e-11)
