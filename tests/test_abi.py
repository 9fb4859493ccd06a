"""C-ABI checks that need no GPU: the library loads, exports every symbol include/altro_hip.h
declares, records problem definitions, validates arguments like the reference's ALTRO_ASSERTs, and
FAILS LOUDLY (ALTRO_HIP_ERROR, no CPU fallback) when a compute entry point is called without a
usable HIP device."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "altro_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(altro_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("altro_create", "altro_destroy", "altro_solve_al", "altro_solve_ilqr", "altro_set_lqr_cost",
                 "altro_add_constraint", "altro_get_trajectory", "altro_get_gains", "altro_get_stats",
                 "altro_update_expansions", "altro_backward_pass", "altro_forward_pass", "altro_pack_results_device"):
        assert must in names
    assert len(names) >= 45


def test_library_exports_every_declared_symbol(A):
    lib = A.load_library()
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_oracle_mirrors_the_abi(oracle_lib):
    """The oracle exposes the same entry points with the prefix oracle_ (tests drive both alike)."""
    skip = {"altro_device_info", "altro_get_timing", "altro_register_model_source",
            "altro_solve_al_async", "altro_solve_poll", "altro_wait", "altro_user_model_path"}  # device-side / host-threading / plugin-cache conveniences
    missing = [n for n in declared_functions() if n not in skip and not hasattr(oracle_lib, "oracle_" + n[len("altro_"):])]
    assert not missing, missing


def test_struct_layouts_match_the_header(A):
    # field order / sizes of the ctypes mirrors (a mismatch would corrupt every call)
    assert ctypes.sizeof(A.Desc) == 24
    assert ctypes.sizeof(A.Stats) == 24 + 9 * 8
    assert [f for f, _ in A.Options._fields_][:5] == ["max_iterations_total", "max_iterations_outer",
                                                      "max_iterations_inner", "cost_tolerance", "gradient_tolerance"]
    assert ctypes.sizeof(A.Options) % 8 == 0
    # altro_timing: 6 doubles, 4 ints, 2 long longs, 6 ints, then the device-side sweep loop's: 1 double, 3 ints (+ 4 bytes of
    # padding), 1 long long
    assert ctypes.sizeof(A.Timing) == 6 * 8 + 4 * 4 + 2 * 8 + 6 * 4 + 8 + 3 * 4 + 4 + 8
    assert [f for f, _ in A.Timing._fields_] == ["total_ms", "init_ms", "expansions_ms", "backward_pass_ms", "forward_pass_ms",
                                                "fused_ms", "sweeps", "fused_sweeps", "launches", "sweep_launches",
                                                "instance_iterations",
                                                "fused_instance_iterations", "host_naps", "twin_workgroups", "twin_claims",
                                                "twin_handovers", "fused_workgroup_iterations", "segment_columns",
                                                "loop_ms", "loop_workgroups", "loop_iterations", "loop_handover",
                                                "loop_instance_iterations"]


def test_default_options_are_the_reference_defaults(A):
    # altro/common/solver_options.hpp:23-56
    s = A.BatchSolver(3, 2, 10, 1)
    o = s.default_options()
    assert (o.max_iterations_total, o.max_iterations_outer, o.max_iterations_inner) == (300, 30, 100)
    assert (o.cost_tolerance, o.gradient_tolerance) == (1e-4, 1e-2)
    assert (o.bp_reg_increase_factor, o.bp_reg_initial, o.bp_reg_max, o.bp_reg_min) == (1.6, 0.0, 1e8, 1e-8)
    assert o.bp_reg_fail_threshold == 100 and o.check_forwardpass_bounds == 1
    assert (o.state_max, o.control_max) == (1e8, 1e8)
    assert (o.line_search_max_iterations, o.line_search_lower_bound, o.line_search_upper_bound,
            o.line_search_decrease_factor) == (20, 1e-8, 10.0, 2.0)
    assert (o.constraint_tolerance, o.maximum_penalty, o.initial_penalty, o.reset_duals) == (1e-4, 1e8, 1.0, 1)
    s.set_options(constraint_tolerance=1e-6)
    assert s.get_options().constraint_tolerance == 1e-6


def test_argument_validation(A):
    with pytest.raises(A.AltroError):
        A.BatchSolver(0, 2, 10, 1)  # invalid dimensions
    s = A.BatchSolver(3, 2, 10, 4)
    with pytest.raises(A.AltroError):  # knot range out of bounds (Problem::SetCostFunction asserts)
        s.set_lqr_cost(0, 12, np.eye(3), np.eye(2), np.zeros(3), np.zeros(2))
    with pytest.raises(A.AltroError):  # "Lower bound isn't less than the upper bound." (basic_constraints.hpp:131-136)
        s.add_control_bound(0, 10, [1.0, 1.0], [0.0, 2.0])
    with pytest.raises(A.AltroError):  # ALTRO_ASSERT(rho >= 0), constraint_values.hpp:80
        s.set_penalty(-1.0)
    with pytest.raises(A.AltroError):  # ALTRO_ASSERT(phi >= 1), constraint_values.hpp:85
        s.set_penalty_scaling(0.5)
    # the typed user functors and the device-side packing entry points validate before they touch a device
    with pytest.raises(A.AltroError):
        s.set_user_cost(0, 5, np.zeros(3), type=-1)
    with pytest.raises(A.AltroError):
        s.add_user_constraint(0, 5, np.zeros(2), type=-1)
    with pytest.raises(A.AltroError):
        s.add_user_constraint(0, 12, np.zeros(2), type=0)   # past the last knot
    with pytest.raises(A.AltroError):
        s.pack_trajectory_device(0, 0)


def test_no_cpu_fallback(A, P):
    """Without a GPU every compute entry point must fail with a HIP error -- never silently run on CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device error path cannot be exercised")
    s = P.unicycle_turn90(lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d), batch=2)
    for call in (s.solve, s.solve_ilqr, s.rollout, s.update_expansions, s.get_trajectory):
        with pytest.raises(A.AltroError) as e:
            call()
        assert "(2)" in str(e.value) or "hip" in str(e.value).lower()


def test_missing_library_fails_loudly(A, tmp_path):
    with pytest.raises(A.AltroError):
        A.load_library(str(tmp_path / "libaltro_hip.so"))


def test_get_trajectory_fills_the_callers_buffers(A, P, oracle_make):
    """The Python mirror of altro_get_trajectory writes into arrays the caller hands in (an MPC loop keeps its buffers) and
    refuses arrays that are not the C layout the ABI fills."""
    import numpy as np
    s = P.batch_turn90(oracle_make, batch=3)
    s.rollout()
    X, U = s.get_trajectory()
    X2, U2 = np.zeros_like(X), np.zeros_like(U)
    Xr, Ur = s.get_trajectory(X2, U2)
    assert Xr is X2 and Ur is U2 and np.array_equal(X2, X) and np.array_equal(U2, U)
    import pytest
    with pytest.raises(ValueError):
        s.get_trajectory(np.zeros(X.shape, dtype=np.float32), U2)
    with pytest.raises(ValueError):
        s.get_trajectory(np.zeros(X.shape[::-1]).T, U2)


def test_get_desc_returns_what_the_handle_was_created_with(A):
    """altro_get_desc (round 4): callers that size device buffers for the pack entry points read the dimensions from the
    handle (libaltro_group.so does); works before any device state exists."""
    import ctypes
    s = A.BatchSolver(6, 2, 50, 37, A.F32)
    d = A.Desc()
    assert A.load_library().altro_get_desc(s._h, ctypes.byref(d)) == 0
    assert (d.n, d.m, d.N, d.batch, d.dtype, d.device_id) == (6, 2, 50, 37, A.F32, 0)
    assert A.load_library().altro_get_desc(s._h, None) != 0
