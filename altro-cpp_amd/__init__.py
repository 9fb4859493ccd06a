"""altro-cpp_amd: ctypes host binding of libaltro_hip.so, the MI355X-native batched AL-iLQR solver.

The directory name carries a hyphen (fixed by the project layout), so it is imported through
``__graft_entry__.load_package()`` / ``tests/conftest.py`` under the module name ``altro_cpp_amd``.

This module is a *binding*: all arithmetic happens inside the HIP library behind the C-ABI declared
in ``include/altro_hip.h``.  There is no CPU fallback; loading fails loudly when the library has not
been built (``python -c 'import __graft_entry__ as g; g.build()'``).

``BatchSolver`` mirrors the method names of the reference's
``altro::augmented_lagrangian::AugmentedLagrangianiLQR<n,m>`` (altro/augmented_lagrangian/al_solver.hpp:28-224)
and ``altro::ilqr::iLQR<n,m>`` (altro/ilqr/ilqr.hpp:47-813) with an added batch dimension.
The class is generic over (library, symbol prefix) so the parity tests can drive the CPU oracle
(`oracle/`, prefix ``oracle_``) through the very same Python code; the product itself never does.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ALTRO_HIP_LIB: load another build of the library (experiments: scripts/README.md); still no CPU fallback
LIB_PATH = os.environ.get("ALTRO_HIP_LIB") or os.path.join(_HERE, "csrc", "libaltro_hip.so")

# --- enums (include/altro_hip.h) -------------------------------------------------------------------
OK, INVALID_ARG, HIP_ERROR, NOT_READY, UNSUPPORTED = 0, 1, 2, 3, 4
F64, F32 = 0, 1
MODEL_UNICYCLE, MODEL_TRIPLE_INTEGRATOR, MODEL_QUADROTOR12 = 1, 2, 3
CON_GOAL, CON_CONTROL_BOUND, CON_CIRCLE, CON_USER = 1, 2, 3, 4
# altro::SolverStatus, altro/common/solver_stats.hpp:20-31
(SOLVED, UNSOLVED, STATE_LIMIT, CONTROL_LIMIT, COST_INCREASE, MAX_ITERATIONS, MAX_OUTER_ITERATIONS,
 MAX_INNER_ITERATIONS, MAX_PENALTY, BACKWARD_PASS_REGULARIZATION_FAILED) = range(10)
HISTORY_FIELDS = ("cost", "alpha", "improvement_ratio", "gradient", "cost_decrease",
                  "regularization", "violations", "max_penalty")


class Desc(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("N", C.c_int), ("batch", C.c_int),
                ("dtype", C.c_int), ("device_id", C.c_int)]


class Options(C.Structure):
    """altro::SolverOptions (altro/common/solver_options.hpp:19-57)."""
    _fields_ = [
        ("max_iterations_total", C.c_int), ("max_iterations_outer", C.c_int),
        ("max_iterations_inner", C.c_int), ("cost_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double), ("bp_reg_increase_factor", C.c_double),
        ("bp_reg_enable", C.c_int), ("bp_reg_initial", C.c_double), ("bp_reg_max", C.c_double),
        ("bp_reg_min", C.c_double), ("bp_reg_fail_threshold", C.c_int),
        ("check_forwardpass_bounds", C.c_int), ("state_max", C.c_double),
        ("control_max", C.c_double), ("line_search_max_iterations", C.c_int),
        ("line_search_lower_bound", C.c_double), ("line_search_upper_bound", C.c_double),
        ("line_search_decrease_factor", C.c_double), ("constraint_tolerance", C.c_double),
        ("maximum_penalty", C.c_double), ("initial_penalty", C.c_double),
        ("reset_duals", C.c_int), ("profiler_enable", C.c_int),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("status", C.c_int), ("status_ilqr", C.c_int), ("iterations_inner", C.c_int),
        ("iterations_outer", C.c_int), ("iterations_total", C.c_int), ("reserved", C.c_int),
        ("cost", C.c_double), ("initial_cost", C.c_double), ("cost_decrease", C.c_double),
        ("gradient", C.c_double), ("violation", C.c_double), ("max_penalty", C.c_double),
        ("alpha", C.c_double), ("regularization", C.c_double), ("improvement_ratio", C.c_double),
    ]


STATS_DTYPE = np.dtype([(name, np.int32 if t is C.c_int else np.float64) for name, t in Stats._fields_])
assert STATS_DTYPE.itemsize == C.sizeof(Stats)


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("init_ms", C.c_double), ("expansions_ms", C.c_double),
                ("backward_pass_ms", C.c_double), ("forward_pass_ms", C.c_double), ("fused_ms", C.c_double),
                ("sweeps", C.c_int), ("fused_sweeps", C.c_int), ("launches", C.c_int), ("sweep_launches", C.c_int),
                ("instance_iterations", C.c_longlong), ("fused_instance_iterations", C.c_longlong),
                ("host_naps", C.c_int), ("twin_workgroups", C.c_int),
                ("twin_claims", C.c_int), ("twin_handovers", C.c_int),
                ("fused_workgroup_iterations", C.c_int), ("segment_columns", C.c_int),
                ("loop_ms", C.c_double), ("loop_workgroups", C.c_int), ("loop_iterations", C.c_int),
                ("loop_handover", C.c_int), ("loop_instance_iterations", C.c_longlong)]


class AltroError(RuntimeError):
    pass


_lib_cache = {}


def load_library(path=None):
    """dlopen libaltro_hip.so.  Raises (never falls back) when the HIP extension is missing."""
    path = path or LIB_PATH
    if path not in _lib_cache:
        if not os.path.exists(path):
            raise AltroError(
                f"{path} not found: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        _lib_cache[path] = C.CDLL(path)
    return _lib_cache[path]


MODEL_USER_BASE = 1000


def register_model_source(name, source, check_jacobian=True, _lib=None, _prefix="altro_"):
    """User-defined dynamics (altro_register_model_source, include/altro_hip.h): `source` defines
    ``struct UserModel { static constexpr int n, m; f(x, u, xdot); jac(x, u, J); }``.  Compiles (or loads from the
    on-disk cache) the plugin and returns the model kind for ``BatchSolver.set_model``."""
    lib = _lib if _lib is not None else load_library()
    f = getattr(lib, _prefix + "register_model_source")
    f.restype = C.c_int
    kind = C.c_int(0)
    st = f(name.encode(), source.encode(), C.c_int(1 if check_jacobian else 0), C.byref(kind))
    if st != OK:
        g = getattr(lib, _prefix + "last_error")
        g.restype = C.c_char_p
        g.argtypes = [C.c_void_p]
        msg = g(None)
        raise AltroError(f"{_prefix}register_model_source failed ({st}): {msg.decode() if msg else ''}")
    return kind.value


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def user_model_path(kind, _lib=None):
    """Path of the cached plugin (.so) of a registered user model."""
    lib = _lib if _lib is not None else load_library()
    buf = C.create_string_buffer(4096)
    f = lib.altro_user_model_path
    f.restype = C.c_int
    if f(C.c_int(kind), buf, C.c_int(len(buf))) < 0:
        raise AltroError(f"unknown user model kind {kind}")
    return buf.value.decode()


class BatchSolver:
    """Batched AL-iLQR solver handle (one device, one stream)."""

    def __init__(self, n, m, N, batch=1, dtype=F64, device_id=0, _lib=None, _prefix="altro_"):
        self._lib = _lib if _lib is not None else load_library()
        self._p = _prefix
        self.n, self.m, self.N, self.batch, self.dtype = int(n), int(m), int(N), int(batch), int(dtype)
        self._h = C.c_void_p()
        d = Desc(self.n, self.m, self.N, self.batch, self.dtype, int(device_id))
        f = self._fn("create")
        f.restype = C.c_int
        st = f(C.byref(d), C.byref(self._h))
        if st != OK:
            raise AltroError(f"{self._p}create failed ({st}): {self._errmsg(None)}")

    # -- plumbing ---------------------------------------------------------------------------------
    def _fn(self, name):
        return getattr(self._lib, self._p + name)

    def _errmsg(self, h):
        f = self._fn("last_error")
        f.restype = C.c_char_p
        f.argtypes = [C.c_void_p]
        s = f(h)
        return s.decode() if s else ""

    def _call(self, name, *args):
        f = self._fn(name)
        f.restype = C.c_int
        st = f(self._h, *args)
        if st != OK:
            raise AltroError(f"{self._p}{name} failed ({st}): {self._errmsg(self._h)}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            f = self._fn("destroy")
            f.restype = None
            f.argtypes = [C.c_void_p]
            f(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- problem definition (altro::problem::Problem, problem.hpp:113-202) -------------------------
    def set_model(self, kind, params=()):
        p = _f64(list(params)) if len(params) else None
        self._call("set_model", C.c_int(kind), _dp(p), C.c_int(0 if p is None else p.size))

    def set_knot_models(self, model_of_knot):
        """Problem::SetDynamics(model, k) with models that differ along the horizon (problem.hpp:155-166): index of knot
        k's model (k = 0 .. N-1) in the user source's ALTRO_USER_MODELS list."""
        km = np.ascontiguousarray(model_of_knot, dtype=np.int32)
        self._call("set_knot_models", km.ctypes.data_as(C.POINTER(C.c_int)), C.c_int(km.size))

    def set_uniform_step(self, h):
        self._call("set_uniform_step", C.c_float(np.float32(h)))

    def set_steps(self, hk):
        """Trajectory::SetStep(k, h), k = 0 .. N-1 (trajectory.hpp:120): 32-bit float steps."""
        hk = np.ascontiguousarray(hk, dtype=np.float32)
        self._call("set_steps", hk.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(hk.size))

    def set_times(self, tk):
        """Trajectory::SetTime(k, t), k = 0 .. N (trajectory.hpp:119)."""
        tk = np.ascontiguousarray(tk, dtype=np.float32)
        self._call("set_times", tk.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(tk.size))

    def get_steps(self):
        hk = np.zeros(self.N, dtype=np.float32)
        tk = np.zeros(self.N + 1, dtype=np.float32)
        self._call("get_steps", hk.ctypes.data_as(C.POINTER(C.c_float)), tk.ctypes.data_as(C.POINTER(C.c_float)))
        return hk, tk

    def set_lqr_cost(self, k_begin, k_end, Q, R, xref, uref):
        Q, R, xref, uref = _f64(Q), _f64(R), _f64(xref), _f64(uref)
        # numpy matrices are row-major; Q and R are symmetric in every caller, but transpose anyway
        Qc = np.ascontiguousarray(Q.T)
        Rc = np.ascontiguousarray(R.T)
        per = (1 if xref.ndim == 2 else 0) | (2 if uref.ndim == 2 else 0)
        self._call("set_lqr_cost", C.c_int(k_begin), C.c_int(k_end), _dp(Qc), _dp(Rc), _dp(xref),
                   _dp(uref), C.c_int(per))

    def set_user_cost(self, k_begin, k_end, params, type=0):
        """A user cost of the handle's user model (register_model_source) on knots [k_begin, k_end);
        params: [nparams] or [B][nparams]; ``type``: index of the cost class in the source's ALTRO_USER_COSTS list."""
        p = _f64(params)
        per = 1 if p.ndim == 2 else 0
        self._call("set_user_cost_type", C.c_int(type), C.c_int(k_begin), C.c_int(k_end), _dp(p), C.c_int(p.shape[-1]), C.c_int(per))

    def add_constraint(self, kind, k_begin, k_end, params):
        p = _f64(params)
        per = 1 if p.ndim == 2 else 0
        self._call("add_constraint", C.c_int(kind), C.c_int(k_begin), C.c_int(k_end), _dp(p),
                   C.c_int(p.shape[-1]), C.c_int(per))

    def add_user_constraint(self, k_begin, k_end, params, type=0):
        """A user constraint of the handle's user model on knots [k_begin, k_end); params: [nparams] or [B][nparams];
        ``type``: index of the constraint class in the source's ALTRO_USER_CONSTRAINTS list."""
        p = _f64(params)
        per = 1 if p.ndim == 2 else 0
        self._call("add_user_constraint_type", C.c_int(type), C.c_int(k_begin), C.c_int(k_end), _dp(p),
                   C.c_int(p.shape[-1]), C.c_int(per))

    def add_goal_constraint(self, k, xf):
        self.add_constraint(CON_GOAL, k, k + 1, xf)

    def add_control_bound(self, k_begin, k_end, lb, ub):
        self.add_constraint(CON_CONTROL_BOUND, k_begin, k_end, np.concatenate([_f64(lb), _f64(ub)]))

    def add_circle_constraint(self, k_begin, k_end, circles):
        """circles: [nobs][3] (cx, cy, r) or [B][nobs][3]."""
        c = _f64(circles)
        c = c.reshape(c.shape[0], -1) if c.ndim == 3 else c.reshape(-1)
        self.add_constraint(CON_CIRCLE, k_begin, k_end, c)

    def set_initial_state(self, x0):
        x0 = _f64(x0)
        self._call("set_initial_state", _dp(x0), C.c_int(1 if x0.ndim == 2 else 0))

    def set_trajectory(self, X=None, U=None):
        X, U = _f64(X), _f64(U)
        per = 1 if (U is not None and U.ndim == 3) or (X is not None and X.ndim == 3) else 0
        self._call("set_trajectory", _dp(X), _dp(U), C.c_int(per))

    def reset_trajectory(self):
        self._call("reset_trajectory")

    def reset_stats(self):
        """solver.GetStats().Reset(): a bare iLQR solve accumulates iterations_total otherwise (quirk Q11)."""
        self._call("reset_stats")

    # -- options (solver.GetOptions()) -------------------------------------------------------------
    def default_options(self):
        o = Options()
        f = self._fn("default_options")
        f.restype = None
        f(C.byref(o))
        return o

    def get_options(self):
        o = Options()
        self._call("get_options", C.byref(o))
        return o

    def set_options(self, opts=None, **kw):
        o = opts if opts is not None else self.get_options()
        for k, v in kw.items():
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, v)
        self._call("set_options", C.byref(o))

    def set_penalty(self, rho):
        self._call("set_penalty", C.c_double(rho))

    def set_penalty_scaling(self, phi):
        self._call("set_penalty_scaling", C.c_double(phi))

    # -- algorithm ------------------------------------------------------------------------------
    def solve(self):            # AugmentedLagrangianiLQR::Solve, al_solver.hpp:304-334
        self._call("solve_al")

    def solve_async(self):
        """Non-blocking AL solve on a worker thread of the library (MPC pattern); finish with wait()."""
        self._call("solve_al_async")

    def poll(self):
        done = C.c_int(0)
        self._call("solve_poll", C.byref(done))
        return bool(done.value)

    def wait(self):
        self._call("wait")

    def solve_ilqr(self):       # iLQR::Solve, ilqr.hpp:284-316
        self._call("solve_ilqr")

    def al_init(self):
        self._call("al_init")

    def solve_setup(self):
        self._call("solve_setup")

    def rollout(self):
        self._call("rollout")

    def cost(self):
        J = np.empty(self.batch)
        self._call("cost", _dp(J))
        return J

    def update_expansions(self):
        self._call("update_expansions")

    def backward_pass(self):
        self._call("backward_pass")

    def forward_pass(self):
        self._call("forward_pass")

    def update_convergence_statistics(self):
        self._call("update_convergence_statistics")

    def update_duals(self):
        self._call("update_duals")

    def update_penalties(self):
        self._call("update_penalties")

    def get_max_violation(self):
        out = np.empty(self.batch)
        self._call("get_max_violation", _dp(out))
        return out

    def max_violation(self):
        out = np.empty(self.batch)
        self._call("max_violation", _dp(out))
        return out

    def get_max_penalty(self):
        out = np.empty(self.batch)
        self._call("get_max_penalty", _dp(out))
        return out

    # -- results ----------------------------------------------------------------------------------
    def get_trajectory(self, X=None, U=None):
        """X [batch][N+1][n], U [batch][N][m]; pass C-contiguous float64 arrays of these shapes to have them filled in
        place (a caller that solves in a loop keeps its buffers: fresh pages cost more than the copy)."""
        if X is None:
            X = np.empty((self.batch, self.N + 1, self.n))
        if U is None:
            U = np.empty((self.batch, self.N, self.m))
        for a_, shape in ((X, (self.batch, self.N + 1, self.n)), (U, (self.batch, self.N, self.m))):
            if a_.dtype != np.float64 or a_.shape != shape or not a_.flags["C_CONTIGUOUS"]:
                raise ValueError(f"get_trajectory needs C-contiguous float64 arrays of shape {shape}")
        self._call("get_trajectory", _dp(X), _dp(U))
        return X, U

    def get_gains(self):
        """K[b][k] as an (m x n) matrix, d[b][k] (m)."""
        K = np.empty((self.batch, self.N, self.n, self.m))  # column-major m x n per knot
        d = np.empty((self.batch, self.N, self.m))
        self._call("get_gains", _dp(K), _dp(d))
        return np.swapaxes(K, 2, 3), d

    def set_record_ctg(self, enable=True):
        self._call("set_record_ctg", C.c_int(1 if enable else 0))

    def get_ctg(self):
        P = np.empty((self.batch, self.N + 1, self.n, self.n))
        p = np.empty((self.batch, self.N + 1, self.n))
        self._call("get_ctg", _dp(P), _dp(p))
        return np.swapaxes(P, 2, 3), p

    def get_expansion(self, k):
        n, m, B = self.n, self.m, self.batch
        AB = np.empty((B, n + m, n))
        lxx = np.empty((B, n, n))
        lxu = np.empty((B, m, n))
        luu = np.empty((B, m, m))
        lx = np.empty((B, n))
        lu = np.empty((B, m))
        self._call("get_expansion", C.c_int(k), _dp(AB), _dp(lxx), _dp(lxu), _dp(luu), _dp(lx), _dp(lu))
        AB = np.swapaxes(AB, 1, 2)
        return dict(A=AB[:, :, :n], B=AB[:, :, n:], lxx=np.swapaxes(lxx, 1, 2),
                    lxu=np.swapaxes(lxu, 1, 2), luu=np.swapaxes(luu, 1, 2), lx=lx, lu=lu)

    def get_knot_costs(self):
        c = np.empty((self.batch, self.N + 1))
        self._call("get_knot_costs", _dp(c))
        return c

    def num_constraints(self, k=None):
        if k is None:
            f = self._fn("num_constraints")
            f.restype = C.c_int
            return f(self._h)
        f = self._fn("num_constraints_at")
        f.restype = C.c_int
        return f(self._h, C.c_int(k))

    def get_duals(self):
        out = np.empty((self.batch, max(self.num_constraints(), 0)))
        if out.size:
            self._call("get_duals", _dp(out))
        return out

    def set_duals(self, lam):
        lam = _f64(lam)
        self._call("set_duals", _dp(lam))

    def get_penalties(self):
        out = np.empty((self.batch, max(self.num_constraints(), 0)))
        if out.size:
            self._call("get_penalties", _dp(out))
        return out

    def get_constraint_values(self):
        out = np.empty((self.batch, max(self.num_constraints(), 0)))
        if out.size:
            self._call("get_constraint_values", _dp(out))
        return out

    def get_stats(self):
        """numpy structured array, one record per instance (altro_stats)."""
        out = np.zeros(self.batch, dtype=STATS_DTYPE)
        self._call("get_stats", out.ctypes.data_as(C.c_void_p))
        return out

    def get_timing(self):
        t = Timing()
        self._call("get_timing", C.byref(t))
        return {k: getattr(t, k) for k, _ in Timing._fields_}

    def set_record_history(self, capacity=301):
        self._call("set_record_history", C.c_int(capacity))

    def get_history(self, instance, field, cap=1024):
        fi = HISTORY_FIELDS.index(field) if isinstance(field, str) else int(field)
        out = np.empty(cap)
        f = self._fn("get_history")
        f.restype = C.c_int
        cnt = f(self._h, C.c_int(instance), C.c_int(fi), _dp(out), C.c_int(cap))
        if cnt < 0:
            raise AltroError("get_history failed: " + self._errmsg(self._h))
        return out[:cnt].copy()

    # -- device interop ---------------------------------------------------------------------------
    def pack_results_device(self, device_ptr):
        self._call("pack_results_device", C.c_void_p(device_ptr))

    def pack_trajectory_device(self, x_ptr, u_ptr):
        """X[B][N+1][n], U[B][N][m] doubles into device memory of this handle's device (either pointer may be 0)."""
        self._call("pack_trajectory_device", C.c_void_p(x_ptr or None), C.c_void_p(u_ptr or None))

    def device_info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int()
        self._call("device_info", name, C.c_int(256), C.byref(cu))
        return name.value.decode(), cu.value
