"""Problem factories: host-side counterparts of the reference's examples/problems/*.

Each factory configures an already-created ``BatchSolver`` (any implementation of the C-ABI) the way
the reference's ``UnicycleProblem::MakeProblem`` / ``TripleIntegratorProblem::MakeProblem`` build an
``altro::problem::Problem`` (examples/problems/unicycle.cpp:11-89, unicycle.hpp:84-92,
examples/problems/triple_integrator.hpp:22-105), and the ``batch_*`` helpers generate the seeded
synthetic batches of SURVEY.md section 8(d) (instance 0 is always the exact reference problem).

Quirk Q1 (float time step) is reproduced: ``h = float32(tf) / N`` and the cost weights inherit it.
"""
import numpy as np

from . import (CON_CIRCLE, CON_CONTROL_BOUND, CON_GOAL, F32, F64, MODEL_QUADROTOR12,
               MODEL_TRIPLE_INTEGRATOR, MODEL_UNICYCLE, BatchSolver)

SEED_BASE = 20260927


def _f32step(tf, N):
    return np.float32(np.float32(tf) / np.float32(N))


# ---------------------------------------------------------------------------------------------------
# Unicycle (examples/problems/unicycle.cpp)
# ---------------------------------------------------------------------------------------------------
def unicycle_turn90(make, batch=1, N=100, dtype=F64, constraints=True, xf=None, u0=None, **kw):
    """kTurn90 scenario (unicycle.cpp:17-25, unicycle.hpp:41-55): goal + control bounds +-1.5.

    ``make(n, m, N, batch, dtype)`` returns a BatchSolver-like object.  ``xf`` may be [3] or [B][3].
    """
    s = make(3, 2, N, batch, dtype)
    h = _f32step(3.0, N)
    hd = float(h)
    Q = np.eye(3) * (1e-2 * hd)
    R = np.eye(2) * (1e-2 * hd)
    Qf = np.eye(3) * 100.0
    xf = np.array([1.5, 1.5, np.pi / 2]) if xf is None else np.asarray(xf, dtype=np.float64)
    u0 = np.array([0.1, 0.1]) if u0 is None else np.asarray(u0, dtype=np.float64)
    uref = np.zeros(2)
    s.set_model(MODEL_UNICYCLE)
    s.set_uniform_step(h)
    s.set_lqr_cost(0, N, Q, R, xf, uref)
    s.set_lqr_cost(N, N + 1, Qf, R * 0, xf, uref)
    if constraints:
        s.add_control_bound(0, N, [-1.5, -1.5], [1.5, 1.5])
        s.add_constraint(CON_GOAL, N, N + 1, xf)
    s.set_initial_state(np.zeros(3))
    U = np.tile(u0, (N, 1)) if u0.ndim == 1 else np.repeat(u0[:, None, :], N, axis=1)
    s.set_trajectory(None, U)
    return s


THREE_OBSTACLE_CIRCLES = np.array([[0.25 * 3.0, 0.25 * 3.0, 0.425],
                                   [0.5 * 3.0, 0.5 * 3.0, 0.425],
                                   [0.75 * 3.0, 0.75 * 3.0, 0.425]])


def unicycle_three_obstacles(make, batch=1, N=100, dtype=F64, constraints=True, circles=None, **kw):
    """kThreeObstacles scenario (unicycle.cpp:27-60): the perf/benchmark_unicycle.cpp problem.

    The circle constraint is added BEFORE the bounds (first in the inequality list), on knots
    1..N-1; bounds v in [0,3], w in [-3,3] on 0..N-1; goal at N."""
    s = make(3, 2, N, batch, dtype)
    h = _f32step(5.0, N)
    hd = float(h)
    Q = np.eye(3) * (1.0 * hd)
    R = np.eye(2) * (0.5 * hd)
    Qf = np.eye(3) * 10.0
    xf = np.array([3.0, 3.0, 0.0])
    uref = np.zeros(2)
    circles = THREE_OBSTACLE_CIRCLES if circles is None else np.asarray(circles, dtype=np.float64)
    s.set_model(MODEL_UNICYCLE)
    s.set_uniform_step(h)
    s.set_lqr_cost(0, N, Q, R, xf, uref)
    s.set_lqr_cost(N, N + 1, Qf, R * 0, xf, uref)
    # The reference registers the obstacles even with add_constraints=false (unicycle.cpp:55-59),
    # but a plain iLQR ignores every constraint (ilqr.hpp:117-119; quirk Q10).  On this ABI "plain
    # iLQR" means "no constraint registered", so constraints=False registers none.
    if constraints:
        s.add_circle_constraint(1, N, circles)
        s.add_control_bound(0, N, [0.0, -3.0], [3.0, 3.0])
        s.add_constraint(CON_GOAL, N, N + 1, xf)
    s.set_initial_state(np.zeros(3))
    s.set_trajectory(None, np.full((N, 2), 0.01))
    return s


# ---------------------------------------------------------------------------------------------------
# Triple integrator (examples/problems/triple_integrator.hpp, test/ilqr/ilqr_test.cpp:29-122)
# ---------------------------------------------------------------------------------------------------
def triple_integrator(make, batch=1, N=10, dtype=F64, constraints=False, goal_only=False,
                      xf=None, h=0.1, **kw):
    """dof=2 triple integrator.  uref = 0 (quirk Q9: the reference factory leaves it uninitialised;
    its unit-test fixture uses zeros).  ``goal_only`` reproduces the ilqr_test.cpp fixture (goal
    constraint at N, no bounds); ``constraints`` reproduces MakeProblem(add_constraints=true)."""
    dof, n, m = 2, 6, 2
    s = make(n, m, N, batch, dtype)
    hf = np.float32(h)
    Q = np.eye(n)
    R = np.eye(m) * 1e-3
    Qf = np.eye(n) * 1e5
    if xf is None:
        xf = np.zeros(n)
        xf[0], xf[1] = 1.0, 2.0
    xf = np.asarray(xf, dtype=np.float64)
    x0 = -xf
    uref = np.zeros(m)
    s.set_model(MODEL_TRIPLE_INTEGRATOR, [dof])
    s.set_uniform_step(hf)
    s.set_lqr_cost(0, N, Q, R, xf, uref)
    s.set_lqr_cost(N, N + 1, Qf, R * 0, xf, uref)
    if constraints:
        ub = np.array([100.0 * (i + 1) for i in range(dof)])
        s.add_control_bound(0, N, -ub, ub)
        s.add_constraint(CON_GOAL, N, N + 1, xf)
    elif goal_only:
        s.add_constraint(CON_GOAL, N, N + 1, xf)
    s.set_initial_state(x0)
    s.set_trajectory(None, np.zeros((N, m)))
    return s


# ---------------------------------------------------------------------------------------------------
# Quadrotor-like 12-state model (BASELINE config 5; build-defined, no reference counterpart)
# ---------------------------------------------------------------------------------------------------
def quadrotor12(make, batch=1, N=200, dtype=F32, xf_pos=None, **kw):
    n, m = 12, 4
    s = make(n, m, N, batch, dtype)
    h = np.float32(0.02)
    hd = float(h)
    Q = np.eye(n) * (1e-2 * hd)
    R = np.eye(m) * (1e-2 * hd)
    Qf = np.eye(n) * 100.0
    if xf_pos is None:
        xf_pos = np.array([1.0, -1.0, 0.5])
    xf_pos = np.asarray(xf_pos, dtype=np.float64)
    xf = np.zeros(xf_pos.shape[:-1] + (n,))
    xf[..., :3] = xf_pos
    uref = np.zeros(m)
    s.set_model(MODEL_QUADROTOR12)
    s.set_uniform_step(h)
    s.set_lqr_cost(0, N, Q, R, xf, uref)
    s.set_lqr_cost(N, N + 1, Qf, R * 0, xf, uref)
    s.add_control_bound(0, N, [-5.0, -3.0, -3.0, -3.0], [5.0, 3.0, 3.0, 3.0])
    s.add_constraint(CON_GOAL, N, N + 1, xf)
    s.set_initial_state(np.zeros(n))
    s.set_trajectory(None, np.zeros((N, m)))
    return s


# ---------------------------------------------------------------------------------------------------
# A user-defined model (tests/models/cartpole.hpp through altro_register_model_source): move the cart
# ---------------------------------------------------------------------------------------------------
def cartpole_move(make, model_kind, batch=1, N=60, dtype=F64, goal=None, **kw):
    """Cart-pole with the pole hanging down: drive the cart to p = goal and come to rest (n = 4, m = 1),
    force bounded by +-3, goal constraint at the last knot.  ``goal`` may be a scalar or [B]."""
    n, m = 4, 1
    s = make(n, m, N, batch, dtype)
    h = np.float32(0.05)
    hd = float(h)
    goal = np.full(batch, 1.0) if goal is None else np.broadcast_to(np.asarray(goal, dtype=np.float64), (batch,))
    xf = np.zeros((batch, n))
    xf[:, 0] = goal
    Q = np.eye(n) * (1e-1 * hd)
    R = np.eye(m) * (1e-2 * hd)
    Qf = np.eye(n) * 100.0
    s.set_model(model_kind)
    s.set_uniform_step(h)
    s.set_lqr_cost(0, N, Q, R, xf, np.zeros(m))
    s.set_lqr_cost(N, N + 1, Qf, R * 0, xf, np.zeros(m))
    s.add_control_bound(0, N, [-3.0], [3.0])
    s.add_constraint(CON_GOAL, N, N + 1, xf)
    s.set_initial_state(np.zeros(n))
    s.set_trajectory(None, np.zeros((N, m)))
    return s


def cartpole_steps(make, model_kind, knot_models, batch=1, N=60, dtype=F64, goal=None, **kw):
    """The cart-pole move with a DIFFERENT discrete dynamics per knot (tests/models/cartpole_steps.hpp: 0 RK4, 1 explicit
    Euler, 2 the user's own discrete map) -- Problem::SetDynamics(model, k), problem.hpp:155-166.  ``knot_models``: [N]
    indices into the source's ALTRO_USER_MODELS list, or None (every knot model 0)."""
    n, m = 4, 1
    s = make(n, m, N, batch, dtype)
    h = np.float32(0.05)
    hd = float(h)
    goal = np.full(batch, 1.0) if goal is None else np.broadcast_to(np.asarray(goal, dtype=np.float64), (batch,))
    xf = np.zeros((batch, n))
    xf[:, 0] = goal
    s.set_model(model_kind)
    if knot_models is not None:
        s.set_knot_models(knot_models)
    s.set_uniform_step(h)
    s.set_lqr_cost(0, N, np.eye(n) * (1e-1 * hd), np.eye(m) * (1e-2 * hd), xf, np.zeros(m))
    s.set_lqr_cost(N, N + 1, np.eye(n) * 100.0, np.zeros((m, m)), xf, np.zeros(m))
    s.add_control_bound(0, N, [-3.0], [3.0])
    s.add_constraint(CON_GOAL, N, N + 1, xf)
    s.set_initial_state(np.zeros(n))
    s.set_trajectory(None, np.zeros((N, m)))
    return s


def pendulum_swing(make, model_kind, batch=1, N=80, dtype=F64, goal=None, **kw):
    """tests/models/pendulum_discrete.hpp (a user model that is only a DiscreteDynamics): bring the damped pendulum from
    rest at theta = 0 to theta = goal and hold it there (n = 2, m = 1), torque bounded by +-6."""
    n, m = 2, 1
    s = make(n, m, N, batch, dtype)
    h = np.float32(0.04)
    hd = float(h)
    goal = np.full(batch, 0.8) if goal is None else np.broadcast_to(np.asarray(goal, dtype=np.float64), (batch,))
    xf = np.zeros((batch, n))
    xf[:, 0] = goal
    uf = (9.81 * np.sin(goal)).reshape(batch, 1)  # the torque that holds the pendulum at the goal
    s.set_model(model_kind)
    s.set_uniform_step(h)
    s.set_lqr_cost(0, N, np.eye(n) * (1.0 * hd), np.eye(m) * (1e-2 * hd), xf, uf)
    s.set_lqr_cost(N, N + 1, np.eye(n) * 100.0, np.zeros((m, m)), xf, uf)
    s.add_control_bound(0, N, [-6.0], [6.0])
    s.add_constraint(CON_GOAL, N, N + 1, xf)
    s.set_initial_state(np.zeros(n))
    s.set_trajectory(None, np.zeros((N, m)))
    return s


def cartpole_track(make, model_kind, batch=1, N=60, dtype=F64, goal=None, sway=0.04, **kw):
    """The cart-pole move with the user's own cost and constraint (tests/models/cartpole_track.hpp): stage and
    terminal cost = UserCost (with the non-quadratic pendulum term), the sway of the pole tip limited to +-sway by
    UserConstraint on every knot after the first, plus the built-in force bound and goal constraint."""
    n, m = 4, 1
    s = make(n, m, N, batch, dtype)
    h = np.float32(0.05)
    hd = float(h)
    goal = np.full(batch, 1.0) if goal is None else np.broadcast_to(np.asarray(goal, dtype=np.float64), (batch,))
    xf = np.zeros((batch, n))
    xf[:, 0] = goal
    stage = np.stack([goal, np.full(batch, 1e-1 * hd), np.full(batch, 2.0 * hd), np.full(batch, 1e-1 * hd),
                      np.full(batch, 1e-1 * hd), np.full(batch, 1e-2 * hd)], axis=1)
    term = np.stack([goal, np.full(batch, 100.0), np.full(batch, 100.0), np.full(batch, 100.0),
                     np.full(batch, 100.0), np.zeros(batch)], axis=1)
    corridor = np.stack([np.full(batch, -sway), np.full(batch, sway)], axis=1)
    s.set_model(model_kind)
    s.set_uniform_step(h)
    s.set_user_cost(0, N, stage)
    s.set_user_cost(N, N + 1, term)
    s.add_control_bound(0, N, [-3.0], [3.0])
    s.add_user_constraint(1, N + 1, corridor)
    s.add_constraint(CON_GOAL, N, N + 1, xf)
    s.set_initial_state(np.zeros(n))
    s.set_trajectory(None, np.zeros((N, m)))
    return s


def cartpole_multi(make, model_kind, batch=1, N=60, dtype=F64, goal=None, sway=0.05, vmax=0.6, **kw):
    """The cart-pole move with SEVERAL user cost / constraint classes in one problem (tests/models/cartpole_multi.hpp):
    stage cost = SwingCost (cost type 0), terminal cost = TipCost (type 1, another parameter count); on the knots after
    the first the sway limit (constraint type 0, 2 inequality rows) AND the cart speed limit (type 2, 1 inequality row)
    share each knot with the built-in force bound; at the last knot the tip-over-goal equality (type 1)."""
    n, m = 4, 1
    s = make(n, m, N, batch, dtype)
    h = np.float32(0.05)
    hd = float(h)
    goal = np.full(batch, 1.0) if goal is None else np.broadcast_to(np.asarray(goal, dtype=np.float64), (batch,))
    stage = np.stack([goal, np.full(batch, 1e-1 * hd), np.full(batch, 2.0 * hd), np.full(batch, 1e-1 * hd),
                      np.full(batch, 1e-1 * hd), np.full(batch, 1e-2 * hd)], axis=1)
    term = np.stack([goal, np.full(batch, 100.0), np.full(batch, 100.0)], axis=1)
    s.set_model(model_kind)
    s.set_uniform_step(h)
    s.set_user_cost(0, N, stage, type=0)
    s.set_user_cost(N, N + 1, term, type=1)
    s.add_control_bound(0, N, [-3.0], [3.0])
    s.add_user_constraint(1, N, np.stack([np.full(batch, -sway), np.full(batch, sway)], axis=1), type=0)
    s.add_user_constraint(1, N, np.array([vmax]), type=2)
    s.add_user_constraint(N, N + 1, goal[:, None].copy(), type=1)
    s.set_initial_state(np.zeros(n))
    s.set_trajectory(None, np.zeros((N, m)))
    return s


# ---------------------------------------------------------------------------------------------------
# Seeded synthetic batches; instance 0 = the exact reference problem.
# The generator is std::mt19937_64 (the C++ facade, include/altro/problems.hpp, draws the same numbers from
# the standard library's engine), uniform doubles as a + (b - a) * (x >> 11) * 2^-53, one instance after the
# other: the Python binding, bench.py and the C++ drivers solve the SAME batches.
# ---------------------------------------------------------------------------------------------------
class Mt19937_64:
    """std::mt19937_64 (Matsumoto & Nishimura's 64-bit Mersenne Twister), bit-exact."""
    NN, MM = 312, 156
    MATRIX_A, UM, LM = 0xB5026F5AA96619E9, 0xFFFFFFFF80000000, 0x7FFFFFFF
    MASK = (1 << 64) - 1

    def __init__(self, seed=5489):
        mt = [0] * self.NN
        mt[0] = seed & self.MASK
        for i in range(1, self.NN):
            mt[i] = (6364136223846793005 * (mt[i - 1] ^ (mt[i - 1] >> 62)) + i) & self.MASK
        self.mt, self.mti = mt, self.NN

    def _twist(self):
        mt, NN, MM = self.mt, self.NN, self.MM
        for i in range(NN):
            x = (mt[i] & self.UM) | (mt[(i + 1) % NN] & self.LM)
            mt[i] = mt[(i + MM) % NN] ^ (x >> 1) ^ (self.MATRIX_A if x & 1 else 0)
        self.mti = 0

    def next(self):
        if self.mti >= self.NN:
            self._twist()
        x = self.mt[self.mti]
        self.mti += 1
        x ^= (x >> 29) & 0x5555555555555555
        x ^= (x << 17) & 0x71D67FFFEDA60000
        x ^= (x << 37) & 0xFFF7EEE000000000
        x ^= x >> 43
        return x & self.MASK

    def uniform(self, a, b):
        return a + (b - a) * ((self.next() >> 11) * (1.0 / 9007199254740992.0))


_batch_cache = {}


def _cached(key, build):
    if key not in _batch_cache:
        _batch_cache[key] = build()
    return _batch_cache[key].copy()


def batch_turn90_goals(batch, seed=SEED_BASE + 3):
    """Per-instance goals of BASELINE config 3: xf = (1.5+dx, 1.5+dy, pi/2+dth), instance 0 exact."""
    def build():
        g = Mt19937_64(seed)
        xf = np.tile(np.array([1.5, 1.5, np.pi / 2]), (batch, 1))
        for b in range(1, batch):
            xf[b, 0] = 1.5 + g.uniform(-0.5, 0.5)
            xf[b, 1] = 1.5 + g.uniform(-0.5, 0.5)
            xf[b, 2] = np.pi / 2 + g.uniform(-0.3, 0.3)
        return xf
    return _cached(("turn90", batch, seed), build)


def batch_obstacle_circles(batch, seed=SEED_BASE + 4):
    """Per-instance obstacles of BASELINE config 4: the three reference circles, centres jittered +-0.1."""
    def build():
        g = Mt19937_64(seed)
        circles = np.tile(THREE_OBSTACLE_CIRCLES, (batch, 1, 1))
        for b in range(1, batch):
            for i in range(3):
                circles[b, i, 0] += g.uniform(-0.1, 0.1)
                circles[b, i, 1] += g.uniform(-0.1, 0.1)
        return circles
    return _cached(("obstacles", batch, seed), build)


def batch_triple_integrator_goals(batch, seed=SEED_BASE + 2):
    """Per-instance goals of BASELINE config 2: xf[0:2] ~ U([0.5, 2]^2), instance 0 = (1, 2)."""
    def build():
        g = Mt19937_64(seed)
        xf = np.zeros((batch, 6))
        xf[:, 0], xf[:, 1] = 1.0, 2.0
        for b in range(1, batch):
            xf[b, 0] = g.uniform(0.5, 2.0)
            xf[b, 1] = g.uniform(0.5, 2.0)
        return xf
    return _cached(("tripleint", batch, seed), build)


def batch_quadrotor12_goals(batch, seed=SEED_BASE + 5):
    """Per-instance hover goals of BASELINE config 5: xf_p ~ U([-2, 2]^3), instance 0 = (1, -1, 0.5)."""
    def build():
        g = Mt19937_64(seed)
        pos = np.tile(np.array([1.0, -1.0, 0.5]), (batch, 1))
        for b in range(1, batch):
            for i in range(3):
                pos[b, i] = g.uniform(-2.0, 2.0)
        return pos
    return _cached(("quad12", batch, seed), build)


def _block(arr, shard):
    """Rows [lo, hi) of a per-instance array of the GLOBAL batch (one rank's shard), or all of it."""
    return arr if shard is None else arr[shard[0]:shard[1]]


def batch_turn90(make, batch, N=100, dtype=F64, seed=SEED_BASE + 3, shard=None):
    """BASELINE config 3: kTurn90 with per-instance goal xf = (1.5+dx, 1.5+dy, pi/2+dth).

    ``batch`` is the size of the (global) seeded batch; ``shard = (lo, hi)`` builds only that block of it."""
    xf = _block(batch_turn90_goals(batch, seed), shard)
    return unicycle_turn90(make, batch=len(xf), N=N, dtype=dtype, xf=xf)


def batch_three_obstacles(make, batch, N=100, dtype=F32, seed=SEED_BASE + 4, shard=None):
    """BASELINE config 4: kThreeObstacles with per-instance obstacle centres jittered +-0.1."""
    circles = _block(batch_obstacle_circles(batch, seed), shard)
    return unicycle_three_obstacles(make, batch=len(circles), N=N, dtype=dtype, circles=circles)


def batch_triple_integrator(make, batch, N=50, dtype=F64, seed=SEED_BASE + 2, shard=None):
    """BASELINE config 2: unconstrained triple integrator, 51 knots, xf[0:2] ~ U([0.5,2]^2)."""
    xf = _block(batch_triple_integrator_goals(batch, seed), shard)
    return triple_integrator(make, batch=len(xf), N=N, dtype=dtype, xf=xf)


def batch_quadrotor12(make, batch, N=200, dtype=F32, seed=SEED_BASE + 5, shard=None):
    """BASELINE config 5: hover-to-hover, xf_p ~ U([-2,2]^3)."""
    pos = _block(batch_quadrotor12_goals(batch, seed), shard)
    return quadrotor12(make, batch=len(pos), N=N, dtype=dtype, xf_pos=pos)


def make_hip(n, m, N, batch, dtype, device_id=0):
    return BatchSolver(n, m, N, batch, dtype, device_id)
