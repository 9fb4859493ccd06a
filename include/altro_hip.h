/*
 * altro_hip.h — C-ABI of the MI355X-native batched AL-iLQR solver (libaltro_hip.so).
 *
 * This header is the drop-in boundary for the hot path of optimusride/altro-cpp
 * (AltroCpp v0.3.4).  The reference has no FFI layer of its own: its public surface is the C++
 * classes altro::problem::Problem, altro::ilqr::iLQR<n,m> and
 * altro::augmented_lagrangian::AugmentedLagrangianiLQR<n,m>, whose extension points are host
 * virtual functions over Eigen::Ref arguments that a GPU kernel cannot call.  Each entry point
 * below therefore replaces one method (or group of methods) of those classes; the reference
 * file:line it replaces is cited next to it.  The header-only C++ facade in include/altro/ keeps
 * the reference's class and method names on top of this ABI.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.  Every function returns an
 *    altro_status; no exception crosses the boundary.  altro_last_error() gives the text.
 *  - All host arrays exchanged over the ABI are IEEE fp64, caller-owned, and are copied during
 *    the call.  The device computes in fp64; the dtype chosen at altro_create selects how the two
 *    bulk per-knot records are STORED in device memory (see altro_dtype).
 *  - NEW relative to the reference: a batch of B independent problem instances per handle.
 *    Host layout is instance-major, then knot point, then Eigen column-major matrix:
 *    X[b][k][i] (k = 0..N), U[b][k][j] (k = 0..N-1), K[b][k][col][row] with K being m x n
 *    (what KnotPointFunctions::GetFeedbackGain() returns per knot,
 *    altro/ilqr/knot_point_function_type.hpp:265).
 *  - "N" is the number of SEGMENTS, as in the reference (altro/ilqr/ilqr.hpp:788); the
 *    trajectory has N+1 knot points.
 *  - The time step is a 32-bit float, as in the reference (altro/common/knotpoint.hpp:179-180,
 *    altro/common/trajectory.hpp:122-130); it is promoted to the compute dtype inside RK4.
 *  - A handle is bound to one device and one HIP stream; calls on one handle must be serialised
 *    by the caller, distinct handles are independent.  solve calls are synchronous.
 *  - The product has NO CPU fallback: every compute entry point fails with ALTRO_HIP_ERROR if no
 *    HIP device is usable.
 */
#ifndef ALTRO_HIP_H_
#define ALTRO_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct altro_solver_s* altro_handle;

typedef enum altro_status {
  ALTRO_OK = 0,
  ALTRO_INVALID_ARG = 1,
  ALTRO_HIP_ERROR = 2,
  ALTRO_NOT_READY = 3,
  ALTRO_UNSUPPORTED = 4
} altro_status;

/* Storage precision on the device.
 *   ALTRO_F64: everything in fp64.
 *   ALTRO_F32: the expansion records ([A|B], lxx, lxu, luu, lx, lu) and the gain records (K, d) -- 3/4 of
 *     the elements a knot point moves per iteration -- are stored in fp32; the trajectory, the multipliers,
 *     every cost and every arithmetic operation (Riccati recursion, rollouts, line search, AL updates) stay
 *     fp64.  Iteration counts and statuses follow the fp64 schedule; states agree with the fp64 solve to
 *     ~1e-6 (stated tolerance 1e-3).  An all-fp32 solver loses 10-15 points of solved fraction on the
 *     obstacle problems (scripts/cpu_fp32_study.py), which is why this build has none. */
typedef enum altro_dtype { ALTRO_F64 = 0, ALTRO_F32 = 1 } altro_dtype;

/* Closed registry of continuous-time models, all discretised with RK4
 * (altro/problem/integration.hpp:123-169 via altro/problem/discretized_model.hpp:36-45). */
typedef enum altro_model_kind {
  ALTRO_MODEL_UNICYCLE = 1,          /* examples/unicycle.cpp:12-33, n=3 m=2 */
  ALTRO_MODEL_TRIPLE_INTEGRATOR = 2, /* examples/triple_integrator.cpp:9-33, n=3*dof m=dof */
  ALTRO_MODEL_QUADROTOR12 = 3,       /* build-defined 12-state/4-control model (BASELINE config 5) */
  ALTRO_MODEL_USER_BASE = 1000       /* kinds >= this: user models, see altro_register_model_source */
} altro_model_kind;

/* Closed registry of constraints (examples/basic_constraints.hpp, obstacle_constraints.hpp). */
typedef enum altro_constraint_kind {
  ALTRO_CON_GOAL = 1,          /* equality   c = x - xf          basic_constraints.hpp:15-40   */
  ALTRO_CON_CONTROL_BOUND = 2, /* inequality c = [lb-u; u-ub]    basic_constraints.hpp:42-151  */
  ALTRO_CON_CIRCLE = 3,        /* inequality c_i = r^2 - |p-c_i|^2  obstacle_constraints.hpp:69-127 */
  ALTRO_CON_USER = 4           /* the UserConstraint of the handle's user model, see altro_register_model_source */
} altro_constraint_kind;

/* altro::SolverStatus, altro/common/solver_stats.hpp:20-31 (same numeric values). */
typedef enum altro_solver_status {
  ALTRO_SOLVED = 0,
  ALTRO_UNSOLVED = 1,
  ALTRO_STATE_LIMIT = 2,
  ALTRO_CONTROL_LIMIT = 3,
  ALTRO_COST_INCREASE = 4,
  ALTRO_MAX_ITERATIONS = 5,
  ALTRO_MAX_OUTER_ITERATIONS = 6,
  ALTRO_MAX_INNER_ITERATIONS = 7,
  ALTRO_MAX_PENALTY = 8,
  ALTRO_BACKWARD_PASS_REGULARIZATION_FAILED = 9
} altro_solver_status;

typedef struct altro_desc {
  int n;         /* state dimension   */
  int m;         /* control dimension */
  int N;         /* number of segments (N+1 knot points) */
  int batch;     /* number of independent problem instances */
  int dtype;     /* altro_dtype */
  int device_id; /* HIP device ordinal */
} altro_desc;

/* altro::SolverOptions, altro/common/solver_options.hpp:19-57, field for field (bools as int).
 * Console/profiler-file fields are host-only in the reference and are not mirrored. */
typedef struct altro_options {
  int max_iterations_total;          /* 300  */
  int max_iterations_outer;          /* 30   */
  int max_iterations_inner;          /* 100  */
  double cost_tolerance;             /* 1e-4 */
  double gradient_tolerance;         /* 1e-2 */
  double bp_reg_increase_factor;     /* 1.6  */
  int bp_reg_enable;                 /* 1 (declared but never read by the reference) */
  double bp_reg_initial;             /* 0.0  */
  double bp_reg_max;                 /* 1e8  */
  double bp_reg_min;                 /* 1e-8 */
  int bp_reg_fail_threshold;         /* 100  */
  int check_forwardpass_bounds;      /* 1    */
  double state_max;                  /* 1e8  */
  double control_max;                /* 1e8  */
  int line_search_max_iterations;    /* 20   */
  double line_search_lower_bound;    /* 1e-8 */
  double line_search_upper_bound;    /* 10.0 */
  double line_search_decrease_factor;/* 2    */
  double constraint_tolerance;       /* 1e-4 */
  double maximum_penalty;            /* 1e8  */
  double initial_penalty;            /* 1.0  */
  int reset_duals;                   /* 1    */
  int profiler_enable;               /* 0: when 1, per-section device timings are recorded */
} altro_options;

/* Per-instance result record: the scalar members of altro::SolverStats
 * (altro/common/solver_stats.hpp:52-63) plus the `.back()` value of each logged vector. */
typedef struct altro_stats {
  int status;           /* AL status after altro_solve_al, iLQR status after altro_solve_ilqr */
  int status_ilqr;      /* status of the inner iLQR solver (ilqr.hpp:164) */
  int iterations_inner; /* of the last inner solve */
  int iterations_outer;
  int iterations_total;
  int reserved;
  double cost;              /* stats.cost.back()            */
  double initial_cost;      /* stats.initial_cost           */
  double cost_decrease;     /* stats.cost_decrease.back()   */
  double gradient;          /* stats.gradient.back()        */
  double violation;         /* stats.violations.back()      */
  double max_penalty;       /* stats.max_penalty.back()     */
  double alpha;             /* stats.alpha.back()           */
  double regularization;    /* stats.regularization.back()  */
  double improvement_ratio; /* stats.improvement_ratio.back() ("z") */
} altro_stats;

/* Device timing of the last solve, section names after the reference's profiler tree
 * (altro/ilqr/ilqr.hpp:294,351,386,513; altro/augmented_lagrangian/al_solver.hpp:289,309). */
typedef struct altro_timing {
  double total_ms;         /* "al" or "ilqr": wall time of the solve call                   */
  double init_ms;          /* "init": AL Init + first rollout/cost                          */
  double expansions_ms;    /* sum over sweeps of the expansions kernel (HIP events)         */
  double backward_pass_ms; /* sum over sweeps of the backward-pass kernel                   */
  double forward_pass_ms;  /* sum over sweeps of the forward-pass (+AL update) kernel       */
  double fused_ms;         /* the persistent tail launch (k_sweep_fused): every remaining    */
                           /* iteration of the straggler instances, one workgroup each      */
  int sweeps;              /* batched iLQR sweeps = longest chain of iterations             */
  int fused_sweeps;        /* how many of them ran inside the persistent launch             */
  int launches;            /* number of kernel launches                                     */
  int sweep_launches;      /* batched sweeps launched (all chains of sweeps together)       */
  long long instance_iterations; /* sum over instances of iterations_total                  */
  long long fused_instance_iterations; /* (instance, iteration) units run by the fused launch */
  int host_naps;           /* times the host thread slept (~50 us) instead of spinning while it    */
                           /* waited for a sweep counter (large batches)                             */
  int twin_workgroups;     /* twin workgroups of the persistent launch: each may take over the second half  */
                           /* of one straggler's rejection streak (0: none launched, ALTRO_HIP_TWIN=0)        */
  int twin_claims;         /* ... twins that found a streak and claimed its second half                      */
  int twin_handovers;      /* ... claims the primary confirmed: instances finished by their twin              */
  int fused_workgroup_iterations; /* most iterations ONE workgroup of the persistent launch ran (a twin or its    */
                           /* primary: their share of the instance's iterations)                               */
  int segment_columns;     /* shadow columns the batched sweeps used for segments of rejection streaks (a streak   */
                           /* split in four: three columns; 0: none split, ALTRO_HIP_SEGMENTS=0)                   */
  /* the device-side sweep loop (k_sweep_loop, round 6): ONE launch of persistent workgroups runs the bulk phase --   */
  /* expansions, backward pass, forward pass of every instance, iteration after iteration -- without the host         */
  double loop_ms;          /* duration of that launch (0: the host-paced sweeps ran, ALTRO_HIP_SWEEP_LOOP=0)        */
  int loop_workgroups;     /* persistent workgroups that ran at least one iteration                                */
  int loop_iterations;     /* most iterations one of them ran                                                      */
  int loop_handover;       /* instances it handed to the persistent tail kernel                                    */
  long long loop_instance_iterations; /* (instance, iteration) units it ran                                        */
} altro_timing;

/* ---- lifetime -------------------------------------------------------------------------------- */

/* Replaces AugmentedLagrangianiLQR<n,m>(int N) / iLQR<n,m>(int N)
 * (al_solver.hpp:35, ilqr.hpp:50-55) with a batch dimension. */
altro_status altro_create(const altro_desc* desc, altro_handle* out);
void altro_destroy(altro_handle h);
/* The descriptor the handle was created with (iLQR::NumSegments / StateDimension / ControlDimension,
 * altro/ilqr/ilqr.hpp:140-166, plus batch, dtype and device): callers that size device buffers for
 * altro_pack_results_device / altro_pack_trajectory_device read the dimensions from the handle instead of trusting
 * their own copy (libaltro_group.so does). */
altro_status altro_get_desc(altro_handle h, altro_desc* out);
/* Text of the last error on this handle (or of the last failed altro_create when h == NULL). */
const char* altro_last_error(altro_handle h);
/* Fill *opts with the reference defaults (solver_options.hpp:23-56). */
void altro_default_options(altro_options* opts);

/* ---- problem definition (replaces altro::problem::Problem setters, problem.hpp:113-202) ------- */

/* Problem::SetDynamics for all k with DiscretizedModel<Model, RungeKutta4> (problem.hpp:155-166).
 * params: TRIPLE_INTEGRATOR -> {dof}; UNICYCLE -> none; QUADROTOR12 -> none. */
altro_status altro_set_model(altro_handle h, int kind, const double* params, int nparams);

/* USER-DEFINED DYNAMICS -- the reference's plug-in surface for models (problem::ContinuousDynamics,
 * altro/problem/dynamics.hpp:59-95: Evaluate + Jacobian, wrapped in DiscretizedModel<Model, RungeKutta4>,
 * problem/discretized_model.hpp:24-65).  A kernel cannot call the caller's virtual functions, so the model crosses
 * this boundary as SOURCE: a translation-unit fragment that defines
 *     struct UserModel {
 *       static constexpr int n = ..., m = ...;
 *       template <class T> ALTRO_MODEL_FN static void f(const T* x, const T* u, T* xdot);
 *       template <class T> ALTRO_MODEL_FN static void jac(const T* x, const T* u, T* J);   // n x (n+m), column-major
 *     };
 * The library compiles it with hipcc for the device's architecture into a plugin that carries every kernel of the
 * solver instantiated for these dynamics (cached on disk by content hash: first registration ~1 minute, later ones
 * milliseconds; ALTRO_HIP_CACHE_DIR, default <library dir>/_user_cache), and returns the model kind to pass to
 * altro_set_model.  check_jacobian != 0 runs the device-side FunctionBase::CheckJacobian
 * (altro/common/functionbase.cpp:35-73: forward differences, 64 random points, tolerance 1e-4) once -- now if a
 * device is present, else when the first handle using the model is created; a mismatch is ALTRO_INVALID_ARG.
 * Errors (compiler output included) are reported through altro_last_error(NULL).  No CPU fallback.
 *
 * USER-DEFINED COST AND CONSTRAINT -- the other two plug-in classes of the reference (problem::CostFunction,
 * altro/problem/costfunction.hpp:52-73: Evaluate / Gradient / Hessian; constraints::Constraint<ConType>,
 * altro/constraints/constraint.hpp:173-202: OutputDimension / Evaluate / Jacobian) travel in the same source, both
 * optional and announced by a macro:
 *     struct UserCost {
 *       static constexpr int nparams = ...;      // doubles handed to every call (altro_set_user_cost)
 *       template <class T> ALTRO_MODEL_FN static T    eval(const T* x, const T* u, const T* par);
 *       template <class T> ALTRO_MODEL_FN static void gradient(const T* x, const T* u, const T* par, T* dx, T* du);
 *       template <class T> ALTRO_MODEL_FN static void hessian(const T* x, const T* u, const T* par,
 *                                                          T* dxdx, T* dxdu, T* dudu);  // n x n, n x m, m x m, column-major
 *     };
 *     #define ALTRO_USER_COST UserCost
 *     struct UserConstraint {
 *       static constexpr int p = ..., nparams = ...;   // OutputDimension; doubles handed to every call
 *       static constexpr bool equality = false;        // ConType: constraints::Equality or NegativeOrthant (c <= 0)
 *       template <class T> ALTRO_MODEL_FN static void eval(const T* x, const T* u, const T* par, T* c);
 *       template <class T> ALTRO_MODEL_FN static void jacobian(const T* x, const T* u, const T* par, T* J);  // p x (n+m), column-major
 *     };
 *     #define ALTRO_USER_CONSTRAINT UserConstraint
 * With check_jacobian != 0 the gradient, the Hessian and the constraint Jacobian are checked against finite
 * differences on the device as well (FunctionBase::CheckGradient / CheckHessian / CheckJacobian,
 * functionbase.cpp:42-125).  At the terminal knot u is the zero vector, as in the reference.
 *
 * OTHER DISCRETISATIONS AND PER-KNOT MODELS -- what else Problem::SetDynamics accepts in the reference:
 *   - DiscretizedModel<Model, ExplicitEuler> (altro/problem/integration.hpp:87-104): the model struct declares
 *         static constexpr int integrator = 1;       // 0 / absent: RungeKutta4
 *     (or the source starts with `#define ALTRO_USER_INTEGRATOR 1`, the default for every model of the source):
 *     x+ = x + f(x, u, t) h, [A | B] = Identity(n, n + m) + jac h -- on every kernel of the solver, like RK4.
 *   - the caller's own problem::DiscreteDynamics subclass (altro/problem/dynamics.hpp:148-187: Evaluate(x, u, t, h,
 *     xnext), Jacobian(x, u, t, h, jac)): the struct declares `static constexpr bool discrete = true` and defines,
 *     instead of f / jac,
 *         template <class T> ALTRO_MODEL_FN static void step(const T* x, const T* u, float t, float h, T* xnext);
 *         template <class T> ALTRO_MODEL_FN static void step_jac(const T* x, const T* u, float t, float h, T* J);  // n x (n+m)
 *     with the knot's 32-bit float time and step (knotpoint.hpp:179-180); step_jac is checked against finite differences
 *     of step.  Such a model runs on the general kernels, like a time-varying one.
 *   - a DIFFERENT model on every knot (problem.hpp:155-166, 187-191), all of one (n, m): the source defines several
 *     structs (each continuous / Euler / discrete, time-varying or not) and lists them,
 *         #define ALTRO_USER_MODELS ModelA, ModelB
 *     (no `struct UserModel` needed then), and altro_set_knot_models assigns each knot its index in the list. */
altro_status altro_register_model_source(const char* name, const char* source, int check_jacobian, int* kind_out);

/* Problem::SetDynamics(model, k) with models that differ along the horizon (problem.hpp:155-166; the vector overload
 * :187-191): model_of_knot[k], k = 0 .. N-1 (`count` must be N), is the index of knot k's model in the ALTRO_USER_MODELS
 * list of the handle's user-model source.  Part of the problem definition (before the first compute call); without it
 * every knot uses model 0.  A handle whose model is not a list accepts only zeros (checked at the first compute call). */
altro_status altro_set_knot_models(altro_handle h, const int* model_of_knot, int count);

/* Path of the compiled plugin of a registered user model (the on-disk cache entry: <cache>/altro_user_<hash>.so, next
 * to its generated .hip and the compiler's .log).  Returns the length of the path, -1 for an unknown kind.  A deployment
 * prunes its cache directory with it: everything that is not the path of a model it registers is stale. */
int altro_user_model_path(int kind, char* buf, int len);

/* Trajectory::SetUniformStep (trajectory.hpp:122-130).  hstep > 0.  Belongs to the trajectory, not to the
 * problem definition: may be called at any time (also between solves).  Calls that integrate (rollout,
 * expansions, forward pass, solves) return ALTRO_NOT_READY until a step has been set. */
altro_status altro_set_uniform_step(altro_handle h, float hstep);

/* Trajectory::SetStep(k, h) for k = 0 .. N-1 and Trajectory::SetTime(k, t) for k = 0 .. N (trajectory.hpp:119-120;
 * per-knot float h, t in knotpoint.hpp:179-180): `count` must be N resp. N + 1.  Like SetUniformStep they belong to the
 * trajectory and may change between solves; altro_set_uniform_step afterwards overwrites both again.  Steps without
 * times leave the times as they were (zero if never set).  The times only matter to a time-varying user model
 * (`static constexpr bool time_varying = true`: f(x, u, t, xdot), jac(x, u, t, J) with a 32-bit float t -- the
 * counterpart of ContinuousDynamics::Evaluate(x, u, t, xdot), dynamics.hpp:59-95; RungeKutta4's stage times
 * t, t + h/2, t + h/2, t + h and its Jacobian times t, t/2, t/2, t are the reference's, integration.hpp:123-150).
 * A trajectory with per-knot steps (or a time-varying model) runs on the solver's general kernels: same results and
 * schedule as the reference, without the fused persistent path.
 * altro_get_steps copies the steps [N] and times [N + 1] back (either pointer may be NULL). */
altro_status altro_set_steps(altro_handle h, const float* hk, int count);
altro_status altro_set_times(altro_handle h, const float* tk, int count);
altro_status altro_get_steps(altro_handle h, float* hk, float* tk);

/* Problem::SetCostFunction(QuadraticCost::LQRCost(Q,R,xref,uref,terminal), k) for
 * k_begin <= k < k_end (problem.hpp:113-127, examples/quadratic_cost.hpp:29-39).
 * Q is n x n, R is m x m (column-major).  per_instance bit 0: xref is [B][n] instead of [n];
 * bit 1: uref is [B][m] instead of [m]. */
altro_status altro_set_lqr_cost(altro_handle h, int k_begin, int k_end, const double* Q,
                                const double* R, const double* xref, const double* uref,
                                int per_instance);

/* Problem::SetCostFunction(std::make_shared<UserCost>(params), k) for k_begin <= k < k_end (problem.hpp:113-127)
 * with the UserCost of the handle's user model (altro_register_model_source).  params: UserCost::nparams doubles,
 * [B][nparams] when per_instance != 0.  The last cost set on a knot wins, whichever kind. */
altro_status altro_set_user_cost(altro_handle h, int k_begin, int k_end, const double* params, int nparams,
                                 int per_instance);
/* Several cost / constraint CLASSES in one model source -- every knot of the reference's Problem may carry any
 * CostFunction / Constraint subclass (problem.hpp:113-202): the source lists them,
 *     #define ALTRO_USER_COSTS        TrackCost, ParkCost           (instead of ALTRO_USER_COST)
 *     #define ALTRO_USER_CONSTRAINTS  SwayLimit, RestAtGoal         (instead of ALTRO_USER_CONSTRAINT)
 * each a struct of the form described above with its own nparams / p / equality, and `type` is the index in the list
 * (altro_set_user_cost and altro_add_constraint(ALTRO_CON_USER) mean type 0).  The derivative checks at registration
 * cover every type. */
altro_status altro_set_user_cost_type(altro_handle h, int type, int k_begin, int k_end, const double* params, int nparams,
                                      int per_instance);
altro_status altro_add_user_constraint_type(altro_handle h, int type, int k_begin, int k_end, const double* params,
                                            int nparams, int per_instance);

/* Problem::SetConstraint(con, k) for k_begin <= k < k_end (problem.hpp:178-202).  Insertion order
 * is kept: at each knot the AL cost visits all equalities, then all inequalities, each in
 * insertion order (al_cost.hpp:267-272).
 *   GOAL:          params = xf[n]                         (per_instance: [B][n])
 *   CONTROL_BOUND: params = lb[m], ub[m]  (+-inf allowed; only finite bounds produce rows)
 *   CIRCLE:        params = (cx, cy, r) x nobs            (per_instance: [B][3*nobs])
 * nparams is the length of ONE instance's block. */
altro_status altro_add_constraint(altro_handle h, int kind, int k_begin, int k_end,
                                  const double* params, int nparams, int per_instance);

/* Problem::SetInitialState (problem.hpp:100-106).  per_instance: x0 is [B][n] instead of [n]. */
altro_status altro_set_initial_state(altro_handle h, const double* x0, int per_instance);

/* iLQR::SetTrajectory (ilqr.hpp:231-235): initial guess.  X may be NULL (zeros); U is [N][m] or,
 * with per_instance, [B][N][m] (X: [N+1][n] / [B][N+1][n]). */
altro_status altro_set_trajectory(altro_handle h, const double* X, const double* U,
                                  int per_instance);

/* Re-install, device-side, the trajectory last given to altro_set_trajectory (what
 * `*traj_ptr = prob_def.InitialTrajectory()` does between solves in perf/benchmark_unicycle.cpp:66)
 * without a host round trip. */
altro_status altro_reset_trajectory(altro_handle h);

/* solver.GetStats().Reset() (solver_stats.cpp:31-45): clears the iteration counters and the logged rows.
 * AugmentedLagrangianiLQR::Solve does this itself (al_solver.hpp:298); a bare iLQR::Solve does NOT, so
 * iterations_total -- and with it the max_iterations_total cap -- accumulates over repeated altro_solve_ilqr
 * calls exactly as in the reference (ilqr.hpp:284-316) unless the caller resets.  Asynchronous on the
 * handle's stream, like altro_reset_trajectory. */
altro_status altro_reset_stats(altro_handle h);

/* solver.GetOptions() (al_solver.hpp:44, ilqr.hpp:161). */
altro_status altro_set_options(altro_handle h, const altro_options* opts);
altro_status altro_get_options(altro_handle h, altro_options* opts);

/* AugmentedLagrangianiLQR::SetPenalty / SetPenaltyScaling (al_solver.hpp:271-285). */
altro_status altro_set_penalty(altro_handle h, double rho);
altro_status altro_set_penalty_scaling(altro_handle h, double phi);

/* ---- solves ---------------------------------------------------------------------------------- */

/* AugmentedLagrangianiLQR::Solve (al_solver.hpp:304-334) for every instance. */
altro_status altro_solve_al(altro_handle h);
/* iLQR::Solve (ilqr.hpp:284-316) on the AL cost with the current duals/penalties. */
altro_status altro_solve_ilqr(altro_handle h);

/* Non-blocking variant of altro_solve_al for the MPC pattern the reference is designed for
 * (docs/Overview.dox:48-54; warm start: al_solver.hpp:292-297): the solve runs on a worker thread of
 * the library (one per handle, parked between solves) while the caller prepares the next problem.
 * Between altro_solve_al_async and altro_wait only altro_solve_poll, altro_wait and altro_last_error
 * answer; every other call on the handle returns ALTRO_NOT_READY without touching the solve in flight.
 * altro_solve_poll sets *done to 0/1 without blocking; altro_wait blocks and returns the status the
 * synchronous call would have returned (ALTRO_NOT_READY if no asynchronous solve is pending). */
altro_status altro_solve_al_async(altro_handle h);
altro_status altro_solve_poll(altro_handle h, int* done);
altro_status altro_wait(altro_handle h);

/* ---- step-level entry points (used by the parity tests the way the reference tests use the
 *      public methods of iLQR / AugmentedLagrangianiLQR) ---------------------------------------- */
altro_status altro_al_init(altro_handle h);            /* AL Init           al_solver.hpp:287-302 */
altro_status altro_solve_setup(altro_handle h);        /* iLQR::SolveSetup  ilqr.hpp:629-645      */
altro_status altro_rollout(altro_handle h);            /* iLQR::Rollout     ilqr.hpp:453-459      */
altro_status altro_cost(altro_handle h, double* J);    /* iLQR::Cost        ilqr.hpp:326-334; J[B] may be NULL */
altro_status altro_update_expansions(altro_handle h);  /* iLQR::UpdateExpansions ilqr.hpp:350-366 */
altro_status altro_backward_pass(altro_handle h);      /* iLQR::BackwardPass     ilqr.hpp:385-445 */
altro_status altro_forward_pass(altro_handle h);       /* iLQR::ForwardPass      ilqr.hpp:512-558 */
altro_status altro_update_convergence_statistics(altro_handle h); /* ilqr.hpp:568-587 */
altro_status altro_update_duals(altro_handle h);       /* al_solver.hpp:336-345 */
altro_status altro_update_penalties(altro_handle h);   /* al_solver.hpp:347-355 */
/* GetMaxViolation (stored c_, al_solver.hpp:417-424) / MaxViolation (re-evaluates the cost first,
 * al_solver.hpp:403-408) / GetMaxPenalty (al_solver.hpp:426-434); out[B]. */
altro_status altro_get_max_violation(altro_handle h, double* out);
altro_status altro_max_violation(altro_handle h, double* out);
altro_status altro_get_max_penalty(altro_handle h, double* out);

/* ---- results --------------------------------------------------------------------------------- */
altro_status altro_get_trajectory(altro_handle h, double* X, double* U); /* ilqr.hpp:140 */
/* GetFeedbackGain / GetFeedforwardGain (knot_point_function_type.hpp:265-268): K[B][N][n][m]
 * column-major m x n per knot, d[B][N][m]. */
altro_status altro_get_gains(altro_handle h, double* K, double* d);
/* GetCostToGoHessian / Gradient (knot_point_function_type.hpp:254-255): P[B][N+1][n*n],
 * p[B][N+1][n].  Only recorded for every knot when altro_set_record_ctg(h, 1) was called. */
altro_status altro_set_record_ctg(altro_handle h, int enable);
altro_status altro_get_ctg(altro_handle h, double* P, double* p);
/* GetDynamicsExpansion / GetCostExpansion of knot k (knot_point_function_type.hpp:249-252):
 * AB[B][n*(n+m)] column-major n x (n+m); lxx[B][n*n], lxu[B][n*m], luu[B][m*m], lx[B][n],
 * lu[B][m].  Any pointer may be NULL. */
altro_status altro_get_expansion(altro_handle h, int k, double* AB, double* lxx, double* lxu,
                                 double* luu, double* lx, double* lu);
/* costs_ (ilqr.hpp:163): costs[B][N+1]. */
altro_status altro_get_knot_costs(altro_handle h, double* costs);
/* Total number of constraint rows of one instance (al_solver.hpp:262-269) and at knot k. */
int altro_num_constraints(altro_handle h);
int altro_num_constraints_at(altro_handle h, int k);
/* Duals / penalties / stored constraint values, [B][rows]; rows ordered by knot, then equalities,
 * then inequalities, each in insertion order (constraint_values.hpp:58-60). */
altro_status altro_get_duals(altro_handle h, double* lambda);
altro_status altro_set_duals(altro_handle h, const double* lambda);
altro_status altro_get_penalties(altro_handle h, double* rho);
altro_status altro_get_constraint_values(altro_handle h, double* c);
altro_status altro_get_stats(altro_handle h, altro_stats* stats /*[B]*/);
altro_status altro_get_timing(altro_handle h, altro_timing* t);

/* Per-iteration history of one instance (SolverStats vectors, solver_stats.hpp:56-63).  Recording
 * is off by default; altro_set_record_history(h, capacity) allocates capacity rows per instance.
 * field: 0 cost, 1 alpha, 2 improvement_ratio, 3 gradient, 4 cost_decrease, 5 regularization,
 * 6 violations, 7 max_penalty.  Returns the number of rows written to out (<= cap). */
altro_status altro_set_record_history(altro_handle h, int capacity);
int altro_get_history(altro_handle h, int instance, int field, double* out, int cap);
/* All eight fields of one instance in one call: out[field][cap] (row stride cap), returns the rows per field.  One
 * device synchronisation instead of one per field -- what a SolverStats mirror refreshes after every compute call. */
int altro_get_history_all(altro_handle h, int instance, double* out, int cap);

/* ---- device interop (multi-GPU gather without a host round trip) ------------------------------ */
/* Pack {cost, violation, iterations_total, status} as 4 fp64 per instance into caller-provided
 * DEVICE memory dst[B][4] on this handle's device (e.g. a torch tensor's data_ptr) so it can be
 * fed to an RCCL all_gather. */
altro_status altro_pack_results_device(altro_handle h, void* dst_device);
/* The trajectories in caller-provided DEVICE memory on this handle's device, in the layout of altro_get_trajectory:
 * X_device[B][N+1][n], U_device[B][N][m] doubles (either may be NULL) -- the payload of the optional second collective of
 * SURVEY.md section 8(e), an all-gather of whole trajectories when the caller wants them on one device. */
altro_status altro_pack_trajectory_device(altro_handle h, void* X_device, void* U_device);
/* Device name and multiprocessor count of the handle's device. */
altro_status altro_device_info(altro_handle h, char* name, int name_len, int* cu_count);

#ifdef __cplusplus
}
#endif
#endif /* ALTRO_HIP_H_ */
