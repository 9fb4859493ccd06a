// oracle/altro_oracle.cpp
//
// TEST INFRASTRUCTURE ONLY.  CPU restatement of the AL-iLQR hot path of optimusride/altro-cpp
// (AltroCpp v0.3.4).  It is the parity checker for the HIP product and the timed "port" CPU
// baseline of bench.py.  Nothing under altro-cpp_amd/ or include/altro/ may include, link or call
// this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// The reference itself cannot be compiled in this image (it hard-requires Eigen >= 3.3 and
// fmt 6.1.2, neither of which is installed and neither of which is vendored:
// /root/reference/CMakeLists.txt:50-57), so this file restates the algorithm in plain C++17 with
// explicit loops, following the reference file:line cited at each function.  PARITY IS PINNED:
// tests/test_oracle_reference_constants.py checks this restatement against every known-answer
// constant the reference's own tests hold for the path (the
// constants are data copied into tests/golden/reference_constants.json with their test file:line).
//
// Arithmetic: IEEE fp64 (or fp32 when the handle is created with ALTRO_F32) except the time step
// h, which is a 32-bit float exactly as in the reference (altro/common/knotpoint.hpp:179-180).
// All reference quirks Q1-Q12 of SURVEY.md section 8(a) are reproduced on purpose.
//
// The C API below mirrors include/altro_hip.h one-to-one with the prefix `oracle_` so that the
// parity tests drive both implementations through the same Python code.

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include <pthread.h>
#include <sched.h>
#include <time.h>

#include "../include/altro_hip.h"  // POD structs and enums only

namespace {

// Oracle-only dtype code (next to ALTRO_F64 = 0 / ALTRO_F32 = 1 of include/altro_hip.h): fp64 arithmetic with
// the expansion and gain records rounded to fp32 -- what the product's ALTRO_F32 engine computes.
constexpr int ORACLE_F64_F32REC = 2;
// STUDY modes (scripts/study_fp32_trials.py; nothing in the product computes this way): as
// ORACLE_F64_F32REC, but the line search evaluates its TRIALS in fp32 -- rollout and cost on an all-fp32 shadow of the
// instance -- and only an accepted trial is rolled out and costed again in fp64.
//   3: the fp64 re-evaluation replaces the trial's numbers and the step is taken whatever it says
//   4: the fp64 re-evaluation must pass the acceptance test again, else the search goes on with the next step length
constexpr int ORACLE_F32_TRIALS = 3, ORACLE_F32_TRIALS_RECHECK = 4;

// ------------------------------------------------------------------------------------------------
// Continuous-time models.  jac is n x (n+m), column-major, fully written.
// ------------------------------------------------------------------------------------------------
template <class T>
struct UnicycleModel {  // examples/unicycle.cpp:12-33
  static constexpr int n = 3, m = 2;
  int dof = 0;
  void f(const T* x, const T* u, T* xd) const {
    T theta = x[2], v = u[0], omega = u[1];
    xd[0] = v * std::cos(theta);
    xd[1] = v * std::sin(theta);
    xd[2] = omega;
  }
  void jac(const T* x, const T* u, T* J) const {
    for (int i = 0; i < n * (n + m); ++i) J[i] = T(0);
    T theta = x[2], v = u[0];
    J[0 + 2 * n] = -v * std::sin(theta);
    J[0 + 3 * n] = std::cos(theta);
    J[1 + 2 * n] = v * std::cos(theta);
    J[1 + 3 * n] = std::sin(theta);
    J[2 + 4 * n] = T(1);
  }
};

template <class T, int DOF>
struct TripleIntegratorModel {  // examples/triple_integrator.cpp:9-33
  static constexpr int n = 3 * DOF, m = DOF;
  void f(const T* x, const T* u, T* xd) const {
    for (int i = 0; i < DOF; ++i) {
      xd[i] = x[i + DOF];
      xd[i + DOF] = x[i + 2 * DOF];
      xd[i + 2 * DOF] = u[i];
    }
  }
  void jac(const T*, const T*, T* J) const {
    for (int i = 0; i < n * (n + m); ++i) J[i] = T(0);
    for (int i = 0; i < DOF; ++i) {
      J[i + (i + DOF) * n] = T(1);
      J[(i + DOF) + (i + 2 * DOF) * n] = T(1);
      J[(i + 2 * DOF) + (i + 3 * DOF) * n] = T(1);
    }
  }
};

// Build-defined 12-state / 4-control "quadrotor-like" model for BASELINE config 5 (the reference
// has no such model; parity for it is oracle-vs-GPU only).  State x = (p[3], phi[3], v[3], w[3]),
// control u = (a, tau[3]): small-angle attitude, gravity linearised about hover, with the
// bilinear attitude x thrust coupling kept:
//   p' = v,  phi' = w,  v' = ((g + a) * phi_y, -(g + a) * phi_x, a),  w' = tau.
template <class T>
struct Quadrotor12Model {
  static constexpr int n = 12, m = 4;
  static constexpr double g = 9.81;
  void f(const T* x, const T* u, T* xd) const {
    T a = u[0];
    for (int i = 0; i < 3; ++i) {
      xd[i] = x[6 + i];
      xd[3 + i] = x[9 + i];
      xd[9 + i] = u[1 + i];
    }
    xd[6] = (T(g) + a) * x[4];
    xd[7] = -(T(g) + a) * x[3];
    xd[8] = a;
  }
  void jac(const T* x, const T* u, T* J) const {
    for (int i = 0; i < n * (n + m); ++i) J[i] = T(0);
    T a = u[0];
    for (int i = 0; i < 3; ++i) {
      J[i + (6 + i) * n] = T(1);
      J[(3 + i) + (9 + i) * n] = T(1);
      J[(9 + i) + (n + 1 + i) * n] = T(1);
    }
    J[6 + 4 * n] = T(g) + a;
    J[7 + 3 * n] = -(T(g) + a);
    J[6 + n * n] = x[4];
    J[7 + n * n] = -x[3];
    J[8 + n * n] = T(1);
  }
};

#ifdef ORACLE_USER_MODEL
// A user model (the source text altro_register_model_source() takes, e.g. tests/models/cartpole.hpp) compiled
// for the host: the oracle build with -DORACLE_USER_MODEL='"file"' answers model kinds >= ALTRO_MODEL_USER_BASE
// with it, so that a user model has a CPU reference (tests/test_user_model_gpu.py).
#define ALTRO_MODEL_FN inline
namespace altro_user {
using std::cos;
using std::sin;
#include ORACLE_USER_MODEL
}  // namespace altro_user
template <class T>
struct UserModelAdapter;  // (defined below, behind Pick<>)
#endif
template <class M>
struct IsUserModel : std::false_type {};
#ifdef ORACLE_USER_MODEL
template <class T>
struct IsUserModel<UserModelAdapter<T>> : std::true_type {};
#endif
// The user's cost / constraint classes of the same source (problem::CostFunction, costfunction.hpp:52-73;
// constraints::Constraint<ConType>, constraint.hpp:173-202).  A reference problem may hold any mix of subclasses
// (problem.hpp:66-133): the source lists its types (ALTRO_USER_COSTS A, B / ALTRO_USER_CONSTRAINTS C, D, or a single
// ALTRO_USER_COST / ALTRO_USER_CONSTRAINT) and a cost / constraint carries the index of its type, which is what a
// virtual call resolves in the reference.  Pick<List>::at(i, fn) calls fn with a null pointer of the i-th type.
template <class... Fs>
struct FunctorTypes {
  static constexpr int count = (int)sizeof...(Fs);
};
template <class L>
struct Pick;
template <>
struct Pick<FunctorTypes<>> {
  template <class Fn>
  static bool at(int, Fn&&) { return false; }
};
template <class F, class... Rest>
struct Pick<FunctorTypes<F, Rest...>> {
  template <class Fn>
  static bool at(int i, Fn&& fn) {
    if (i != 0) return Pick<FunctorTypes<Rest...>>::at(i - 1, fn);
    fn(static_cast<F*>(nullptr));
    return true;
  }
};
#ifdef ORACLE_USER_MODEL
namespace user_types_ {
using namespace ::altro_user;
#if defined(ALTRO_USER_COSTS)
using Costs = FunctorTypes<ALTRO_USER_COSTS>;
#elif defined(ALTRO_USER_COST)
using Costs = FunctorTypes<ALTRO_USER_COST>;
#else
using Costs = FunctorTypes<>;
#endif
#if defined(ALTRO_USER_CONSTRAINTS)
using Cons = FunctorTypes<ALTRO_USER_CONSTRAINTS>;
#elif defined(ALTRO_USER_CONSTRAINT)
using Cons = FunctorTypes<ALTRO_USER_CONSTRAINT>;
#else
using Cons = FunctorTypes<>;
#endif
}  // namespace user_types_
using UserCosts = user_types_::Costs;
using UserCons = user_types_::Cons;
#else
using UserCosts = FunctorTypes<>;
using UserCons = FunctorTypes<>;
#endif
// parameter count of cost type t; (parameters, rows, cone) of constraint type t; false for a type the source lacks
inline bool UserCostShape(int t, int* nparams) {
  return Pick<UserCosts>::at(t, [&](auto* f) { *nparams = std::remove_pointer_t<decltype(f)>::nparams; });
}
inline bool UserConShape(int t, int* nparams, int* p, bool* equality) {
  return Pick<UserCons>::at(t, [&](auto* f) {
    using F = std::remove_pointer_t<decltype(f)>;
    *nparams = F::nparams;
    *p = F::p;
    *equality = F::equality;
  });
}

#ifdef ORACLE_USER_MODEL
// What a user source may put behind Problem::SetDynamics(model, k) (problem.hpp:155-166), as include/altro_hip.h lists it:
// a continuous model f / jac under RungeKutta4 (default) or ExplicitEuler (`integrator = 1`, integration.hpp:87-104), a
// DiscreteDynamics of its own (`discrete = true`: step / step_jac, dynamics.hpp:148-187), time-varying or not, and
// SEVERAL of them, one per knot (#define ALTRO_USER_MODELS A, B; `which` = the knot's index into the list).
#ifndef ALTRO_USER_INTEGRATOR
#define ALTRO_USER_INTEGRATOR 0
#endif
template <class U, class = void>
struct UserTimeVarying : std::false_type {};
template <class U>
struct UserTimeVarying<U, std::void_t<decltype(U::time_varying)>> : std::integral_constant<bool, U::time_varying> {};
template <class U, class = void>
struct UserDiscrete : std::false_type {};
template <class U>
struct UserDiscrete<U, std::void_t<decltype(U::discrete)>> : std::integral_constant<bool, U::discrete> {};
template <class U, class = void>
struct UserIntegrator : std::integral_constant<int, ALTRO_USER_INTEGRATOR> {};
template <class U>
struct UserIntegrator<U, std::void_t<decltype(U::integrator)>> : std::integral_constant<int, U::integrator> {};
namespace user_models_ {
using namespace ::altro_user;
template <class M0, class...>
struct Head {
  using type = M0;
};
#if defined(ALTRO_USER_MODELS)
using Models = FunctorTypes<ALTRO_USER_MODELS>;
using First = Head<ALTRO_USER_MODELS>::type;
#else
using Models = FunctorTypes<UserModel>;
using First = UserModel;
#endif
}  // namespace user_models_
template <class T>
struct UserModelAdapter {
  using Models = user_models_::Models;
  static constexpr int n = user_models_::First::n, m = user_models_::First::m;
  static constexpr bool time_varying = true;  // (the adapter always takes the knot time; the model decides whether it reads it)
  int dof = 0;
  mutable int which = 0;  // the model of the knot being evaluated (Instance::Dynamics sets it)
  void f(const T* x, const T* u, float t, T* xd) const {
    Pick<Models>::at(which, [&](auto* p) {
      using S = std::remove_pointer_t<decltype(p)>;
      if constexpr (UserDiscrete<S>::value) std::fill(xd, xd + n, T(0));  // (never reached: Instance::Dynamics takes step())
      else if constexpr (UserTimeVarying<S>::value) S::f(x, u, t, xd);
      else S::f(x, u, xd);
    });
  }
  void jac(const T* x, const T* u, float t, T* J) const {
    Pick<Models>::at(which, [&](auto* p) {
      using S = std::remove_pointer_t<decltype(p)>;
      if constexpr (UserDiscrete<S>::value) std::fill(J, J + n * (n + m), T(0));
      else if constexpr (UserTimeVarying<S>::value) S::jac(x, u, t, J);
      else S::jac(x, u, J);
    });
  }
  // 0: RungeKutta4, 1: ExplicitEuler, 2: the model's own DiscreteDynamics
  int Kind() const {
    int kind = 0;
    Pick<Models>::at(which, [&](auto* p) {
      using S = std::remove_pointer_t<decltype(p)>;
      kind = UserDiscrete<S>::value ? 2 : (UserIntegrator<S>::value == 1 ? 1 : 0);
    });
    return kind;
  }
  void step(const T* x, const T* u, float t, float h, T* xn) const {
    Pick<Models>::at(which, [&](auto* p) {
      using S = std::remove_pointer_t<decltype(p)>;
      if constexpr (UserDiscrete<S>::value) S::step(x, u, t, h, xn);
    });
  }
  void step_jac(const T* x, const T* u, float t, float h, T* J) const {
    Pick<Models>::at(which, [&](auto* p) {
      using S = std::remove_pointer_t<decltype(p)>;
      if constexpr (UserDiscrete<S>::value) S::step_jac(x, u, t, h, J);
    });
  }
  static int Count() { return Models::count; }
};
#endif

// ------------------------------------------------------------------------------------------------
// Host-side problem specification (dtype independent, fp64).
// ------------------------------------------------------------------------------------------------
struct CostSpec {
  int k_begin, k_end;
  std::vector<double> Q, R, xref, uref;
  int per_instance;
  int user = 0;  // oracle_set_user_cost_type: 1 + index of the user's cost type, with `params`
  std::vector<double> params;
};
struct ConSpec {
  int kind, k_begin, k_end, nparams, per_instance;
  std::vector<double> params;
  int user_type = 0;  // ALTRO_CON_USER: index of the user's constraint type
};

// SolverStats Log / NewIteration semantics (altro/common/solver_stats.hpp:133-136,192-203,
// solver_stats.cpp:31-66): every Log writes the LAST row; NewIteration appends a copy of it.
enum Field { F_COST = 0, F_ALPHA, F_Z, F_GRAD, F_DJ, F_REG, F_VIOL, F_PEN, F_COUNT };
struct Stats {
  double initial_cost = 0.0;
  int iterations_inner = 0, iterations_outer = 0, iterations_total = 0;
  int len = 0;
  std::vector<double> v[F_COUNT];
  void Reset() {
    initial_cost = 0.0;
    iterations_inner = iterations_outer = iterations_total = 0;
    len = 0;
    for (auto& x : v) x.clear();
  }
  void NewIteration() {
    len++;
    for (auto& x : v) {
      x.resize(len);
      x.back() = (len > 1) ? x[len - 2] : 0.0;
    }
  }
  void Touch() {  // Log of a non-float entry ("iters", "iter_al") still registers the first row
    if (len == 0) NewIteration();
  }
  void Log(Field f, double value) {
    if (len == 0) NewIteration();
    v[f].back() = value;
  }
  double Back(Field f) const { return v[f].empty() ? 0.0 : v[f].back(); }
};

struct SolverBase {
  virtual ~SolverBase() {}
  virtual void AlInit() = 0;
  virtual void SolveSetup() = 0;
  virtual void Rollout() = 0;
  virtual double Cost() = 0;
  virtual void UpdateExpansions() = 0;
  virtual void BackwardPass() = 0;
  virtual void ForwardPass() = 0;
  virtual void UpdateConvergenceStatistics() = 0;
  virtual void UpdateDuals() = 0;
  virtual void UpdatePenalties() = 0;
  virtual double GetMaxViolation() = 0;
  virtual double MaxViolation() = 0;
  virtual double GetMaxPenalty() = 0;
  virtual void SolveILQR() = 0;
  virtual void SolveAL() = 0;
  virtual void SetPenalty(double rho) = 0;
  virtual void SetPenaltyScaling(double phi) = 0;
  virtual void SetTrajectory(const double* X, const double* U) = 0;
  virtual void SetInitialState(const double* x0) = 0;
  virtual void GetTrajectory(double* X, double* U) = 0;
  virtual void GetGains(double* K, double* d) = 0;
  virtual void GetCtg(double* P, double* p) = 0;
  virtual void GetExpansion(int k, double* AB, double* lxx, double* lxu, double* luu, double* lx,
                            double* lu) = 0;
  virtual void GetKnotCosts(double* c) = 0;
  virtual int NumRows() = 0;
  virtual int NumRowsAt(int k) = 0;
  virtual void GetDuals(double* out) = 0;
  virtual void SetDuals(const double* in) = 0;
  virtual void GetPenalties(double* out) = 0;
  virtual void GetConVals(double* out) = 0;
  virtual void GetStats(altro_stats* s) = 0;
  virtual Stats& RawStats() = 0;
  virtual void SetRoundRecords(bool on) = 0;
  // fp32-trial study: `shadow` = an all-fp32 instance of the same problem that evaluates the trials
  virtual void SetShadow(SolverBase* shadow, bool recheck) = 0;
  virtual void ShadowLoad(const double* X, const double* U, const double* K, const double* d, const double* lam,
                          const double* pen) = 0;
  virtual bool ShadowTrial(double alpha, double* J, double* cvals) = 0;
  virtual void SetConVals(const double* in) = 0;
  long long trials_f32 = 0, reevaluations = 0, recheck_rejections = 0;  // counters of the study
  altro_options opts;
};

// Time-varying dynamics (ContinuousDynamics::Evaluate(x, u, t, xdot), altro/problem/dynamics.hpp:59-95): a model that
// declares `static constexpr bool time_varying = true` takes a float time; the others never see it.
template <class M, class = void>
struct ModelTimeVarying : std::false_type {};
template <class M>
struct ModelTimeVarying<M, std::void_t<decltype(M::time_varying)>> : std::integral_constant<bool, M::time_varying> {};
template <class M, class T>
void ModelF(const M& mdl, const T* x, const T* u, float t, T* xd) {
  if constexpr (ModelTimeVarying<M>::value) mdl.f(x, u, t, xd);
  else mdl.f(x, u, xd);
}
template <class M, class T>
void ModelJac(const M& mdl, const T* x, const T* u, float t, T* J) {
  if constexpr (ModelTimeVarying<M>::value) mdl.jac(x, u, t, J);
  else mdl.jac(x, u, J);
}

// ------------------------------------------------------------------------------------------------
// One problem instance.
// ------------------------------------------------------------------------------------------------
template <class T, class Model>
struct Instance final : SolverBase {
  static constexpr int n = Model::n, m = Model::m, nm = n + m;
  Model model;
  int N;
  std::vector<float> h;  // N+1 entries, h[N] = 0 (trajectory.hpp:122-130)
  std::vector<float> tm;  // N+1 knot times (KnotPoint::t_, knotpoint.hpp:179)
  std::vector<int> km;    // N+1: model of the user source's list that knot k uses (Problem::SetDynamics(model, k)); all 0 by default

  struct QCost {  // examples/quadratic_cost.hpp:13-27
    T Q[n * n], R[m * m], H[n * m], q[n], r[m], c;
    int user = 0;         // 1 + t: the user's t-th problem::CostFunction subclass instead
    std::vector<T> upar;  // its parameters
  };
  struct Con {  // altro/constraints/constraint_values.hpp:39-51
    int kind, type /*0 equality, 1 inequality*/, p;
    int utype = 0;              // USER: index of the user's constraint class
    std::vector<T> par;         // GOAL: xf[n]; CIRCLE: (cx,cy,r)*; BOUND: finite lb values, finite ub values
    std::vector<int> lo, hi;    // BOUND: finite index lists (basic_constraints.hpp:138-145)
    std::vector<T> c, lam, pen; // c_, lambda_, penalty_
    T phi = T(10);              // kDefaultPenaltyScaling, constraint_values.hpp:30
  };
  std::vector<QCost> cost;             // N+1
  std::vector<std::vector<Con>> cons;  // N+1, equalities first then inequalities (al_cost.hpp:267-272)

  std::vector<T> x0;
  std::vector<T> X, U, Xb, Ub;  // (N+1)*n, (N+1)*m (u_N == 0)
  // expansions
  std::vector<T> AB, lxx, lxu, luu, lx, lu, costs;
  // gains and cost-to-go (knot_point_function_type.hpp:293-298)
  std::vector<T> K, d, P, p;
  T deltaV[2] = {0, 0};
  T rho_ = 0, drho_ = 0;
  std::vector<T> scratch_jac_, scratch_jp_, scratch_lp_;
  int status_ = ALTRO_UNSOLVED;     // iLQR status_ (ilqr.hpp:798)
  int status_al_ = ALTRO_UNSOLVED;  // AL status_ (al_solver.hpp:222)
  Stats stats;
  // ORACLE_F64_F32REC (not a reference mode): emulates the product's ALTRO_F32 engine, which computes in
  // fp64 and keeps only the two bulk per-knot RECORDS -- the expansion ([A|B], lxx, lxu, luu, lx, lu) and the
  // gains (K, d) -- in fp32.  The fp64 restatement below is unchanged; the stored records are rounded to
  // float at the points where the device stores them.
  bool round_records = false;
  void RoundRec(T* v, int count) const {
    // (the empty asm keeps g++ 11 -O3 from vectorising this loop: its 4-wide epilogue drops the
    // double -> float -> double conversion and leaves elements 8..11 of a 15-element record unrounded)
    if (round_records)
      for (int i = 0; i < count; ++i) {
        float f = (float)v[i];
        asm volatile("" : "+x"(f));
        v[i] = (T)f;
      }
  }

  Instance(int N_, const Model& mdl) : model(mdl), N(N_) {
    h.assign(N + 1, 0.0f);
    tm.assign(N + 1, 0.0f);
    km.assign(N + 1, 0);
    cost.resize(N + 1);
    cons.resize(N + 1);
    x0.assign(n, T(0));
    X.assign((N + 1) * n, T(0));
    U.assign((N + 1) * m, T(0));
    Xb = X;
    Ub = U;
    AB.assign((N + 1) * n * nm, T(0));
    lxx.assign((N + 1) * n * n, T(0));
    lxu.assign((N + 1) * n * m, T(0));
    luu.assign((N + 1) * m * m, T(0));
    lx.assign((N + 1) * n, T(0));
    lu.assign((N + 1) * m, T(0));
    costs.assign(N + 1, T(0));
    K.assign((N + 1) * m * n, T(0));  // zero-initialised: knot_point_function_type.hpp:271-278
    d.assign((N + 1) * m, T(0));
    P.assign((N + 1) * n * n, T(0));
    p.assign((N + 1) * n, T(0));
    altro_options o;
    std::memset(&o, 0, sizeof(o));
    opts = o;
  }
  Stats& RawStats() override { return stats; }
  void SetRoundRecords(bool on) override { round_records = on; }
  // ---- fp32-trial study (ORACLE_F32_TRIALS*) ---------------------------------------------------------------------
  SolverBase* shadow_ = nullptr;
  bool shadow_recheck_ = false;
  void SetShadow(SolverBase* sh, bool recheck) override {
    shadow_ = sh;
    shadow_recheck_ = recheck;
  }
  void ShadowLoad(const double* Xi, const double* Ui, const double* Ki, const double* di, const double* lam,
                  const double* pen) override {
    for (int i = 0; i < (N + 1) * n; ++i) X[i] = T(Xi[i]);
    for (int i = 0; i < N * m; ++i) U[i] = T(Ui[i]);
    for (int i = 0; i < N * m * n; ++i) K[i] = T(Ki[i]);
    for (int i = 0; i < N * m; ++i) d[i] = T(di[i]);
    ForRows([&](int r, Con& c, int i) {
      c.lam[i] = T(lam[r]);
      c.pen[i] = T(pen[r]);
    });
  }
  bool ShadowTrial(double alpha, double* J, double* cvals) override {
    if (!RolloutClosedLoop(T(alpha))) return false;
    *J = (double)CostOf(Xb, Ub);
    ForRows([&](int r, Con& c, int i) { cvals[r] = (double)c.c[i]; });
    return true;
  }
  void SetConVals(const double* in) override {
    ForRows([&](int r, Con& c, int i) { c.c[i] = T(in[r]); });
  }

  // ---- problem definition -----------------------------------------------------------------
  void SetLQRCost(int k, const double* Q, const double* R, const double* xref, const double* uref) {
    // QuadraticCost::LQRCost, examples/quadratic_cost.hpp:29-39
    QCost& c = cost[k];
    c.user = 0;
    for (int i = 0; i < n * n; ++i) c.Q[i] = T(Q[i]);
    for (int i = 0; i < m * m; ++i) c.R[i] = T(R[i]);
    for (int i = 0; i < n * m; ++i) c.H[i] = T(0);
    T xr[n], ur[m], Qx[n], Ru[m];
    for (int i = 0; i < n; ++i) xr[i] = T(xref[i]);
    for (int i = 0; i < m; ++i) ur[i] = T(uref[i]);
    for (int i = 0; i < n; ++i) {
      T s = 0;
      for (int j = 0; j < n; ++j) s += c.Q[i + j * n] * xr[j];
      Qx[i] = s;
      c.q[i] = -s;
    }
    for (int i = 0; i < m; ++i) {
      T s = 0;
      for (int j = 0; j < m; ++j) s += c.R[i + j * m] * ur[j];
      Ru[i] = s;
      c.r[i] = -s;
    }
    T a = 0, b = 0;
    for (int i = 0; i < n; ++i) a += xr[i] * Qx[i];
    for (int i = 0; i < m; ++i) b += ur[i] * Ru[i];
    c.c = T(0.5) * a + T(0.5) * b;
  }
  void SetUserCost(int k, int type, const double* par, int npar) {
    QCost& c = cost[k];
    c.user = 1 + type;
    c.upar.assign(par, par + npar);
  }
  void AddConstraint(int k, int kind, const double* par, int npar, int utype = 0) {
    Con c;
    c.kind = kind;
    if (kind == ALTRO_CON_USER) {
      int np = 0, rows = 0;
      bool eq = false;
      UserConShape(utype, &np, &rows, &eq);  // (oracle_add_constraint_type has checked that the type exists)
      c.utype = utype;
      c.type = eq ? 0 : 1;
      c.p = rows;
      c.par.assign(par, par + npar);
    } else if (kind == ALTRO_CON_GOAL) {
      c.type = 0;
      c.p = n;
      c.par.resize(n);
      for (int i = 0; i < n; ++i) c.par[i] = T(par[i]);
    } else if (kind == ALTRO_CON_CONTROL_BOUND) {
      c.type = 1;
      // GetFiniteIndices, basic_constraints.hpp:138-145
      for (int j = 0; j < m; ++j)
        if (std::abs(par[j]) < std::numeric_limits<double>::max()) c.lo.push_back(j);
      for (int j = 0; j < m; ++j)
        if (std::abs(par[m + j]) < std::numeric_limits<double>::max()) c.hi.push_back(j);
      for (int j : c.lo) c.par.push_back(T(par[j]));
      for (int j : c.hi) c.par.push_back(T(par[m + j]));
      c.p = (int)(c.lo.size() + c.hi.size());
    } else {  // CIRCLE
      c.type = 1;
      c.p = npar / 3;
      c.par.resize(npar);
      for (int i = 0; i < npar; ++i) c.par[i] = T(par[i]);
    }
    c.c.assign(c.p, T(0));
    c.lam.assign(c.p, T(0));
    c.pen.assign(c.p, T(1));  // penalty_.setOnes, constraint_values.hpp:44
    // keep equalities before inequalities, insertion order within each (al_cost.hpp:267-272)
    std::vector<Con>& v = cons[k];
    if (c.type == 0) {
      size_t pos = 0;
      while (pos < v.size() && v[pos].type == 0) ++pos;
      v.insert(v.begin() + pos, std::move(c));
    } else {
      v.push_back(std::move(c));
    }
  }
  void SetInitialState(const double* x) override {
    for (int i = 0; i < n; ++i) x0[i] = T(x[i]);
  }
  void SetTrajectory(const double* Xin, const double* Uin) override {
    for (int i = 0; i < (N + 1) * n; ++i) X[i] = Xin ? T(Xin[i]) : T(0);
    for (int i = 0; i < (N + 1) * m; ++i) U[i] = T(0);
    if (Uin)
      for (int i = 0; i < N * m; ++i) U[i] = T(Uin[i]);
    // SetTrajectory: Zbar_ = copy of Z_, SetZero (ilqr.hpp:231-235)
    std::fill(Xb.begin(), Xb.end(), T(0));
    std::fill(Ub.begin(), Ub.end(), T(0));
  }

  // ---- constraint evaluation ---------------------------------------------------------------
  // con_->Evaluate (basic_constraints.hpp:27-31,98-111; obstacle_constraints.hpp:99-107)
  static void ConEval(Con& c, const T* x, const T* u) {
    if (c.kind == ALTRO_CON_USER) {
      if constexpr (IsUserModel<Model>::value)
        Pick<UserCons>::at(c.utype, [&](auto* f) { std::remove_pointer_t<decltype(f)>::eval(x, u, c.par.data(), c.c.data()); });
    } else if (c.kind == ALTRO_CON_GOAL) {
      for (int i = 0; i < n; ++i) c.c[i] = x[i] - c.par[i];
    } else if (c.kind == ALTRO_CON_CONTROL_BOUND) {
      int nl = (int)c.lo.size();
      for (int i = 0; i < nl; ++i) c.c[i] = c.par[i] - u[c.lo[i]];
      for (size_t i = 0; i < c.hi.size(); ++i) c.c[nl + i] = u[c.hi[i]] - c.par[nl + i];
    } else {
      T px = x[0], py = x[1];
      for (int i = 0; i < c.p; ++i) {
        T dx = px - c.par[3 * i], dy = py - c.par[3 * i + 1], r = c.par[3 * i + 2];
        c.c[i] = -(dx * dx + dy * dy - r * r);  // -Distance2, obstacle_constraints.hpp:42-44
      }
    }
  }
  // con_->Jacobian; jac is p x (n+m), row r at jac[r*nm .. r*nm+nm)
  static void ConJac(const Con& c, const T* x, const T* u, T* jac) {
    for (int i = 0; i < c.p * nm; ++i) jac[i] = T(0);
    if (c.kind == ALTRO_CON_USER) {
      std::vector<T> Jcm((size_t)c.p * nm, T(0));  // the user's Jacobian is p x (n+m) column-major (Eigen's default)
      if constexpr (IsUserModel<Model>::value)
        Pick<UserCons>::at(c.utype, [&](auto* f) { std::remove_pointer_t<decltype(f)>::jacobian(x, u, c.par.data(), Jcm.data()); });
      for (int r = 0; r < c.p; ++r)
        for (int j = 0; j < nm; ++j) jac[r * nm + j] = Jcm[r + j * c.p];
    } else if (c.kind == ALTRO_CON_GOAL) {
      for (int i = 0; i < n; ++i) jac[i * nm + i] = T(1);
    } else if (c.kind == ALTRO_CON_CONTROL_BOUND) {
      int nl = (int)c.lo.size();
      for (int i = 0; i < nl; ++i) jac[i * nm + n + c.lo[i]] = T(-1);
      for (size_t i = 0; i < c.hi.size(); ++i) jac[(nl + i) * nm + n + c.hi[i]] = T(1);
    } else {
      T px = x[0], py = x[1];
      for (int i = 0; i < c.p; ++i) {
        jac[i * nm + 0] = 2 * (c.par[3 * i] - px);  // obstacle_constraints.hpp:117-118
        jac[i * nm + 1] = 2 * (c.par[3 * i + 1] - py);
      }
    }
  }
  // DualCone::Projection (constraint.hpp:70-73 identity for equalities, :103-108 for inequalities)
  static T DualProj(const Con& c, T v) { return c.type == 0 ? v : std::min(T(0), v); }
  // DualCone::Jacobian diagonal (constraint.hpp:74-78, :109-114; quirk Q5: v == 0 is active)
  static T DualProjJac(const Con& c, T v) { return c.type == 0 ? T(1) : (v > 0 ? T(0) : T(1)); }

  // ConstraintValues::AugLag, constraint_values.hpp:111-119 (quirk Q2: scalar rho = penalty_(0))
  static T AugLag(Con& c, const T* x, const T* u) {
    const T rho = c.pen[0];
    ConEval(c, x, u);
    T a = 0, b = 0;
    for (int i = 0; i < c.p; ++i) {
      T lp = DualProj(c, c.lam[i] - rho * c.c[i]);
      a += lp * lp;
      b += c.lam[i] * c.lam[i];
    }
    T J = a - b;
    return J / (2 * rho);
  }

  // QuadraticCost::Evaluate, examples/quadratic_cost.cpp:8-11
  static T QuadEval(const QCost& c, const T* x, const T* u) {
    if constexpr (IsUserModel<Model>::value) {
      if (c.user) {  // CostFunction::Evaluate of the user's class
        T J = T(0);
        Pick<UserCosts>::at(c.user - 1, [&](auto* f) { J = std::remove_pointer_t<decltype(f)>::eval(x, u, c.upar.data()); });
        return J;
      }
    }
    T xQx = 0, xHu = 0, uRu = 0, qx = 0, ru = 0;
    for (int i = 0; i < n; ++i) {
      T s = 0;
      for (int j = 0; j < n; ++j) s += c.Q[i + j * n] * x[j];
      xQx += x[i] * s;
      T t = 0;
      for (int j = 0; j < m; ++j) t += c.H[i + j * n] * u[j];
      xHu += x[i] * t;
      qx += c.q[i] * x[i];
    }
    for (int i = 0; i < m; ++i) {
      T s = 0;
      for (int j = 0; j < m; ++j) s += c.R[i + j * m] * u[j];
      uRu += u[i] * s;
      ru += c.r[i] * u[i];
    }
    return T(0.5) * xQx + xHu + T(0.5) * uRu + qx + ru + c.c;
  }

  // ALCost::Evaluate, al_cost.hpp:264-274.  Side effect: stores c_ of every constraint (quirk Q6).
  T KnotCost(int k, const T* x, const T* u) {
    T J = QuadEval(cost[k], x, u);
    for (Con& c : cons[k]) J += AugLag(c, x, u);
    return J;
  }

  // CostExpansion::CalcExpansion -> ALCost::Gradient / Hessian (cost_expansion.hpp:118-125,
  // al_cost.hpp:276-308, quadratic_cost.cpp:13-28, constraint_values.hpp:131-177)
  void CostExpansion(int k, const T* x, const T* u) {
    const QCost& qc = cost[k];
    T* gx = &lx[k * n];
    T* gu = &lu[k * m];
    T* hxx = &lxx[k * n * n];
    T* hxu = &lxu[k * n * m];
    T* huu = &luu[k * m * m];
    if (IsUserModel<Model>::value && qc.user) {  // CostFunction::Gradient / Hessian of the user's cost (costfunction.hpp:59-73)
      if constexpr (IsUserModel<Model>::value) {
        Pick<UserCosts>::at(qc.user - 1, [&](auto* f) {
          using F = std::remove_pointer_t<decltype(f)>;
          F::gradient(x, u, qc.upar.data(), gx, gu);
          F::hessian(x, u, qc.upar.data(), hxx, hxu, huu);
        });
      }
    } else {
      for (int i = 0; i < n; ++i) {
        T s = 0, t = 0;
        for (int j = 0; j < n; ++j) s += qc.Q[i + j * n] * x[j];
        for (int j = 0; j < m; ++j) t += qc.H[i + j * n] * u[j];
        gx[i] = s + qc.q[i] + t;
      }
      for (int i = 0; i < m; ++i) {
        T s = 0, t = 0;
        for (int j = 0; j < m; ++j) s += qc.R[i + j * m] * u[j];
        for (int j = 0; j < n; ++j) t += qc.H[j + i * n] * x[j];
        gu[i] = s + qc.r[i] + t;
      }
      for (int i = 0; i < n * n; ++i) hxx[i] = qc.Q[i];
      for (int i = 0; i < n * m; ++i) hxu[i] = qc.H[i];
      for (int i = 0; i < m * m; ++i) huu[i] = qc.R[i];
    }
    // scratch reused across calls (one solver instance is only ever used by one thread at a time)
    std::vector<T>& jac = scratch_jac_;
    std::vector<T>& jp = scratch_jp_;
    std::vector<T>& lp = scratch_lp_;
    for (Con& c : cons[k]) {
      const T rho = c.pen[0];
      ConEval(c, x, u);
      if ((int)jac.size() < c.p * nm) {
        jac.resize(c.p * nm);
        jp.resize(c.p * nm);
      }
      if ((int)lp.size() < c.p) lp.resize(c.p);
      ConJac(c, x, u, jac.data());
      for (int r = 0; r < c.p; ++r) {
        T v = c.lam[r] - rho * c.c[r];
        lp[r] = DualProj(c, v);
        T pj = DualProjJac(c, v);
        for (int j = 0; j < nm; ++j) jp[r * nm + j] = pj * jac[r * nm + j];
      }
      // gradient: dx = -(P Cx)^T lambda_bar, du likewise (constraint_values.hpp:141-142)
      for (int i = 0; i < n; ++i) {
        T s = 0;
        for (int r = 0; r < c.p; ++r) s += jp[r * nm + i] * lp[r];
        gx[i] += -s;
      }
      for (int i = 0; i < m; ++i) {
        T s = 0;
        for (int r = 0; r < c.p; ++r) s += jp[r * nm + n + i] * lp[r];
        gu[i] += -s;
      }
      // Gauss-Newton Hessian rho (PC)^T (PC) (constraint_values.hpp:165-172)
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
          T s = 0;
          for (int r = 0; r < c.p; ++r) s += (rho * jp[r * nm + i]) * jp[r * nm + j];
          hxx[i + j * n] += s;
        }
      for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i) {
          T s = 0;
          for (int r = 0; r < c.p; ++r) s += (rho * jp[r * nm + i]) * jp[r * nm + n + j];
          hxu[i + j * n] += s;
        }
      for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i) {
          T s = 0;
          for (int r = 0; r < c.p; ++r) s += (rho * jp[r * nm + n + i]) * jp[r * nm + n + j];
          huu[i + j * m] += s;
        }
    }
  }

  // ---- dynamics ------------------------------------------------------------------------------
  // RungeKutta4::Integrate, altro/problem/integration.hpp:123-131
  // (the stage times: `t + 0.5 * h` with float t, h is evaluated in double and narrowed to Evaluate's float parameter)
  // (`which`: the knot's model when the user source lists several; the other DiscreteDynamics kinds of a user model --
  //  ExplicitEuler::Integrate, integration.hpp:90-94, and the caller's own Evaluate(x, u, t, h, xnext), dynamics.hpp:148-187)
  void Dynamics(const T* x, const T* u, float hf, T* xn, float t = 0.0f, int which = 0) const {
    const T hh = T(hf);
    if constexpr (IsUserModel<Model>::value) {
      model.which = which;
      const int kind = model.Kind();
      if (kind == 2) {
        model.step(x, u, t, hf, xn);
        return;
      }
      if (kind == 1) {
        T xd[n];
        model.f(x, u, t, xd);
        for (int i = 0; i < n; ++i) xn[i] = x[i] + xd[i] * hh;
        return;
      }
    }
    const float th = (float)((double)t + 0.5 * (double)hf), t1 = (float)((double)t + (double)hf);
    T k1[n] = {}, k2[n] = {}, k3[n] = {}, k4[n] = {}, xt[n];  // (zeroed: a user-model index outside the list evaluates nothing)
    ModelF(model, x, u, t, k1);
    for (int i = 0; i < n; ++i) xt[i] = x[i] + k1[i] * T(0.5) * hh;
    ModelF(model, xt, u, th, k2);
    for (int i = 0; i < n; ++i) xt[i] = x[i] + k2[i] * T(0.5) * hh;
    ModelF(model, xt, u, th, k3);
    for (int i = 0; i < n; ++i) xt[i] = x[i] + k3[i] * hh;
    ModelF(model, xt, u, t1, k4);
    for (int i = 0; i < n; ++i) xn[i] = x[i] + hh * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) / 6;
  }
  // RungeKutta4::Jacobian, integration.hpp:132-169
  // (the middle Jacobians are taken at time 0.5 * t and the last one at t -- integration.hpp:144-150, as written)
  void DynamicsJacobian(const T* x, const T* u, float hf, T* J, float t = 0.0f, int which = 0) const {
    const T hh = T(hf);
    if constexpr (IsUserModel<Model>::value) {
      model.which = which;
      const int kind = model.Kind();
      if (kind == 2) {  // DiscreteDynamics::Jacobian(x, u, t, h, jac)
        model.step_jac(x, u, t, hf, J);
        return;
      }
      if (kind == 1) {  // ExplicitEuler::Jacobian (integration.hpp:95-101): Identity(n, n + m) + jac * h
        model.jac(x, u, t, J);
        for (int j = 0; j < nm; ++j)
          for (int i = 0; i < n; ++i) J[i + j * n] = (i == j ? T(1) : T(0)) + J[i + j * n] * hh;
        return;
      }
    }
    const float th = (float)((double)t + 0.5 * (double)hf), tj = (float)(0.5 * (double)t);
    T k1[n] = {}, k2[n] = {}, k3[n] = {}, xt[n];
    T Jc[4][n * nm] = {};
    ModelF(model, x, u, t, k1);
    for (int i = 0; i < n; ++i) xt[i] = x[i] + k1[i] * T(0.5) * hh;
    ModelF(model, xt, u, th, k2);
    for (int i = 0; i < n; ++i) xt[i] = x[i] + k2[i] * T(0.5) * hh;
    ModelF(model, xt, u, th, k3);
    ModelJac(model, x, u, t, Jc[0]);
    for (int i = 0; i < n; ++i) xt[i] = x[i] + T(0.5) * k1[i] * hh;
    ModelJac(model, xt, u, tj, Jc[1]);
    for (int i = 0; i < n; ++i) xt[i] = x[i] + T(0.5) * k2[i] * hh;
    ModelJac(model, xt, u, tj, Jc[2]);
    for (int i = 0; i < n; ++i) xt[i] = x[i] + k3[i] * hh;
    ModelJac(model, xt, u, t, Jc[3]);
    const T* A[4];
    const T* B[4];
    for (int s = 0; s < 4; ++s) {
      A[s] = Jc[s];
      B[s] = Jc[s] + n * n;
    }
    T dA[4][n * n], dB[4][n * m], M[n * n];
    auto matmul_nn = [](const T* a, const T* b, T* c) {  // c = a*b, all n x n
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
          T s = 0;
          for (int l = 0; l < n; ++l) s += a[i + l * n] * b[l + j * n];
          c[i + j * n] = s;
        }
    };
    auto matmul_nm = [](const T* a, const T* b, T* c) {  // c = a(n x n) * b(n x m)
      for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i) {
          T s = 0;
          for (int l = 0; l < n; ++l) s += a[i + l * n] * b[l + j * n];
          c[i + j * n] = s;
        }
    };
    for (int i = 0; i < n * n; ++i) dA[0][i] = A[0][i] * hh;
    const T coef[4] = {T(0), T(0.5), T(0.5), T(1)};
    for (int s = 1; s < 4; ++s) {
      // dA[s] = A[s] * (I + coef*dA[s-1]) * h
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) M[i + j * n] = (i == j ? T(1) : T(0)) + coef[s] * dA[s - 1][i + j * n];
      T tmp[n * n];
      matmul_nn(A[s], M, tmp);
      for (int i = 0; i < n * n; ++i) dA[s][i] = tmp[i] * hh;
    }
    for (int i = 0; i < n * m; ++i) dB[0][i] = B[0][i] * hh;
    for (int s = 1; s < 4; ++s) {
      // dB[s] = B[s]*h + coef * A[s] * dB[s-1] * h
      T tmp[n * m];
      matmul_nm(A[s], dB[s - 1], tmp);
      for (int i = 0; i < n * m; ++i) dB[s][i] = B[s][i] * hh + coef[s] * tmp[i] * hh;
    }
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) {
        int e = i + j * n;
        J[e] = (i == j ? T(1) : T(0)) + (dA[0][e] + 2 * dA[1][e] + 2 * dA[2][e] + dA[3][e]) / 6;
      }
    for (int e = 0; e < n * m; ++e)
      J[n * n + e] = (dB[0][e] + 2 * dB[1][e] + 2 * dB[2][e] + dB[3][e]) / 6;
  }

  // ---- iLQR -----------------------------------------------------------------------------------
  // iLQR::Rollout, ilqr.hpp:453-459
  void Rollout() override {
    for (int i = 0; i < n; ++i) X[i] = x0[i];
    for (int k = 0; k < N; ++k) Dynamics(&X[k * n], &U[k * m], h[k], &X[(k + 1) * n], tm[k], km[k]);
  }
  // iLQR::Cost / CalcIndividualCosts, ilqr.hpp:326-334, 758-763
  T CostOf(const std::vector<T>& Xs, const std::vector<T>& Us) {
    T J = 0;
    for (int k = 0; k <= N; ++k) {
      costs[k] = KnotCost(k, &Xs[k * n], &Us[k * m]);
      J += costs[k];
    }
    return J;
  }
  double Cost() override { return (double)CostOf(X, U); }

  // iLQR::UpdateExpansionsBlock, ilqr.hpp:670-677
  void UpdateExpansions() override {
    for (int k = 0; k <= N; ++k) {
      const T* x = &X[k * n];
      const T* u = &U[k * m];
      CostExpansion(k, x, u);
      if (k < N) DynamicsJacobian(x, u, h[k], &AB[k * n * nm], tm[k], km[k]);
      costs[k] = KnotCost(k, x, u);
      if (round_records) {
        RoundRec(&AB[k * n * nm], n * nm);
        RoundRec(&lxx[k * n * n], n * n);
        RoundRec(&lxu[k * n * m], n * m);
        RoundRec(&luu[k * m * m], m * m);
        RoundRec(&lx[k * n], n);
        RoundRec(&lu[k * m], m);
      }
    }
  }

  // iLQR::IncreaseRegularization / DecreaseRegularization, ilqr.hpp:770-786
  void IncreaseRegularization() {
    drho_ = std::max(drho_ * T(opts.bp_reg_increase_factor), T(opts.bp_reg_increase_factor));
    rho_ = std::max(rho_ * drho_, T(opts.bp_reg_min));
    rho_ = std::min(rho_, T(opts.bp_reg_max));
  }
  void DecreaseRegularization() {
    drho_ = std::min(drho_ / T(opts.bp_reg_increase_factor), 1 / T(opts.bp_reg_increase_factor));
    rho_ = std::max(rho_ * drho_, T(opts.bp_reg_min));
    rho_ = std::min(rho_, T(opts.bp_reg_max));
  }

  // One knot of the backward pass: CalcActionValueExpansion, RegularizeActionValue, CalcGains,
  // CalcCostToGo, AddCostToGo (knot_point_function_type.hpp:149-235).  Returns false on Cholesky
  // failure (Eigen::NumericalIssue), in which case K, d, P, p of this knot are left unchanged.
  bool BackwardKnot(int k, const T* Pn, const T* pn) {
    const T* A = &AB[k * n * nm];
    const T* B = A + n * n;
    T AtP[n * n], BtP[m * n];
    for (int j = 0; j < n; ++j) {
      for (int i = 0; i < n; ++i) {
        T s = 0;
        for (int l = 0; l < n; ++l) s += A[l + i * n] * Pn[l + j * n];
        AtP[i + j * n] = s;
      }
      for (int i = 0; i < m; ++i) {
        T s = 0;
        for (int l = 0; l < n; ++l) s += B[l + i * n] * Pn[l + j * n];
        BtP[i + j * m] = s;
      }
    }
    T Qxx[n * n], Qxu[n * m], Quu[m * m], Qx[n], Qu[m];
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) {
        T s = 0;
        for (int l = 0; l < n; ++l) s += AtP[i + l * n] * A[l + j * n];
        Qxx[i + j * n] = lxx[k * n * n + i + j * n] + s;
      }
    for (int j = 0; j < m; ++j)
      for (int i = 0; i < n; ++i) {
        T s = 0;
        for (int l = 0; l < n; ++l) s += AtP[i + l * n] * B[l + j * n];
        Qxu[i + j * n] = lxu[k * n * m + i + j * n] + s;
      }
    for (int j = 0; j < m; ++j)
      for (int i = 0; i < m; ++i) {
        T s = 0;
        for (int l = 0; l < n; ++l) s += BtP[i + l * m] * B[l + j * n];
        Quu[i + j * m] = luu[k * m * m + i + j * m] + s;
      }
    for (int i = 0; i < n; ++i) {
      T s = 0;
      for (int l = 0; l < n; ++l) s += A[l + i * n] * pn[l];
      Qx[i] = lx[k * n + i] + s;
    }
    for (int i = 0; i < m; ++i) {
      T s = 0;
      for (int l = 0; l < n; ++l) s += B[l + i * n] * pn[l];
      Qu[i] = lu[k * m + i] + s;
    }
    // RegularizeActionValue (control-only): Quu_reg = Quu + rho I
    T L[m * m];
    for (int i = 0; i < m * m; ++i) L[i] = Quu[i];
    for (int i = 0; i < m; ++i) L[i + i * m] += rho_;
    // Eigen::LLT (lower): fails when a pivot is <= 0
    for (int j = 0; j < m; ++j) {
      T x = L[j + j * m];
      for (int l = 0; l < j; ++l) x -= L[j + l * m] * L[j + l * m];
      if (x <= T(0)) return false;
      T ljj = std::sqrt(x);
      L[j + j * m] = ljj;
      for (int i = j + 1; i < m; ++i) {
        T s = L[i + j * m];
        for (int l = 0; l < j; ++l) s -= L[i + l * m] * L[j + l * m];
        L[i + j * m] = s / ljj;
      }
    }
    auto solve = [&](T* b) {  // in place: b <- (L L^T)^-1 b
      for (int i = 0; i < m; ++i) {
        T s = b[i];
        for (int l = 0; l < i; ++l) s -= L[i + l * m] * b[l];
        b[i] = s / L[i + i * m];
      }
      for (int i = m - 1; i >= 0; --i) {
        T s = b[i];
        for (int l = i + 1; l < m; ++l) s -= L[l + i * m] * b[l];
        b[i] = s / L[i + i * m];
      }
    };
    T* Kk = &K[k * m * n];
    T* dk = &d[k * m];
    for (int j = 0; j < n; ++j) {  // K = -Quu_reg^-1 Qxu^T (regularised Q: quirk Q3)
      T col[m];
      for (int i = 0; i < m; ++i) col[i] = Qxu[j + i * n];
      solve(col);
      for (int i = 0; i < m; ++i) Kk[i + j * m] = -col[i];
    }
    {
      T col[m];
      for (int i = 0; i < m; ++i) col[i] = Qu[i];
      solve(col);
      for (int i = 0; i < m; ++i) dk[i] = -col[i];
    }
    // CalcCostToGo with the UN-regularised Q (knot_point_function_type.hpp:220-230)
    T KtQuu[n * m];
    for (int j = 0; j < m; ++j)
      for (int i = 0; i < n; ++i) {
        T s = 0;
        for (int l = 0; l < m; ++l) s += Kk[l + i * m] * Quu[l + j * m];
        KtQuu[i + j * n] = s;
      }
    T* pk = &p[k * n];
    T* Pk = &P[k * n * n];
    for (int i = 0; i < n; ++i) {
      T a = 0, b = 0, c = 0;
      for (int l = 0; l < m; ++l) {
        a += KtQuu[i + l * n] * dk[l];
        b += Kk[l + i * m] * Qu[l];
        c += Qxu[i + l * n] * dk[l];
      }
      pk[i] = Qx[i] + a + b + c;
    }
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) {
        T a = 0, b = 0, c = 0;
        for (int l = 0; l < m; ++l) {
          a += KtQuu[i + l * n] * Kk[l + j * m];
          b += Kk[l + i * m] * Qxu[j + l * n];
          c += Qxu[i + l * n] * Kk[l + j * m];
        }
        Pk[i + j * n] = Qxx[i + j * n] + a + b + c;
      }
    T dv0 = 0, dv1 = 0;
    for (int i = 0; i < m; ++i) {
      dv0 += dk[i] * Qu[i];
      T s = 0;
      for (int l = 0; l < m; ++l) s += Quu[i + l * m] * dk[l];
      dv1 += dk[i] * s;
    }
    deltaV[0] += dv0;
    deltaV[1] += T(0.5) * dv1;
    // the device keeps K, d in registers for the cost-to-go and rounds only the stored gain record
    RoundRec(Kk, m * n);
    RoundRec(dk, m);
    return true;
  }

  // iLQR::BackwardPass, ilqr.hpp:385-445
  void BackwardPass() override {
    // CalcTerminalCostToGo, knot_point_function_type.hpp:135-138
    for (int i = 0; i < n * n; ++i) P[N * n * n + i] = lxx[N * n * n + i];
    for (int i = 0; i < n; ++i) p[N * n + i] = lx[N * n + i];
    int max_reg_count = 0;
    deltaV[0] = 0;  // zeroed ONCE, not per retry (quirk Q4)
    deltaV[1] = 0;
    bool repeat = true;
    while (repeat) {
      const T* Pn = &P[N * n * n];
      const T* pn = &p[N * n];
      for (int k = N - 1; k >= 0; --k) {
        if (!BackwardKnot(k, Pn, pn)) {
          IncreaseRegularization();
          if (rho_ >= T(opts.bp_reg_max)) max_reg_count++;
          if (max_reg_count >= opts.bp_reg_fail_threshold) {
            status_ = ALTRO_BACKWARD_PASS_REGULARIZATION_FAILED;
            repeat = false;
          }
          break;
        }
        Pn = &P[k * n * n];
        pn = &p[k * n];
        if (k == 0) repeat = false;
      }
      if (N == 0) repeat = false;
    }
    stats.Log(F_REG, (double)rho_);
    DecreaseRegularization();
  }

  // iLQR::RolloutClosedLoop, ilqr.hpp:468-499
  bool RolloutClosedLoop(T alpha) {
    for (int i = 0; i < n; ++i) Xb[i] = x0[i];
    for (int k = 0; k < N; ++k) {
      const T* Kk = &K[k * m * n];
      const T* dk = &d[k * m];
      T dx[n];
      for (int i = 0; i < n; ++i) dx[i] = Xb[k * n + i] - X[k * n + i];
      for (int i = 0; i < m; ++i) {
        T s = 0;
        for (int l = 0; l < n; ++l) s += Kk[i + l * m] * dx[l];
        Ub[k * m + i] = U[k * m + i] + s + dk[i] * alpha;
      }
      Dynamics(&Xb[k * n], &Ub[k * m], h[k], &Xb[(k + 1) * n], tm[k], km[k]);
      if (opts.check_forwardpass_bounds) {
        T sx = 0, su = 0;
        for (int i = 0; i < n; ++i) sx += Xb[(k + 1) * n + i] * Xb[(k + 1) * n + i];
        for (int i = 0; i < m; ++i) su += Ub[k * m + i] * Ub[k * m + i];
        if (std::sqrt(sx) > T(opts.state_max)) {
          status_ = ALTRO_STATE_LIMIT;
          return false;
        }
        if (std::sqrt(su) > T(opts.control_max)) {
          status_ = ALTRO_CONTROL_LIMIT;
          return false;
        }
      }
    }
    status_ = ALTRO_UNSOLVED;
    return true;
  }

  // iLQR::ForwardPass, ilqr.hpp:512-558
  void ForwardPass() override {
    T J0 = 0;
    for (int k = 0; k <= N; ++k) J0 += costs[k];
    T alpha = 1, z = -1, J = J0;
    bool success = false;
    if (shadow_) {  // the fp32-trial study: see ORACLE_F32_TRIALS
      const int R = NumRows();
      std::vector<double> Xd((N + 1) * n), Ud((N + 1) * m), Kd(N * m * n), dd(N * m), lam(R), pen(R), cv(R), cv_last(R);
      GetTrajectory(Xd.data(), Ud.data());
      GetGains(Kd.data(), dd.data());
      GetDuals(lam.data());
      GetPenalties(pen.data());
      shadow_->ShadowLoad(Xd.data(), Ud.data(), Kd.data(), dd.data(), lam.data(), pen.data());
      bool any_trial = false;
      for (int it = 0; it < opts.line_search_max_iterations; ++it) {
        double J32 = 0.0;
        trials_f32++;
        if (shadow_->ShadowTrial((double)alpha, &J32, cv.data())) {
          any_trial = true;
          cv_last = cv;
          const T expected = -alpha * (deltaV[0] + alpha * deltaV[1]);
          const T z32 = expected > T(0) ? (J0 - T(J32)) / expected : T(-1);
          if (T(opts.line_search_lower_bound) <= z32 && z32 <= T(opts.line_search_upper_bound) && T(J32) < J0) {
            reevaluations++;
            const bool ok = RolloutClosedLoop(alpha);  // fp64, this instance: Xb, Ub, c_
            if (ok) {
              J = CostOf(Xb, Ub);
              z = expected > T(0) ? (J0 - J) / expected : T(-1);
              const bool pass = T(opts.line_search_lower_bound) <= z && z <= T(opts.line_search_upper_bound) && J < J0;
              if (pass || !shadow_recheck_) {
                success = true;
                any_trial = false;  // c_ is the fp64 evaluation of the accepted step
                stats.Log(F_COST, (double)J);
                stats.Log(F_ALPHA, (double)alpha);
                stats.Log(F_Z, (double)z);
                break;
              }
              recheck_rejections++;
            }
          }
        }
        alpha /= T(opts.line_search_decrease_factor);
      }
      if (any_trial) SetConVals(cv_last.data());  // stale c_ of the last trial (quirk Q6), here an fp32 evaluation
      if (success) {
        X = Xb;
        U = Ub;
      } else {
        IncreaseRegularization();
        J = J0;
      }
      if (J > J0) status_ = ALTRO_COST_INCREASE;
      return;
    }
    for (int it = 0; it < opts.line_search_max_iterations; ++it) {
      if (RolloutClosedLoop(alpha)) {
        J = CostOf(Xb, Ub);
        T expected = -alpha * (deltaV[0] + alpha * deltaV[1]);
        z = expected > T(0) ? (J0 - J) / expected : T(-1);
        if (T(opts.line_search_lower_bound) <= z && z <= T(opts.line_search_upper_bound) && J < J0) {
          success = true;
          stats.Log(F_COST, (double)J);
          stats.Log(F_ALPHA, (double)alpha);
          stats.Log(F_Z, (double)z);
          break;
        }
      }
      alpha /= T(opts.line_search_decrease_factor);
    }
    if (success) {
      X = Xb;
      U = Ub;
    } else {
      IncreaseRegularization();
      J = J0;
    }
    if (J > J0) status_ = ALTRO_COST_INCREASE;
  }

  // iLQR::NormalizedFeedforwardGain + UpdateConvergenceStatistics, ilqr.hpp:568-587, 662-668
  void UpdateConvergenceStatistics() override {
    T g = 0;
    for (int k = 0; k < N; ++k) {
      T mx = -std::numeric_limits<T>::infinity();
      for (int i = 0; i < m; ++i) mx = std::max(mx, std::abs(d[k * m + i]) / (std::abs(U[k * m + i]) + 1));
      g += mx;
    }
    double dgrad = N > 0 ? (double)(g / T(N)) : 0.0;
    double dJ;
    if (stats.iterations_inner == 0) {
      dJ = stats.initial_cost - stats.Back(F_COST);
    } else {
      dJ = stats.v[F_COST][stats.len - 2] - stats.v[F_COST][stats.len - 1];
    }
    stats.iterations_inner++;
    stats.iterations_total++;
    stats.Log(F_DJ, dJ);
    stats.Log(F_VIOL, GetMaxViolation());  // max_violation_callback_ (stored c_, quirk Q6)
    stats.Log(F_GRAD, dgrad);
    stats.NewIteration();
  }
  // iLQR::IsDone, ilqr.hpp:597-619
  bool IsDone() {
    bool cost_decrease = stats.Back(F_DJ) < opts.cost_tolerance;
    bool gradient = stats.Back(F_GRAD) < opts.gradient_tolerance;
    if (cost_decrease && gradient) {
      status_ = ALTRO_SOLVED;
      return true;
    } else if (stats.iterations_inner >= opts.max_iterations_inner) {
      status_ = ALTRO_MAX_INNER_ITERATIONS;
      return true;
    } else if (stats.iterations_total >= opts.max_iterations_total) {
      status_ = ALTRO_MAX_ITERATIONS;
      return true;
    } else if (status_ != ALTRO_UNSOLVED) {
      return true;
    }
    return false;
  }
  // iLQR::SolveSetup + ResetInternalVariables, ilqr.hpp:629-645, 680-690
  void SolveSetup() override {
    stats.iterations_inner = 0;
    status_ = ALTRO_UNSOLVED;
    std::fill(costs.begin(), costs.end(), T(0));
    deltaV[0] = deltaV[1] = 0;
    rho_ = T(opts.bp_reg_initial);
    drho_ = 0;
  }
  // iLQR::Solve, ilqr.hpp:284-316
  void SolveILQR() override {
    SolveSetup();
    Rollout();
    stats.initial_cost = Cost();
    for (int iter = 0; iter < opts.max_iterations_inner; ++iter) {
      UpdateExpansions();
      BackwardPass();
      ForwardPass();
      UpdateConvergenceStatistics();
      if (IsDone()) break;
    }
  }

  // ---- augmented Lagrangian outer loop -----------------------------------------------------
  void SetPenalty(double rho) override {  // al_solver.hpp:271-277
    for (auto& v : cons)
      for (Con& c : v) std::fill(c.pen.begin(), c.pen.end(), T(rho));
  }
  void SetPenaltyScaling(double phi) override {  // al_solver.hpp:279-285
    for (auto& v : cons)
      for (Con& c : v) c.phi = T(phi);
  }
  // ConstraintValues::UpdateDuals, constraint_values.hpp:192-194 (per-row penalty, stored c_)
  void UpdateDuals() override {
    for (auto& v : cons)
      for (Con& c : v)
        for (int i = 0; i < c.p; ++i) c.lam[i] = DualProj(c, c.lam[i] - c.pen[i] * c.c[i]);
  }
  void UpdatePenalties() override {  // constraint_values.hpp:202-207
    for (auto& v : cons)
      for (Con& c : v)
        for (int i = 0; i < c.p; ++i) c.pen[i] *= c.phi;
  }
  // ConstraintValues::MaxViolation / ALCost::MaxViolation / GetMaxViolation
  // (constraint_values.hpp:215-220, al_cost.hpp:343-353, al_solver.hpp:417-424)
  double GetMaxViolation() override {
    T mx = 0;
    for (auto& v : cons)
      for (Con& c : v)
        for (int i = 0; i < c.p; ++i) {
          T viol = c.type == 0 ? std::abs(c.c[i]) : std::abs(c.c[i] - std::min(T(0), c.c[i]));
          mx = std::max(mx, viol);
        }
    return (double)mx;
  }
  double MaxViolation() override {  // al_solver.hpp:403-408
    Cost();
    return GetMaxViolation();
  }
  double GetMaxPenalty() override {  // al_solver.hpp:426-434
    T mx = 0;
    for (auto& v : cons)
      for (Con& c : v)
        for (int i = 0; i < c.p; ++i) mx = std::max(mx, c.pen[i]);
    return (double)mx;
  }
  // AugmentedLagrangianiLQR::Init, al_solver.hpp:287-302
  void AlInit() override {
    if (opts.reset_duals)
      for (auto& v : cons)
        for (Con& c : v) std::fill(c.lam.begin(), c.lam.end(), T(0));
    if (opts.initial_penalty > 0) SetPenalty(opts.initial_penalty);  // quirk Q8
    stats.Reset();
    stats.Touch();  // Log("iter_al", 0)
    stats.Log(F_VIOL, MaxViolation());
    stats.Log(F_PEN, GetMaxPenalty());
  }
  // AugmentedLagrangianiLQR::IsDone, al_solver.hpp:368-401
  bool AlIsDone() {
    const bool satisfied = stats.Back(F_VIOL) < opts.constraint_tolerance;
    const bool max_pen = stats.Back(F_PEN) > opts.maximum_penalty;
    const bool max_outer = stats.iterations_outer >= opts.max_iterations_outer;
    const bool max_total = stats.iterations_total >= opts.max_iterations_total;
    if (status_ != ALTRO_SOLVED) {
      status_al_ = status_;
      return true;
    }
    if (satisfied) {
      status_al_ = ALTRO_SOLVED;
      return true;
    }
    if (max_pen) {
      status_al_ = ALTRO_MAX_PENALTY;
      return true;
    }
    if (max_outer) {
      status_al_ = ALTRO_MAX_OUTER_ITERATIONS;
      return true;
    }
    if (max_total) {
      status_al_ = ALTRO_MAX_ITERATIONS;
      return true;
    }
    return false;
  }
  // AugmentedLagrangianiLQR::Solve, al_solver.hpp:304-334
  void SolveAL() override {
    AlInit();
    for (int it = 0; it < opts.max_iterations_outer; ++it) {
      SolveILQR();
      UpdateDuals();
      // UpdateConvergenceStatistics, al_solver.hpp:357-366
      stats.iterations_outer++;
      stats.Log(F_VIOL, GetMaxViolation());
      stats.Log(F_PEN, GetMaxPenalty());
      if (AlIsDone()) break;
      UpdatePenalties();
    }
  }

  // ---- getters -------------------------------------------------------------------------------
  void GetTrajectory(double* Xo, double* Uo) override {
    if (Xo)
      for (int i = 0; i < (N + 1) * n; ++i) Xo[i] = (double)X[i];
    if (Uo)
      for (int i = 0; i < N * m; ++i) Uo[i] = (double)U[i];
  }
  void GetGains(double* Ko, double* dout) override {
    if (Ko)
      for (int i = 0; i < N * m * n; ++i) Ko[i] = (double)K[i];
    if (dout)
      for (int i = 0; i < N * m; ++i) dout[i] = (double)d[i];
  }
  void GetCtg(double* Po, double* po) override {
    if (Po)
      for (int i = 0; i < (N + 1) * n * n; ++i) Po[i] = (double)P[i];
    if (po)
      for (int i = 0; i < (N + 1) * n; ++i) po[i] = (double)p[i];
  }
  void GetExpansion(int k, double* ABo, double* a, double* b, double* c, double* gx, double* gu) override {
    if (ABo)
      for (int i = 0; i < n * nm; ++i) ABo[i] = (double)AB[k * n * nm + i];
    if (a)
      for (int i = 0; i < n * n; ++i) a[i] = (double)lxx[k * n * n + i];
    if (b)
      for (int i = 0; i < n * m; ++i) b[i] = (double)lxu[k * n * m + i];
    if (c)
      for (int i = 0; i < m * m; ++i) c[i] = (double)luu[k * m * m + i];
    if (gx)
      for (int i = 0; i < n; ++i) gx[i] = (double)lx[k * n + i];
    if (gu)
      for (int i = 0; i < m; ++i) gu[i] = (double)lu[k * m + i];
  }
  void GetKnotCosts(double* c) override {
    for (int k = 0; k <= N; ++k) c[k] = (double)costs[k];
  }
  int NumRowsAt(int k) override {
    int r = 0;
    for (Con& c : cons[k]) r += c.p;
    return r;
  }
  int NumRows() override {
    int r = 0;
    for (int k = 0; k <= N; ++k) r += NumRowsAt(k);
    return r;
  }
  template <class F>
  void ForRows(F f) {
    int r = 0;
    for (auto& v : cons)
      for (Con& c : v)
        for (int i = 0; i < c.p; ++i) f(r++, c, i);
  }
  void GetDuals(double* out) override {
    ForRows([&](int r, Con& c, int i) { out[r] = (double)c.lam[i]; });
  }
  void SetDuals(const double* in) override {
    ForRows([&](int r, Con& c, int i) { c.lam[i] = T(in[r]); });
  }
  void GetPenalties(double* out) override {
    ForRows([&](int r, Con& c, int i) { out[r] = (double)c.pen[i]; });
  }
  void GetConVals(double* out) override {
    ForRows([&](int r, Con& c, int i) { out[r] = (double)c.c[i]; });
  }
  void GetStats(altro_stats* s) override {
    s->status = status_al_;
    s->status_ilqr = status_;
    s->iterations_inner = stats.iterations_inner;
    s->iterations_outer = stats.iterations_outer;
    s->iterations_total = stats.iterations_total;
    s->reserved = 0;
    s->cost = stats.Back(F_COST);
    s->initial_cost = stats.initial_cost;
    s->cost_decrease = stats.Back(F_DJ);
    s->gradient = stats.Back(F_GRAD);
    s->violation = stats.Back(F_VIOL);
    s->max_penalty = stats.Back(F_PEN);
    s->alpha = stats.Back(F_ALPHA);
    s->regularization = stats.Back(F_REG);
    s->improvement_ratio = stats.Back(F_Z);
  }
};

}  // namespace

// ------------------------------------------------------------------------------------------------
// C API (mirror of include/altro_hip.h with the prefix oracle_)
// ------------------------------------------------------------------------------------------------
struct oracle_solver_s {
  altro_desc desc;
  int model_kind = 0;
  int dof = 0;
  float hstep = 0.0f;
  std::vector<float> hk, tk;  // per-knot steps [N] / times [N + 1] (Trajectory::SetStep / SetTime); empty: uniform
  std::vector<int> knot_model;  // per-knot model indices [N] (altro_set_knot_models); empty: model 0
  std::vector<CostSpec> costs;
  std::vector<ConSpec> cons;
  std::vector<double> x0;
  int x0_per_instance = 0;
  std::vector<double> X, U;
  int traj_per_instance = 0;
  bool has_X = false;
  altro_options opts;
  double penalty = -1.0, phi = -1.0;
  int nthreads = 1;
  bool built = false;
  bool ilqr_mode = false;
  std::vector<std::unique_ptr<SolverBase>> inst;
  std::vector<std::unique_ptr<SolverBase>> shadow;  // fp32-trial study only (ORACLE_F32_TRIALS*)
  std::string err;
  // CPU-baseline runs (oracle_bench_*): wall time between the start barrier of the thread team and its last task,
  // the threads that ran and the slowest / fastest thread's busy time
  double bench_seconds = 0.0, bench_busy_min = 0.0, bench_busy_max = 0.0;
  double bench_cpu_seconds = 0.0;  // CPU time the team's threads were actually given (CLOCK_THREAD_CPUTIME_ID, summed)
  int bench_threads = 0;
};
typedef oracle_solver_s* oracle_handle;

namespace {

// steps and times of the handle -> one instance (Trajectory::SetUniformStep, trajectory.hpp:122-130, or per knot)
void ApplyKnotTimes(oracle_handle h, std::vector<float>& hv, std::vector<float>& tv) {
  const int N = h->desc.N;
  for (int k = 0; k < N; ++k) hv[k] = h->hk.empty() ? h->hstep : h->hk[k];
  hv[N] = 0.0f;
  for (int k = 0; k <= N; ++k) {
    if (!h->tk.empty()) tv[k] = h->tk[k];
    else if (!h->hk.empty()) tv[k] = 0.0f;
    else tv[k] = k < N ? static_cast<float>(k) * h->hstep : h->hstep * N;
  }
}

template <class T, class Model>
std::unique_ptr<SolverBase> MakeInstance(oracle_handle h, int b, const Model& mdl) {
  const altro_desc& D = h->desc;
  auto up = std::make_unique<Instance<T, Model>>(D.N, mdl);
  Instance<T, Model>& I = *up;
  constexpr int n = Model::n, m = Model::m;
  ApplyKnotTimes(h, I.h, I.tm);
  for (int k = 0; k < D.N && !h->knot_model.empty(); ++k) I.km[k] = h->knot_model[k];
  for (const CostSpec& c : h->costs) {
    if (c.user) {
      int np = 0;
      UserCostShape(c.user - 1, &np);
      const double* par = c.params.data() + (c.per_instance ? (size_t)b * np : 0);
      for (int k = c.k_begin; k < c.k_end; ++k) I.SetUserCost(k, c.user - 1, par, np);
      continue;
    }
    const double* xr = c.xref.data() + ((c.per_instance & 1) ? (size_t)b * n : 0);
    const double* ur = c.uref.data() + ((c.per_instance & 2) ? (size_t)b * m : 0);
    for (int k = c.k_begin; k < c.k_end; ++k) I.SetLQRCost(k, c.Q.data(), c.R.data(), xr, ur);
  }
  for (const ConSpec& c : h->cons) {
    const double* par = c.params.data() + (c.per_instance ? (size_t)b * c.nparams : 0);
    for (int k = c.k_begin; k < c.k_end; ++k) I.AddConstraint(k, c.kind, par, c.nparams, c.user_type);
  }
  if (!h->x0.empty()) I.SetInitialState(h->x0.data() + (h->x0_per_instance ? (size_t)b * n : 0));
  const double* X = h->has_X ? h->X.data() + (h->traj_per_instance ? (size_t)b * (D.N + 1) * n : 0) : nullptr;
  const double* U = h->U.empty() ? nullptr : h->U.data() + (h->traj_per_instance ? (size_t)b * D.N * m : 0);
  I.SetTrajectory(X, U);
  I.opts = h->opts;
  return up;
}

template <class T>
std::unique_ptr<SolverBase> MakeForModel(oracle_handle h, int b) {
  const altro_desc& D = h->desc;
  if (h->model_kind == ALTRO_MODEL_UNICYCLE && D.n == 3 && D.m == 2)
    return MakeInstance<T>(h, b, UnicycleModel<T>());
  if (h->model_kind == ALTRO_MODEL_TRIPLE_INTEGRATOR && h->dof == 2 && D.n == 6 && D.m == 2)
    return MakeInstance<T>(h, b, TripleIntegratorModel<T, 2>());
  if (h->model_kind == ALTRO_MODEL_TRIPLE_INTEGRATOR && h->dof == 1 && D.n == 3 && D.m == 1)
    return MakeInstance<T>(h, b, TripleIntegratorModel<T, 1>());
  if (h->model_kind == ALTRO_MODEL_QUADROTOR12 && D.n == 12 && D.m == 4)
    return MakeInstance<T>(h, b, Quadrotor12Model<T>());
#ifdef ORACLE_USER_MODEL
  if (h->model_kind >= ALTRO_MODEL_USER_BASE && D.n == UserModelAdapter<T>::n && D.m == UserModelAdapter<T>::m)
    return MakeInstance<T>(h, b, UserModelAdapter<T>());
#endif
  return nullptr;
}

altro_status Build(oracle_handle h) {
  if (h->built) {
    for (auto& I : h->inst) I->opts = h->opts;
    for (auto& I : h->shadow) I->opts = h->opts;
    return ALTRO_OK;
  }
  h->inst.clear();
  h->shadow.clear();
  const bool f32_trials = h->desc.dtype == ORACLE_F32_TRIALS || h->desc.dtype == ORACLE_F32_TRIALS_RECHECK;
  for (int b = 0; b < h->desc.batch; ++b) {
    std::unique_ptr<SolverBase> I =
        h->desc.dtype == ALTRO_F32 ? MakeForModel<float>(h, b) : MakeForModel<double>(h, b);
    if (I && (h->desc.dtype == ORACLE_F64_F32REC || f32_trials)) I->SetRoundRecords(true);
    if (I && f32_trials) {  // the study: trials on an all-fp32 shadow of the instance
      std::unique_ptr<SolverBase> S = MakeForModel<float>(h, b);
      if (S) {
        S->opts = h->opts;
        I->SetShadow(S.get(), h->desc.dtype == ORACLE_F32_TRIALS_RECHECK);
        h->shadow.push_back(std::move(S));
      }
    }
    if (!I) {
      h->err = "unsupported (model, n, m) combination";
      return ALTRO_UNSUPPORTED;
    }
    if (h->penalty >= 0) I->SetPenalty(h->penalty);
    if (h->phi >= 0) I->SetPenaltyScaling(h->phi);
    h->inst.push_back(std::move(I));
  }
  h->built = true;
  return ALTRO_OK;
}

template <class F>
altro_status ForAll(oracle_handle h, F f) {
  if (!h) return ALTRO_INVALID_ARG;
  altro_status st = Build(h);
  if (st != ALTRO_OK) return st;
  const int B = h->desc.batch;
  const int nt = std::max(1, std::min(h->nthreads, B));
  if (nt == 1) {
    for (int b = 0; b < B; ++b) f(*h->inst[b], b);
  } else {
    std::atomic<int> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
      th.emplace_back([&]() {
        for (;;) {
          int b = next.fetch_add(1);
          if (b >= B) break;
          f(*h->inst[b], b);
        }
      });
    for (auto& t : th) t.join();
  }
  return ALTRO_OK;
}


// (cpu, rank among the hardware threads of its physical core) for every CPU of the affinity mask, ordered so that the
// first hardware thread of every physical core comes before any second one.
std::vector<std::pair<int, int>> HostTopology() {
  std::vector<std::pair<int, int>> out;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) != 0) {
    out.emplace_back(0, 0);
    return out;
  }
  std::map<std::pair<int, int>, int> seen;  // (package, core id) -> hardware threads met so far
  auto read_int = [](const std::string& path, int dflt) {
    int v = dflt;
    if (FILE* f = std::fopen(path.c_str(), "r")) {
      if (std::fscanf(f, "%d", &v) != 1) v = dflt;
      std::fclose(f);
    }
    return v;
  };
  for (int cpu = 0; cpu < CPU_SETSIZE; ++cpu) {
    if (!CPU_ISSET(cpu, &set)) continue;
    const std::string base = "/sys/devices/system/cpu/cpu" + std::to_string(cpu) + "/topology/";
    const int pkg = read_int(base + "physical_package_id", 0);
    const int core = read_int(base + "core_id", cpu);  // (no sysfs: every CPU counts as its own core)
    out.emplace_back(cpu, seen[{pkg, core}]++);
  }
  std::stable_sort(out.begin(), out.end(),
                   [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.second < b.second; });
  if (out.empty()) out.emplace_back(0, 0);
  return out;
}

template <class F>
altro_status BenchRun(oracle_handle h, int reps, F solve_one, int stride = 1) {
  if (!h || reps < 1 || stride < 1) return ALTRO_INVALID_ARG;
  altro_status st = Build(h);  // (a no-op after oracle_prepare)
  if (st != ALTRO_OK) return st;
  const altro_desc& D = h->desc;
  const int B = D.batch;
  const int nsub = (B + stride - 1) / stride;  // every stride-th instance
  const int nt = std::max(1, std::min(h->nthreads, nsub));
  const std::vector<std::pair<int, int>> topo = HostTopology();
  const long long tasks = (long long)reps * nsub;
  std::atomic<long long> next(0);
  std::atomic<int> arrived(0);
  std::unique_ptr<std::atomic<int>[]> busy(new std::atomic<int>[B]);
  for (int b = 0; b < B; ++b) busy[b].store(0);
  using clk = std::chrono::steady_clock;
  clk::time_point t_start;
  std::vector<double> t_end(nt, 0.0), t_busy(nt, 0.0), t_cpu(nt, 0.0);
  auto thread_cpu = []() {
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
  };
  auto body = [&](int t) {
    cpu_set_t one;
    CPU_ZERO(&one);
    CPU_SET(topo[t % topo.size()].first, &one);
    pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
    if (arrived.fetch_add(1) + 1 == nt) t_start = clk::now();  // the last thread to arrive starts the clock ...
    while (arrived.load() < nt) std::this_thread::yield();      // ... and releases the team
    const clk::time_point t0 = clk::now();
    const double c0 = thread_cpu();
    for (;;) {
      const long long i = next.fetch_add(1);
      if (i >= tasks) break;
      const int b = (int)(i % nsub) * stride;
      int idle = 0;
      while (!busy[b].compare_exchange_weak(idle, 1, std::memory_order_acquire)) {  // the previous repetition of b
        idle = 0;
        std::this_thread::yield();
      }
      const double* Xb = h->has_X ? h->X.data() + (h->traj_per_instance ? (size_t)b * (D.N + 1) * D.n : 0) : nullptr;
      const double* Ub = h->U.empty() ? nullptr : h->U.data() + (h->traj_per_instance ? (size_t)b * D.N * D.m : 0);
      SolverBase& s = *h->inst[b];
      s.SetTrajectory(Xb, Ub);
      solve_one(s);
      busy[b].store(0, std::memory_order_release);
    }
    const clk::time_point t1 = clk::now();
    t_cpu[t] = thread_cpu() - c0;
    t_busy[t] = std::chrono::duration<double>(t1 - t0).count();
    t_end[t] = std::chrono::duration<double>(t1.time_since_epoch()).count();
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t) th.emplace_back(body, t);
  body(0);
  for (auto& t : th) t.join();
  const double start = std::chrono::duration<double>(t_start.time_since_epoch()).count();
  h->bench_seconds = *std::max_element(t_end.begin(), t_end.end()) - start;
  h->bench_busy_max = *std::max_element(t_busy.begin(), t_busy.end());
  h->bench_busy_min = *std::min_element(t_busy.begin(), t_busy.end());
  h->bench_threads = nt;
  h->bench_cpu_seconds = 0.0;
  for (double c : t_cpu) h->bench_cpu_seconds += c;
  // (the calling thread was pinned for the run: give it its mask back)
  cpu_set_t all;
  CPU_ZERO(&all);
  for (const auto& c : topo) CPU_SET(c.first, &all);
  pthread_setaffinity_np(pthread_self(), sizeof(all), &all);
  return ALTRO_OK;
}

}  // namespace

extern "C" {

void oracle_default_options(altro_options* o) {  // altro/common/solver_options.hpp:23-56
  std::memset(o, 0, sizeof(*o));
  o->max_iterations_total = 300;
  o->max_iterations_outer = 30;
  o->max_iterations_inner = 100;
  o->cost_tolerance = 1e-4;
  o->gradient_tolerance = 1e-2;
  o->bp_reg_increase_factor = 1.6;
  o->bp_reg_enable = 1;
  o->bp_reg_initial = 0.0;
  o->bp_reg_max = 1e8;
  o->bp_reg_min = 1e-8;
  o->bp_reg_fail_threshold = 100;
  o->check_forwardpass_bounds = 1;
  o->state_max = 1e8;
  o->control_max = 1e8;
  o->line_search_max_iterations = 20;
  o->line_search_lower_bound = 1e-8;
  o->line_search_upper_bound = 10.0;
  o->line_search_decrease_factor = 2;
  o->constraint_tolerance = 1e-4;
  o->maximum_penalty = 1e8;
  o->initial_penalty = 1.0;
  o->reset_duals = 1;
  o->profiler_enable = 0;
}

altro_status oracle_create(const altro_desc* desc, oracle_handle* out) {
  if (!desc || !out || desc->n <= 0 || desc->m <= 0 || desc->N <= 0 || desc->batch <= 0)
    return ALTRO_INVALID_ARG;
  oracle_handle h = new oracle_solver_s();
  h->desc = *desc;
  oracle_default_options(&h->opts);
  *out = h;
  return ALTRO_OK;
}
void oracle_destroy(oracle_handle h) { delete h; }
altro_status oracle_get_desc(oracle_handle h, altro_desc* out) {
  if (!h || !out) return ALTRO_INVALID_ARG;
  *out = h->desc;
  return ALTRO_OK;
}
const char* oracle_last_error(oracle_handle h) { return h ? h->err.c_str() : ""; }
altro_status oracle_set_threads(oracle_handle h, int nthreads) {
  h->nthreads = nthreads;
  return ALTRO_OK;
}

altro_status oracle_set_model(oracle_handle h, int kind, const double* params, int nparams) {
  h->model_kind = kind;
  h->dof = (kind == ALTRO_MODEL_TRIPLE_INTEGRATOR && nparams > 0) ? (int)params[0] : 0;
  h->built = false;
  return ALTRO_OK;
}
altro_status oracle_set_uniform_step(oracle_handle h, float hstep) {
  h->hstep = hstep;
  h->hk.clear();
  h->tk.clear();
  h->built = false;
  return ALTRO_OK;
}
altro_status oracle_set_steps(oracle_handle h, const float* hk, int count) {  // Trajectory::SetStep(k, h)
  if (count != h->desc.N) return ALTRO_INVALID_ARG;
  if (h->tk.empty() && h->hk.empty() && h->hstep > 0.0f) {
    h->tk.resize(count + 1);
    for (int k = 0; k < count; ++k) h->tk[k] = static_cast<float>(k) * h->hstep;
    h->tk[count] = h->hstep * count;
  }
  h->hk.assign(hk, hk + count);
  h->built = false;
  return ALTRO_OK;
}
altro_status oracle_set_knot_models(oracle_handle h, const int* model_of_knot, int count) {  // Problem::SetDynamics(model, k)
  if (count != h->desc.N) return ALTRO_INVALID_ARG;
  h->knot_model.assign(model_of_knot, model_of_knot + count);
  h->built = false;
  return ALTRO_OK;
}
altro_status oracle_set_times(oracle_handle h, const float* tk, int count) {  // Trajectory::SetTime(k, t)
  if (count != h->desc.N + 1) return ALTRO_INVALID_ARG;
  h->tk.assign(tk, tk + count);
  h->built = false;
  return ALTRO_OK;
}
altro_status oracle_get_steps(oracle_handle h, float* hk, float* tk) {
  std::vector<float> hv(h->desc.N + 1), tv(h->desc.N + 1);
  ApplyKnotTimes(h, hv, tv);
  for (int k = 0; k < h->desc.N && hk; ++k) hk[k] = hv[k];
  for (int k = 0; k <= h->desc.N && tk; ++k) tk[k] = tv[k];
  return ALTRO_OK;
}
altro_status oracle_set_lqr_cost(oracle_handle h, int k_begin, int k_end, const double* Q,
                                 const double* R, const double* xref, const double* uref,
                                 int per_instance) {
  const int n = h->desc.n, m = h->desc.m, B = h->desc.batch;
  if (k_begin < 0 || k_end > h->desc.N + 1 || k_begin >= k_end) return ALTRO_INVALID_ARG;
  CostSpec c;
  c.k_begin = k_begin;
  c.k_end = k_end;
  c.per_instance = per_instance;
  c.Q.assign(Q, Q + n * n);
  c.R.assign(R, R + m * m);
  c.xref.assign(xref, xref + (size_t)n * ((per_instance & 1) ? B : 1));
  c.uref.assign(uref, uref + (size_t)m * ((per_instance & 2) ? B : 1));
  h->costs.push_back(std::move(c));
  h->built = false;
  return ALTRO_OK;
}
altro_status oracle_set_user_cost_type(oracle_handle h, int type, int k_begin, int k_end, const double* params, int nparams,
                                       int per_instance) {
  if (k_begin < 0 || k_end > h->desc.N + 1 || k_begin >= k_end) return ALTRO_INVALID_ARG;
  int np = 0;
  if (!UserCostShape(type, &np) || nparams != np) return ALTRO_INVALID_ARG;
  CostSpec c;
  c.k_begin = k_begin;
  c.k_end = k_end;
  c.per_instance = per_instance ? 1 : 0;
  c.user = 1 + type;
  if (nparams > 0) c.params.assign(params, params + (size_t)nparams * (per_instance ? h->desc.batch : 1));
  h->costs.push_back(std::move(c));
  h->built = false;
  return ALTRO_OK;
}
altro_status oracle_set_user_cost(oracle_handle h, int k_begin, int k_end, const double* params, int nparams,
                                  int per_instance) {
  return oracle_set_user_cost_type(h, 0, k_begin, k_end, params, nparams, per_instance);
}
static altro_status AddConstraintOfType(oracle_handle h, int kind, int user_type, int k_begin, int k_end,
                                        const double* params, int nparams, int per_instance) {
  if (k_begin < 0 || k_end > h->desc.N + 1 || k_begin >= k_end) return ALTRO_INVALID_ARG;
  ConSpec c;
  c.user_type = user_type;
  c.kind = kind;
  c.k_begin = k_begin;
  c.k_end = k_end;
  c.nparams = nparams;
  c.per_instance = per_instance;
  if (kind == ALTRO_CON_USER) {
    int np = 0, rows = 0;
    bool eq = false;
    if (!UserConShape(user_type, &np, &rows, &eq) || nparams != np) return ALTRO_INVALID_ARG;
  }
  if (nparams > 0) c.params.assign(params, params + (size_t)nparams * (per_instance ? h->desc.batch : 1));
  h->cons.push_back(std::move(c));
  h->built = false;
  return ALTRO_OK;
}
altro_status oracle_add_constraint(oracle_handle h, int kind, int k_begin, int k_end,
                                   const double* params, int nparams, int per_instance) {
  return AddConstraintOfType(h, kind, 0, k_begin, k_end, params, nparams, per_instance);
}
altro_status oracle_add_user_constraint_type(oracle_handle h, int type, int k_begin, int k_end, const double* params,
                                             int nparams, int per_instance) {
  return AddConstraintOfType(h, ALTRO_CON_USER, type, k_begin, k_end, params, nparams, per_instance);
}
altro_status oracle_set_initial_state(oracle_handle h, const double* x0, int per_instance) {
  const int n = h->desc.n;
  h->x0.assign(x0, x0 + (size_t)n * (per_instance ? h->desc.batch : 1));
  h->x0_per_instance = per_instance;
  if (h->built)
    for (int b = 0; b < h->desc.batch; ++b)
      h->inst[b]->SetInitialState(h->x0.data() + (per_instance ? (size_t)b * n : 0));
  if (h->built)
    for (size_t b = 0; b < h->shadow.size(); ++b)
      h->shadow[b]->SetInitialState(h->x0.data() + (per_instance ? b * n : 0));
  return ALTRO_OK;
}
altro_status oracle_set_trajectory(oracle_handle h, const double* X, const double* U, int per_instance) {
  const altro_desc& D = h->desc;
  const size_t mult = per_instance ? D.batch : 1;
  h->has_X = X != nullptr;
  if (X) h->X.assign(X, X + mult * (D.N + 1) * D.n);
  if (U)
    h->U.assign(U, U + mult * D.N * D.m);
  else
    h->U.clear();
  h->traj_per_instance = per_instance;
  if (h->built)
    for (int b = 0; b < D.batch; ++b) {
      const double* Xb = X ? h->X.data() + (per_instance ? (size_t)b * (D.N + 1) * D.n : 0) : nullptr;
      const double* Ub = U ? h->U.data() + (per_instance ? (size_t)b * D.N * D.m : 0) : nullptr;
      h->inst[b]->SetTrajectory(Xb, Ub);
    }
  return ALTRO_OK;
}
altro_status oracle_reset_trajectory(oracle_handle h) {
  const altro_desc& D = h->desc;
  if (Build(h) != ALTRO_OK) return ALTRO_UNSUPPORTED;
  for (int b = 0; b < D.batch; ++b) {
    const double* Xb = h->has_X ? h->X.data() + (h->traj_per_instance ? (size_t)b * (D.N + 1) * D.n : 0) : nullptr;
    const double* Ub = h->U.empty() ? nullptr : h->U.data() + (h->traj_per_instance ? (size_t)b * D.N * D.m : 0);
    h->inst[b]->SetTrajectory(Xb, Ub);
  }
  return ALTRO_OK;
}
altro_status oracle_reset_stats(oracle_handle h) {  // solver.GetStats().Reset(), solver_stats.cpp:31-45
  return ForAll(h, [&](SolverBase& s, int) { s.RawStats().Reset(); });
}
// host-memory counterpart of altro_pack_results_device (the gloo tests gather CPU tensors)
altro_status oracle_pack_results_device(oracle_handle h, void* dst) {
  double* out = static_cast<double*>(dst);
  return ForAll(h, [&](SolverBase& s, int b) {
    altro_stats st;
    s.GetStats(&st);
    out[4 * (size_t)b + 0] = st.cost;
    out[4 * (size_t)b + 1] = st.violation;
    out[4 * (size_t)b + 2] = (double)st.iterations_total;
    out[4 * (size_t)b + 3] = (double)(h->ilqr_mode ? st.status_ilqr : st.status);
  });
}
// host-memory counterpart of altro_pack_trajectory_device
altro_status oracle_pack_trajectory_device(oracle_handle h, void* Xd, void* Ud) {
  const altro_desc& D = h->desc;
  double* X = static_cast<double*>(Xd);
  double* U = static_cast<double*>(Ud);
  return ForAll(h, [&](SolverBase& s, int b) {  // (every instance writes its own rows: safe from several threads)
    s.GetTrajectory(X ? X + (size_t)b * (D.N + 1) * D.n : nullptr, U ? U + (size_t)b * D.N * D.m : nullptr);
  });
}
altro_status oracle_set_options(oracle_handle h, const altro_options* o) {
  h->opts = *o;
  for (auto& I : h->inst) I->opts = *o;
  for (auto& I : h->shadow) I->opts = *o;
  return ALTRO_OK;
}
altro_status oracle_get_options(oracle_handle h, altro_options* o) {
  *o = h->opts;
  return ALTRO_OK;
}
altro_status oracle_set_penalty(oracle_handle h, double rho) {
  h->penalty = rho;
  return ForAll(h, [&](SolverBase& s, int) { s.SetPenalty(rho); });
}
altro_status oracle_set_penalty_scaling(oracle_handle h, double phi) {
  h->phi = phi;
  return ForAll(h, [&](SolverBase& s, int) { s.SetPenaltyScaling(phi); });
}

altro_status oracle_solve_al(oracle_handle h) {
  h->ilqr_mode = false;
  return ForAll(h, [](SolverBase& s, int) { s.SolveAL(); });
}
altro_status oracle_solve_ilqr(oracle_handle h) {
  h->ilqr_mode = true;
  return ForAll(h, [](SolverBase& s, int) { s.SolveILQR(); });
}
// CPU-baseline helpers for bench.py (the `cpu_baseline` leg): `reps` x (re-install the initial guess, solve) per
// instance on a team of PINNED threads.  What the timed region holds is the solver work and nothing else:
//  * the instances are built (and every per-instance array allocated) by oracle_prepare(), outside;
//  * thread t is pinned to the t-th hardware thread of the process's affinity mask, one thread per physical core first
//    (HostTopology), so a team of `physical cores` threads never shares a core;
//  * the team starts at a barrier; the clock runs from that barrier to the end of the last task
//    (oracle_bench_seconds), thread creation and join are outside;
//  * a task is ONE solve of ONE instance, handed out repetition-major by an atomic counter: the tail of the run is
//    one straggler solve, not `reps` of them (a try-lock per instance keeps two repetitions of one instance apart).
altro_status oracle_prepare(oracle_handle h) {
  if (!h) return ALTRO_INVALID_ARG;
  return Build(h);
}
double oracle_bench_seconds(oracle_handle h) { return h ? h->bench_seconds : 0.0; }
int oracle_bench_threads(oracle_handle h) { return h ? h->bench_threads : 0; }
// CPU time the threads of the last run were given, summed: well below threads x wall means the team was descheduled
// (a CPU quota of the container, other tenants) -- the host figure is then bounded by that quota, not by the cores
double oracle_bench_cpu_seconds(oracle_handle h) { return h ? h->bench_cpu_seconds : 0.0; }
// busy time of the slowest (which = 1) / fastest (which = 0) thread of the last run: their ratio is the load balance
double oracle_bench_busy(oracle_handle h, int which) { return h ? (which ? h->bench_busy_max : h->bench_busy_min) : 0.0; }
// hardware threads this process may run on / distinct physical cores among them (sysfs topology)
int oracle_host_threads(void) { return (int)HostTopology().size(); }
int oracle_host_physical_cores(void) {
  int n = 0;
  for (const auto& c : HostTopology()) n += c.second == 0;
  return std::max(n, 1);
}
altro_status oracle_bench_al(oracle_handle h, int reps) {
  h->ilqr_mode = false;
  return BenchRun(h, reps, [](SolverBase& s) { s.SolveAL(); });
}
// every stride-th instance only (the single-thread leg: the same instance mix as the batch in 1/stride of the time);
// AL or bare iLQR solves according to the handle's last oracle_bench_* / oracle_solve_* call
altro_status oracle_bench_subset(oracle_handle h, int stride, int reps) {
  if (h->ilqr_mode)
    return BenchRun(h, reps, [](SolverBase& s) {
      s.RawStats().Reset();
      s.SolveILQR();
    }, stride);
  return BenchRun(h, reps, [](SolverBase& s) { s.SolveAL(); }, stride);
}
altro_status oracle_set_ilqr_mode(oracle_handle h, int on) {
  h->ilqr_mode = on != 0;
  return ALTRO_OK;
}
// the same for a bare iLQR solve (BASELINE configs[1]); the statistics are reset between the repetitions
altro_status oracle_bench_ilqr(oracle_handle h, int reps) {
  h->ilqr_mode = true;
  return BenchRun(h, reps, [](SolverBase& s) {
    s.RawStats().Reset();
    s.SolveILQR();
  });
}
altro_status oracle_al_init(oracle_handle h) { return ForAll(h, [](SolverBase& s, int) { s.AlInit(); }); }
altro_status oracle_solve_setup(oracle_handle h) { return ForAll(h, [](SolverBase& s, int) { s.SolveSetup(); }); }
altro_status oracle_rollout(oracle_handle h) { return ForAll(h, [](SolverBase& s, int) { s.Rollout(); }); }
altro_status oracle_cost(oracle_handle h, double* J) {
  return ForAll(h, [&](SolverBase& s, int b) {
    double v = s.Cost();
    if (J) J[b] = v;
  });
}
altro_status oracle_update_expansions(oracle_handle h) { return ForAll(h, [](SolverBase& s, int) { s.UpdateExpansions(); }); }
altro_status oracle_backward_pass(oracle_handle h) { return ForAll(h, [](SolverBase& s, int) { s.BackwardPass(); }); }
// fp32-trial study (ORACLE_F32_TRIALS*): fp32 trials evaluated, accepted trials re-evaluated in fp64, re-evaluations
// that failed the acceptance test (mode 4), summed over the instances
void oracle_study_counters(oracle_handle h, long long* out) {
  out[0] = out[1] = out[2] = 0;
  for (auto& I : h->inst) {
    out[0] += I->trials_f32;
    out[1] += I->reevaluations;
    out[2] += I->recheck_rejections;
  }
}
altro_status oracle_forward_pass(oracle_handle h) { return ForAll(h, [](SolverBase& s, int) { s.ForwardPass(); }); }
altro_status oracle_update_convergence_statistics(oracle_handle h) {
  return ForAll(h, [](SolverBase& s, int) { s.UpdateConvergenceStatistics(); });
}
altro_status oracle_update_duals(oracle_handle h) { return ForAll(h, [](SolverBase& s, int) { s.UpdateDuals(); }); }
altro_status oracle_update_penalties(oracle_handle h) { return ForAll(h, [](SolverBase& s, int) { s.UpdatePenalties(); }); }
altro_status oracle_get_max_violation(oracle_handle h, double* out) {
  return ForAll(h, [&](SolverBase& s, int b) { out[b] = s.GetMaxViolation(); });
}
altro_status oracle_max_violation(oracle_handle h, double* out) {
  return ForAll(h, [&](SolverBase& s, int b) { out[b] = s.MaxViolation(); });
}
altro_status oracle_get_max_penalty(oracle_handle h, double* out) {
  return ForAll(h, [&](SolverBase& s, int b) { out[b] = s.GetMaxPenalty(); });
}

altro_status oracle_get_trajectory(oracle_handle h, double* X, double* U) {
  const altro_desc& D = h->desc;
  return ForAll(h, [&](SolverBase& s, int b) {
    s.GetTrajectory(X ? X + (size_t)b * (D.N + 1) * D.n : nullptr, U ? U + (size_t)b * D.N * D.m : nullptr);
  });
}
altro_status oracle_get_gains(oracle_handle h, double* K, double* d) {
  const altro_desc& D = h->desc;
  return ForAll(h, [&](SolverBase& s, int b) {
    s.GetGains(K ? K + (size_t)b * D.N * D.m * D.n : nullptr, d ? d + (size_t)b * D.N * D.m : nullptr);
  });
}
altro_status oracle_set_record_ctg(oracle_handle, int) { return ALTRO_OK; }
altro_status oracle_get_ctg(oracle_handle h, double* P, double* p) {
  const altro_desc& D = h->desc;
  return ForAll(h, [&](SolverBase& s, int b) {
    s.GetCtg(P ? P + (size_t)b * (D.N + 1) * D.n * D.n : nullptr, p ? p + (size_t)b * (D.N + 1) * D.n : nullptr);
  });
}
altro_status oracle_get_expansion(oracle_handle h, int k, double* AB, double* lxx, double* lxu,
                                  double* luu, double* lx, double* lu) {
  const int n = h->desc.n, m = h->desc.m;
  return ForAll(h, [&](SolverBase& s, int b) {
    s.GetExpansion(k, AB ? AB + (size_t)b * n * (n + m) : nullptr, lxx ? lxx + (size_t)b * n * n : nullptr,
                   lxu ? lxu + (size_t)b * n * m : nullptr, luu ? luu + (size_t)b * m * m : nullptr,
                   lx ? lx + (size_t)b * n : nullptr, lu ? lu + (size_t)b * m : nullptr);
  });
}
altro_status oracle_get_knot_costs(oracle_handle h, double* costs) {
  const int N = h->desc.N;
  return ForAll(h, [&](SolverBase& s, int b) { s.GetKnotCosts(costs + (size_t)b * (N + 1)); });
}
int oracle_num_constraints(oracle_handle h) {
  if (Build(h) != ALTRO_OK) return -1;
  return h->inst[0]->NumRows();
}
int oracle_num_constraints_at(oracle_handle h, int k) {
  if (Build(h) != ALTRO_OK) return -1;
  return h->inst[0]->NumRowsAt(k);
}
altro_status oracle_get_duals(oracle_handle h, double* lam) {
  int rows = oracle_num_constraints(h);
  return ForAll(h, [&](SolverBase& s, int b) { s.GetDuals(lam + (size_t)b * rows); });
}
altro_status oracle_set_duals(oracle_handle h, const double* lam) {
  int rows = oracle_num_constraints(h);
  return ForAll(h, [&](SolverBase& s, int b) { s.SetDuals(lam + (size_t)b * rows); });
}
altro_status oracle_get_penalties(oracle_handle h, double* rho) {
  int rows = oracle_num_constraints(h);
  return ForAll(h, [&](SolverBase& s, int b) { s.GetPenalties(rho + (size_t)b * rows); });
}
altro_status oracle_get_constraint_values(oracle_handle h, double* c) {
  int rows = oracle_num_constraints(h);
  return ForAll(h, [&](SolverBase& s, int b) { s.GetConVals(c + (size_t)b * rows); });
}
altro_status oracle_get_stats(oracle_handle h, altro_stats* st) {
  const bool ilqr = h->ilqr_mode;
  return ForAll(h, [&](SolverBase& s, int b) {
    s.GetStats(&st[b]);
    if (ilqr) st[b].status = st[b].status_ilqr;
  });
}
altro_status oracle_set_record_history(oracle_handle, int) { return ALTRO_OK; }
int oracle_get_history(oracle_handle h, int instance, int field, double* out, int cap) {
  if (Build(h) != ALTRO_OK || instance < 0 || instance >= h->desc.batch || field < 0 || field >= F_COUNT)
    return -1;
  static const Field map[8] = {F_COST, F_ALPHA, F_Z, F_GRAD, F_DJ, F_REG, F_VIOL, F_PEN};
  const std::vector<double>& v = h->inst[instance]->RawStats().v[map[field]];
  int cnt = std::min<int>(cap, (int)v.size());
  for (int i = 0; i < cnt; ++i) out[i] = v[i];
  return cnt;
}

int oracle_get_history_all(oracle_handle h, int instance, double* out, int cap) {
  int cnt = 0;
  for (int f = 0; f < 8; ++f) {
    const int c = oracle_get_history(h, instance, f, out + (size_t)f * cap, cap);
    if (c < 0) return -1;
    cnt = c;
  }
  return cnt;
}

}  // extern "C"
