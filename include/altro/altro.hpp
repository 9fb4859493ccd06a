// altro/altro.hpp — header-only C++ facade over the C-ABI of libaltro_hip.so.
//
// Keeps the class and method names of the reference (optimusride/altro-cpp, AltroCpp v0.3.4) for
// the AL-iLQR hot path so that a caller of
//     altro::problem::Problem, altro::Trajectory<n,m>,
//     altro::augmented_lagrangian::AugmentedLagrangianiLQR<n,m>, altro::ilqr::iLQR<n,m>
// can switch to the MI355X solver by changing includes and the link line.
//
// Differences that cannot be avoided:
//  * No Eigen (it is not a dependency of this build): vectors and matrices cross this API as
//    std::vector<double> / raw pointers, matrices column-major like Eigen's default.
//  * The reference's plug-in points are host virtual functions (problem::DiscreteDynamics,
//    problem::CostFunction, constraints::Constraint<ConType>) that a GPU kernel cannot call.  They
//    are replaced by DESCRIPTOR types with the reference's class names (examples::Unicycle,
//    examples::QuadraticCost::LQRCost, examples::GoalConstraint, ...): a `kind` plus parameters.
//  * NEW: a batch dimension.  Problem::SetBatch(B) makes every per-instance quantity (initial state,
//    cost reference, goal, obstacle parameters, initial guess) accept B values; the single-instance
//    calls of the reference are the B = 1 case.
//
// Error convention, as in the reference: programming errors -> assertion (ALTRO_ASSERT aborts in
// debug builds, altro/utils/assert.hpp:6-10) / std::runtime_error; numerical failures ->
// SolverStatus.  Solver objects are non-copyable and not thread-safe (altro/ilqr/ilqr.hpp:56-71).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <limits>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../altro_hip.h"

namespace altro {

// altro/common/solver_stats.hpp:20-31
enum class SolverStatus {
  kSolved = 0,
  kUnsolved = 1,
  kStateLimit = 2,
  kControlLimit = 3,
  kCostIncrease = 4,
  kMaxIterations = 5,
  kMaxOuterIterations = 6,
  kMaxInnerIterations = 7,
  kMaxPenalty = 8,
  kBackwardPassRegularizationFailed = 9,
};

// altro/common/log_entry.hpp:27-34
enum class LogLevel {
  kSilent = 0,
  kOuter = 1,
  kOuterDebug = 2,
  kInner = 3,
  kInnerDebug = 4,
  kDebug = 5,
};

constexpr int kPickHardwareThreads = -1;  // altro/common/solver_options.hpp:13

// altro/common/solver_options.hpp:19-65: every field of the reference, same names and defaults.  What the device solver
// does with the host-side ones: `verbose` / `header_frequency` -- the kernels do not print; the rows the reference's logger
// would have printed are recorded on the device and printed AFTER the solve (iLQR::PrintLog; needs the history, which
// batches of up to ilqr::kHistoryBatchLimit instances record by default).  `profiler_enable` times the kernels (HIP
// events), `profiler_output_to_file` / `log_directory` / `profile_filename` say where the solver's destructor writes the
// summed tree (the reference's Timer prints at destruction, timer.cpp:10-14).  `nthreads` / `tasks_per_thread` are accepted
// and have no effect: every (instance, knot) pair is a GPU thread of its own (ilqr.hpp:183-214 has no counterpart).
struct SolverOptions {
  int max_iterations_total = 300;
  int max_iterations_outer = 30;
  int max_iterations_inner = 100;
  double cost_tolerance = 1e-4;
  double gradient_tolerance = 1e-2;
  double bp_reg_increase_factor = 1.6;
  bool bp_reg_enable = true;
  double bp_reg_initial = 0.0;
  double bp_reg_max = 1e8;
  double bp_reg_min = 1e-8;
  int bp_reg_fail_threshold = 100;
  bool check_forwardpass_bounds = true;
  double state_max = 1e8;
  double control_max = 1e8;
  int line_search_max_iterations = 20;
  double line_search_lower_bound = 1e-8;
  double line_search_upper_bound = 10.0;
  double line_search_decrease_factor = 2;
  double constraint_tolerance = 1e-4;
  double maximum_penalty = 1e8;
  double initial_penalty = 1.0;
  bool reset_duals = true;
  int header_frequency = 10;
  LogLevel verbose = LogLevel::kSilent;
  bool profiler_enable = false;
  bool profiler_output_to_file = false;
  std::string log_directory;
  std::string profile_filename = "profiler.out";
  int nthreads = 1;
  int tasks_per_thread = 1;

  int NumThreads() const {  // solver_options.hpp:59-64
    if (nthreads == kPickHardwareThreads) return static_cast<int>(std::thread::hardware_concurrency());
    return std::max(nthreads, 1);
  }

  altro_options ToC() const {
    altro_options o;
    altro_default_options(&o);
    o.max_iterations_total = max_iterations_total;
    o.max_iterations_outer = max_iterations_outer;
    o.max_iterations_inner = max_iterations_inner;
    o.cost_tolerance = cost_tolerance;
    o.gradient_tolerance = gradient_tolerance;
    o.bp_reg_increase_factor = bp_reg_increase_factor;
    o.bp_reg_enable = bp_reg_enable;
    o.bp_reg_initial = bp_reg_initial;
    o.bp_reg_max = bp_reg_max;
    o.bp_reg_min = bp_reg_min;
    o.bp_reg_fail_threshold = bp_reg_fail_threshold;
    o.check_forwardpass_bounds = check_forwardpass_bounds;
    o.state_max = state_max;
    o.control_max = control_max;
    o.line_search_max_iterations = line_search_max_iterations;
    o.line_search_lower_bound = line_search_lower_bound;
    o.line_search_upper_bound = line_search_upper_bound;
    o.line_search_decrease_factor = line_search_decrease_factor;
    o.constraint_tolerance = constraint_tolerance;
    o.maximum_penalty = maximum_penalty;
    o.initial_penalty = initial_penalty;
    o.reset_duals = reset_duals;
    o.profiler_enable = profiler_enable;
    return o;
  }
};

// altro/common/solver_stats.hpp:44-63.  The counters and the per-iteration vectors of ONE instance
// (instance 0 unless SelectInstance() was called).  The vectors hold one row per logged iteration plus the
// row opened by the last NewIteration -- the reference's layout (solver_stats.cpp:54-66), so alpha[0] is the
// first accepted step and .back() the latest value -- when the solver records the history (the default for
// batches of up to kHistoryBatchLimit instances; SetRecordHistory() otherwise); without recording they hold
// the latest row only.  AllInstances() gives the final per-instance records of the whole batch.
struct SolverStats {
  double initial_cost = 0.0;
  int iterations_inner = 0;
  int iterations_outer = 0;
  int iterations_total = 0;
  std::vector<double> cost, alpha, improvement_ratio, gradient, cost_decrease, regularization, violations,
      max_penalty;
  std::vector<altro_stats> instances;
  const std::vector<altro_stats>& AllInstances() const { return instances; }
  void Reset() {  // solver_stats.cpp:31-45 (host mirror; iLQR::ResetStats() also clears the device counters)
    initial_cost = 0.0;
    iterations_inner = iterations_outer = iterations_total = 0;
    for (auto* v : {&cost, &alpha, &improvement_ratio, &gradient, &cost_decrease, &regularization, &violations, &max_penalty})
      v->clear();
  }
};

namespace constraints {
// altro/constraints/constraint.hpp:134-141
struct ConstraintInfo {
  std::string label;
  int index = 0;
  std::vector<double> violation;  // c - Pi_K(c): c for equalities, max(c, 0) for inequalities
  std::string type;
  double MaxViolation() const {
    double v = 0.0;
    for (double x : violation) v = std::max(v, std::abs(x));
    return v;
  }
  std::string ToString(int precision = 4) const {  // constraint.cpp:12-15
    std::string out = label + " at index " + std::to_string(index) + ": [";
    char buf[64];
    for (size_t i = 0; i < violation.size(); ++i) {
      std::snprintf(buf, sizeof(buf), "%.*g", precision, violation[i]);
      out += (i ? ", " : "") + std::string(buf);
    }
    return out + "]";
  }
};

// altro/constraints/constraint.hpp:60-128: the two cones of the reference, as tags
struct Equality {};
struct NegativeOrthant {};
using Inequality = NegativeOrthant;

// What a constraint IS on this facade: a kind plus parameters (the device evaluates it; a kernel cannot call the
// host virtual functions of constraint.hpp:173-202).  The typed classes below and in `examples` derive from it.
struct ConstraintDesc {
  int kind = 0;
  std::vector<double> params;  // one instance's block, or batch blocks back to back
  int nparams = 0;             // length of one instance's block
  std::string label;
  int user_p = 0;              // USER: OutputDimension of the source's UserConstraint
  bool user_equality = false;  // USER: its cone (constraints::Equality / NegativeOrthant)
  int user_type = 0;           // USER: index of the class in the source's ALTRO_USER_CONSTRAINTS list
  bool operator==(const ConstraintDesc& o) const {
    return kind == o.kind && params == o.params && nparams == o.nparams && user_type == o.user_type;
  }
  bool IsEquality() const { return kind == ALTRO_CON_GOAL || (kind == ALTRO_CON_USER && user_equality); }
  // Constraint<ConType>::GetConstraintType, altro/constraints/constraint.hpp:193-201
  std::string GetConstraintType() const { return IsEquality() ? "Equality Constraint" : "Inequality Constraint"; }
  std::string GetLabel() const { return label.empty() ? GetConstraintType() : label; }  // constraint.hpp:191
  int OutputDimension() const {
    if (kind == ALTRO_CON_USER) return user_p;
    if (kind == ALTRO_CON_GOAL) return nparams;
    if (kind == ALTRO_CON_CIRCLE) return nparams / 3;
    int p = 0;  // CONTROL_BOUND: one row per finite bound (basic_constraints.hpp:138-145)
    for (int i = 0; i < nparams; ++i)
      if (std::abs(params[i]) < std::numeric_limits<double>::max()) ++p;
    return p;
  }
};

// altro/constraints/constraint.hpp:173-205: the typed base class a reference caller holds its constraints by
// (`std::shared_ptr<Constraint<Inequality>> obs = std::make_shared<examples::CircleConstraint>(...)`,
// `ConstraintPtr<Equality> goal = ...`).  Problem::SetConstraint takes the pointer and keeps the descriptor.
template <class ConType>
struct Constraint : ConstraintDesc {
  using ConstraintType = ConType;
  virtual ~Constraint() = default;
};
template <class ConType>
using ConstraintPtr = std::shared_ptr<Constraint<ConType>>;
}  // namespace constraints

namespace problem {
// altro/problem/costfunction.hpp:52-73: the base class a reference caller holds its cost functions by
// (`std::shared_ptr<CostFunction>`).  On this facade the only host-side subclasses are descriptors
// (examples::QuadraticCost, examples::UserCost); the caller's own cost travels as source (UserCost).
struct CostFunction {
  virtual ~CostFunction() = default;
};

// altro/problem/integration.hpp:87-169: the two explicit integrators of the reference, as tags.  RungeKutta4 is a
// template over the sizes there (`RungeKutta4<NStates, NControls>`), ExplicitEuler is not.
template <int NStates = -1, int NControls = -1>
struct RungeKutta4 {
  static constexpr int kind = 0;
};
struct ExplicitEuler {
  static constexpr int kind = 1;
};

// altro/problem/dynamics.hpp:148-187: what Problem::SetDynamics stores per knot.  Here it carries what the device needs
// to pick the compiled model: kind (built-in or the plugin of a user source), parameters, dimensions, index in the source.
struct DiscreteDynamics {
  virtual ~DiscreteDynamics() = default;
  virtual int Kind() const = 0;
  virtual int StateDimension() const = 0;
  virtual int ControlDimension() const = 0;
  virtual std::vector<double> Params() const = 0;
  virtual int ModelIndex() const = 0;
};

// problem/discretized_model.hpp:24-65.  RungeKutta4 (the default) for every model; ExplicitEuler for a user model whose
// plugin was compiled for it (examples::UserModel::Euler): the integrator is part of the compiled device code, not a
// run-time switch.
template <class Model, class Integrator = RungeKutta4<Model::NStates, Model::NControls>>
struct DiscretizedModel : DiscreteDynamics {
  static constexpr int NStates = Model::NStates;
  static constexpr int NControls = Model::NControls;
  explicit DiscretizedModel(const Model& m) : model(m) {
    if (Integrator::kind != m.Integrator())
      throw std::runtime_error(Integrator::kind == 1
                                   ? "DiscretizedModel<Model, ExplicitEuler>: the device code of this model integrates with "
                                     "RungeKutta4 (a user model is compiled for ExplicitEuler by examples::UserModel::Euler)"
                                   : "DiscretizedModel<Model, RungeKutta4>: this user model was compiled for ExplicitEuler "
                                     "(examples::UserModel::Euler): wrap it in DiscretizedModel<Model, ExplicitEuler>");
  }
  int Kind() const override { return model.Kind(); }
  int StateDimension() const override { return model.StateDimension(); }
  int ControlDimension() const override { return model.ControlDimension(); }
  std::vector<double> Params() const override { return model.Params(); }
  int ModelIndex() const override { return model.ModelIndex(); }
  Model model;
};
}  // namespace problem

namespace detail {
inline void Check(altro_handle h, altro_status st, const char* what) {
  if (st != ALTRO_OK) {
    const char* msg = altro_last_error(h);
    throw std::runtime_error(std::string(what) + " failed (" + std::to_string((int)st) + "): " + (msg ? msg : ""));
  }
}
// The device timing of a solve (or the sums over several) as a tree in the layout of the reference's Timer
// (altro/common/timer.cpp:24-94, profile_entry.cpp:36-66; sample: perf/profiler_unicycle.out), with the reference's
// section names.  "sweep_fused" is this build's persistent tail launch.
inline void PrintTimingTree(FILE* f, altro_timing t) {
  const double total = t.total_ms * 1e3;
  // A large batch runs its batched sweeps as several chains on streams of their own: their launches overlap, and the
  // summed launch durations of the three sweep kernels can exceed the wall time.  The tree shows wall-time shares:
  // the three sections share what the solve took outside "init" and the persistent launch.
  const double sweep_sum = t.expansions_ms + t.backward_pass_ms + t.forward_pass_ms;
  const double sweep_wall = std::min(sweep_sum, std::max(0.0, t.total_ms - t.init_ms - t.fused_ms));
  const double scale = sweep_sum > 0 ? sweep_wall / sweep_sum : 1.0;
  t.expansions_ms *= scale;
  t.backward_pass_ms *= scale;
  t.forward_pass_ms *= scale;
  const double ilqr = (t.expansions_ms + t.backward_pass_ms + t.forward_pass_ms + t.fused_ms) * 1e3;
  auto row = [&](int depth, const char* name, double us, double parent) {
    char label[64];
    std::snprintf(label, sizeof(label), "%*s%s", 2 * depth, "", name);
    std::fprintf(f, "%-28s %9.0f %8.0f %8.0f\n", label, us, total > 0 ? 100.0 * us / total : 0.0,
                 parent > 0 ? 100.0 * us / parent : 0.0);
  };
  std::fprintf(f, "Description                  Time (us)   %%Total  %%Parent\n");
  std::fprintf(f, "--------------------------------------------------------\n");
  row(0, "al", total, total);
  row(1, "ilqr", ilqr, total);
  row(3, "backward_pass", t.backward_pass_ms * 1e3, ilqr);
  row(3, "expansions", t.expansions_ms * 1e3, ilqr);
  row(3, "forward_pass", t.forward_pass_ms * 1e3, ilqr);
  row(3, "sweep_fused", t.fused_ms * 1e3, ilqr);
  row(1, "init", t.init_ms * 1e3, total);
  std::fprintf(f, "sweeps %d (tail iterations in the fused launch: %d), kernel launches %d (batched sweeps: %d), "
               "instance-iterations %lld\n",
               t.sweeps, t.fused_sweeps, t.launches, t.sweep_launches, t.instance_iterations);
}
}  // namespace detail

// altro/common/trajectory.hpp:24-160.  Host storage of the states, controls and the (float) step of
// `batch` instances; it is the initial guess AND the output of a solve (altro/ilqr/ilqr.hpp:223-235).
template <int n, int m>
class Trajectory {
 public:
  explicit Trajectory(int N, int batch = 1)
      : N_(N), B_(batch), X_((size_t)batch * (N + 1) * n, 0.0), U_((size_t)batch * N * m, 0.0), h_((size_t)N + 1, 0.0f),
        t_((size_t)N + 1, 0.0f) {}
  // the reference's run-time-size constructor (trajectory.hpp:37-38): Trajectory(n, m, N), one instance
  Trajectory(int n_rt, int m_rt, int N) : Trajectory(N, 1) {
    if (n_rt != n || m_rt != m) throw std::runtime_error("Trajectory(n, m, N): sizes disagree with the template arguments.");
  }
  // copyable and assignable like the reference's (trajectory.hpp:67-79): `*traj_ptr = prob_def.InitialTrajectory()`
  int NumSegments() const { return N_; }
  int BatchSize() const { return B_; }
  double* State(int k, int b = 0) { return &X_[((size_t)b * (N_ + 1) + k) * n]; }
  double* Control(int k, int b = 0) { return &U_[((size_t)b * N_ + k) * m]; }
  const double* State(int k, int b = 0) const { return &X_[((size_t)b * (N_ + 1) + k) * n]; }
  const double* Control(int k, int b = 0) const { return &U_[((size_t)b * N_ + k) * m]; }
  void SetUniformStep(float h) {                                  // trajectory.hpp:122-130
    for (int k = 0; k < N_; ++k) {
      h_[k] = h;
      t_[k] = static_cast<float>(k) * h;
    }
    h_[N_] = 0.0f;  // terminal knot has h = 0
    t_[N_] = h * N_;
    uniform_ = true;
  }
  void SetStep(int k, float h) {                                  // trajectory.hpp:120
    uniform_ = uniform_ && h_.at(k) == h;
    h_.at(k) = h;
  }
  void SetTime(int k, float t) {                                  // trajectory.hpp:119
    uniform_ = uniform_ && t_.at(k) == t;
    t_.at(k) = t;
  }
  float GetStep(int k) const { return h_.at(k); }
  // trajectory.hpp:138-153: t[k+1] - t[k] == h[k] for all k
  bool CheckTimeConsistency(const double eps = 1e-6, const bool verbose = false) const {
    for (int k = 0; k < N_; ++k) {
      const float h_calc = t_[k + 1] - t_[k], h_stored = h_[k];
      if (std::abs(h_stored - h_calc) > eps) {
        if (verbose) std::printf("k=%d\t h=%g\nt-=%g\t t+=%g\t dt=%g\n", k, h_stored, t_[k], t_[k + 1], h_calc);
        return false;
      }
    }
    return true;
  }
  float GetTime(int k) const { return t_.at(k); }
  // every step and time is what SetUniformStep wrote (the solver then runs its uniform-step kernels)
  bool IsUniformStep() const { return uniform_; }
  const std::vector<float>& Steps() const { return h_; }
  const std::vector<float>& Times() const { return t_; }
  void SetZero() {
    std::fill(X_.begin(), X_.end(), 0.0);
    std::fill(U_.begin(), U_.end(), 0.0);
  }
  std::vector<double>& States() { return X_; }
  std::vector<double>& Controls() { return U_; }

 private:
  int N_, B_;
  std::vector<double> X_, U_;
  std::vector<float> h_, t_;  // per knot, 32-bit floats like KnotPoint::t_, h_ (knotpoint.hpp:179-180)
  bool uniform_ = false;
};

// ---- descriptor types with the reference's class names --------------------------------------------
namespace examples {
struct Unicycle {  // examples/unicycle.hpp
  static constexpr int NStates = 3;
  static constexpr int NControls = 2;
  static constexpr int kind = ALTRO_MODEL_UNICYCLE;
  int Kind() const { return kind; }
  int StateDimension() const { return 3; }
  int ControlDimension() const { return 2; }
  std::vector<double> Params() const { return {}; }
  int ModelIndex() const { return 0; }
  int Integrator() const { return 0; }
};
struct TripleIntegrator {  // examples/triple_integrator.hpp
  static constexpr int NStates = -1;  // Eigen::Dynamic in the reference (triple_integrator.hpp): 3 * dof at run time
  static constexpr int NControls = -1;
  static constexpr int kind = ALTRO_MODEL_TRIPLE_INTEGRATOR;
  explicit TripleIntegrator(int dof = 1) : dof_(dof) {}
  int Kind() const { return kind; }
  int StateDimension() const { return 3 * dof_; }
  int ControlDimension() const { return dof_; }
  std::vector<double> Params() const { return {static_cast<double>(dof_)}; }
  int ModelIndex() const { return 0; }
  int Integrator() const { return 0; }
  int dof_;
};
struct Quadrotor12 {  // build-defined model of BASELINE config 5
  static constexpr int NStates = 12;
  static constexpr int NControls = 4;
  static constexpr int kind = ALTRO_MODEL_QUADROTOR12;
  int Kind() const { return kind; }
  int StateDimension() const { return 12; }
  int ControlDimension() const { return 4; }
  std::vector<double> Params() const { return {}; }
  int ModelIndex() const { return 0; }
  int Integrator() const { return 0; }
};

// The caller's own plug-in classes.  In the reference a user subclasses problem::ContinuousDynamics
// (altro/problem/dynamics.hpp:59-95), problem::CostFunction (costfunction.hpp:52-73) and
// constraints::Constraint<ConType> (constraint.hpp:173-202); a kernel cannot call host virtual functions, so here
// the three travel as SOURCE (include/altro_hip.h, altro_register_model_source): `struct UserModel { n, m, f, jac }`
// and, optionally, `struct UserCost` / `struct UserConstraint` announced by ALTRO_USER_COST / ALTRO_USER_CONSTRAINT --
// or several classes of each, listed by ALTRO_USER_COSTS / ALTRO_USER_CONSTRAINTS and picked by their index (`type`).
// The constructor compiles (or loads from the on-disk cache) the plugin and runs the device-side
// CheckJacobian / CheckGradient / CheckHessian; errors (compiler output included) are thrown.
//
// The source may also hold the other things Problem::SetDynamics accepts in the reference (include/altro_hip.h):
//   * SEVERAL models of one (n, m), listed by `#define ALTRO_USER_MODELS A, B` -- Model(i) is the descriptor of the i-th,
//     and Problem::SetDynamics(DiscretizedModel<UserModel>(model.Model(i)), k) puts it on knot k (problem.hpp:155-166);
//   * a model under problem::ExplicitEuler (integration.hpp:87-104): UserModel::Euler(...) compiles the source with
//     ALTRO_USER_INTEGRATOR = 1 (every model of the source that does not choose its own `integrator`), to be wrapped in
//     DiscretizedModel<UserModel, ExplicitEuler>;
//   * the caller's own problem::DiscreteDynamics (dynamics.hpp:148-187): a struct with `discrete = true`, step / step_jac.
struct UserModel {
  static constexpr int NStates = -1;  // sizes are the source's (run-time here, like Eigen::Dynamic)
  static constexpr int NControls = -1;
  UserModel(const std::string& name, const std::string& source, int n, int m, bool check_derivatives = true)
      : n_(n), m_(m) {
    Register(name, source, check_derivatives);
  }
  static UserModel Euler(const std::string& name, const std::string& source, int n, int m, bool check_derivatives = true) {
    UserModel um(n, m);
    um.integrator_ = 1;
    um.Register(name, "#define ALTRO_USER_INTEGRATOR 1\n" + source, check_derivatives);
    return um;
  }
  // the index-th model of the source's ALTRO_USER_MODELS list (same plugin, same handle: another knot's dynamics)
  UserModel Model(int index) const {
    UserModel um = *this;
    um.index_ = index;
    return um;
  }
  int Kind() const { return kind_; }
  int StateDimension() const { return n_; }
  int ControlDimension() const { return m_; }
  std::vector<double> Params() const { return {}; }
  int ModelIndex() const { return index_; }
  int Integrator() const { return integrator_; }

 private:
  UserModel(int n, int m) : n_(n), m_(m) {}
  void Register(const std::string& name, const std::string& source, bool check_derivatives) {
    const altro_status st = altro_register_model_source(name.c_str(), source.c_str(), check_derivatives ? 1 : 0, &kind_);
    if (st != ALTRO_OK) {
      const char* msg = altro_last_error(nullptr);
      throw std::runtime_error("altro_register_model_source failed (" + std::to_string((int)st) + "): " + (msg ? msg : ""));
    }
  }
  int kind_ = 0, n_, m_, index_ = 0, integrator_ = 0;
};

// examples/quadratic_cost.hpp:29-39.  xref may hold one reference or `batch` references.
struct QuadraticCost : problem::CostFunction {
  std::vector<double> Q, R, xref, uref;
  bool terminal = false;
  // the UserCost of the problem's UserModel instead (see UserCost below): its parameters, one block or `batch` blocks
  bool user = false;
  std::vector<double> user_params;
  int user_nparams = 0;
  int user_type = 0;  // index of the class in the source's ALTRO_USER_COSTS list
  static QuadraticCost LQRCost(const std::vector<double>& Q, const std::vector<double>& R,
                               const std::vector<double>& xref, const std::vector<double>& uref,
                               bool terminal = false) {
    QuadraticCost c;
    c.Q = Q;
    c.R = R;
    c.xref = xref;
    c.uref = uref;
    c.terminal = terminal;
    return c;
  }
  bool operator==(const QuadraticCost& o) const {
    return Q == o.Q && R == o.R && xref == o.xref && uref == o.uref && terminal == o.terminal && user == o.user &&
           user_params == o.user_params && user_nparams == o.user_nparams && user_type == o.user_type;
  }
};
// problem::CostFunction of the caller (costfunction.hpp:52-73): the `struct UserCost` of the UserModel's source with
// these parameters (UserCost::nparams doubles, or `batch` such blocks back to back); `type` picks the class when the
// source lists several (ALTRO_USER_COSTS), as each knot of a reference problem may hold another CostFunction subclass.
struct UserCost : QuadraticCost {
  explicit UserCost(const std::vector<double>& params, int nparams = -1, int type = 0) {
    user = true;
    user_params = params;
    user_nparams = nparams >= 0 ? nparams : static_cast<int>(params.size());
    user_type = type;
  }
};

using ConstraintDesc = constraints::ConstraintDesc;
// constraints::Constraint<ConType> of the caller (constraint.hpp:173-202): the `struct UserConstraint` of the
// UserModel's source with these parameters; p = its OutputDimension, equality = its cone; `type` picks the class when
// the source lists several (ALTRO_USER_CONSTRAINTS).
struct UserConstraint : ConstraintDesc {
  UserConstraint(const std::vector<double>& par, int p, bool equality = false, int npar = -1,
                 const std::string& name = "User Constraint", int type = 0) {
    kind = ALTRO_CON_USER;
    user_type = type;
    params = par;
    nparams = npar >= 0 ? npar : static_cast<int>(par.size());
    user_p = p;
    user_equality = equality;
    label = name;
  }
};
// examples/basic_constraints.hpp:15-40
struct GoalConstraint : constraints::Constraint<constraints::Equality> {
  explicit GoalConstraint(const std::vector<double>& xf, int n = -1) {
    kind = ALTRO_CON_GOAL;
    params = xf;
    nparams = n > 0 ? n : static_cast<int>(xf.size());
    label = "Goal Constraint";
  }
};
// examples/basic_constraints.hpp:42-151
struct ControlBound : constraints::Constraint<constraints::Inequality> {
  ControlBound(const std::vector<double>& lb, const std::vector<double>& ub) {
    if (lb.size() != ub.size() || lb.empty())
      throw std::runtime_error("Upper and lower bounds must have the same length.");
    kind = ALTRO_CON_CONTROL_BOUND;
    params = lb;
    params.insert(params.end(), ub.begin(), ub.end());
    nparams = static_cast<int>(params.size());
    label = "Control Bound";
  }
};
// examples/obstacle_constraints.hpp:69-127
struct CircleConstraint : constraints::Constraint<constraints::Inequality> {
  CircleConstraint() {
    kind = ALTRO_CON_CIRCLE;
    label = "Circle Constraint";
  }
  void AddObstacle(double px, double py, double radius) {
    params.push_back(px);
    params.push_back(py);
    params.push_back(radius);
    nparams = static_cast<int>(params.size());
  }
  // per-instance obstacles: `all` holds batch blocks of (cx, cy, r) triples
  void SetBatchObstacles(const std::vector<double>& all, int per_instance_len) {
    params = all;
    nparams = per_instance_len;
  }
};
}  // namespace examples

namespace problem {
// altro/problem/problem.hpp:65-307
class Problem {
 public:
  explicit Problem(int N)
      : N_(N), costs_(N + 1), has_cost_(N + 1, false), cons_(N + 1), has_dyn_(N + 1, false), knot_model_(N + 1, 0) {}
  int NumSegments() const { return N_; }
  void SetBatch(int B) { batch_ = B; }
  int BatchSize() const { return batch_; }

  void SetInitialState(const std::vector<double>& x0) { x0_ = x0; }  // [n] or [batch][n]
  const std::vector<double>& GetInitialState() const { return x0_; }

  void SetCostFunction(const examples::QuadraticCost& cost, int k) {
    Range(k);
    costs_[k] = cost;
    has_cost_[k] = true;
  }
  // the reference's signature (problem.hpp:113-116): the cost function by shared pointer to its base class.  The
  // object must be one of the facade's descriptors (examples::QuadraticCost, examples::UserCost); its values are
  // copied, so later edits through the pointer do not reach the problem.
  void SetCostFunction(std::shared_ptr<CostFunction> costfun, int k) {
    if (!costfun) throw std::runtime_error("Cannot pass a nullptr for the cost function.");
    const auto* q = dynamic_cast<const examples::QuadraticCost*>(costfun.get());
    if (!q)
      throw std::runtime_error("Problem::SetCostFunction: a host-side CostFunction subclass cannot run on the device; hand "
                               "the cost over as source (examples::UserCost, altro_register_model_source)");
    SetCostFunction(*q, k);
  }
  // problem.hpp:133-139: an interval of consecutive knots
  template <class CostFun>
  void SetCostFunction(const std::vector<std::shared_ptr<CostFun>>& costfuns, int k_start = 0) {
    for (size_t i = 0; i < costfuns.size(); ++i) SetCostFunction(std::shared_ptr<CostFunction>(costfuns[i]), (int)i + k_start);
  }

  void SetDynamics(const DiscreteDynamics& dm, int k) {
    Range(k);
    if (k >= N_) throw std::runtime_error("dynamics are set on knots 0..N-1");
    // The reference keeps one model PER KNOT (models_[k], problem.hpp:155-166).  A handle of this build carries one
    // compiled model source for the whole horizon; knots may use DIFFERENT models of that source (a user source that
    // lists several: ALTRO_USER_MODELS, examples::UserModel::Model(i)) -- the knot's index travels through
    // altro_set_knot_models.  A model of another kind / source on another knot would silently solve the wrong dynamics, so
    // it is refused (the state and control dimensions could not change along the horizon either).
    bool any = false;
    for (int j = 0; j < N_; ++j) any = any || has_dyn_[j];
    if (any && (model_kind_ != dm.Kind() || model_params_ != dm.Params()))
      throw std::runtime_error("Problem::SetDynamics: a model of another kind on knot " + std::to_string(k) +
                               " -- the models of one problem come from ONE source (a user source may list several: "
                               "#define ALTRO_USER_MODELS A, B and examples::UserModel::Model(i))");
    model_kind_ = dm.Kind();
    model_params_ = dm.Params();
    n_ = dm.StateDimension();
    m_ = dm.ControlDimension();
    has_dyn_[k] = true;
    knot_model_[k] = dm.ModelIndex();
    if (k == N_ - 1) has_dyn_[N_] = true;  // IdentityDynamics at the terminal knot (problem.hpp:161-164)
  }
  // the reference's signature (problem.hpp:155-166)
  void SetDynamics(std::shared_ptr<DiscreteDynamics> model, int k) {
    if (!model) throw std::runtime_error("Cannot pass a nullptr for the dynamics.");
    SetDynamics(*model, k);
  }
  // problem.hpp:187-193: an interval of consecutive knots, by pointer ...
  template <class Dynamics>
  void SetDynamics(const std::vector<std::shared_ptr<Dynamics>>& models, int k_start = 0) {
    for (size_t i = 0; i < models.size(); ++i) SetDynamics(std::shared_ptr<DiscreteDynamics>(models[i]), (int)i + k_start);
  }
  // ... or by value: models[k] on knot k, k = 0 .. N-1
  template <class Model, class Integrator>
  void SetDynamics(const std::vector<DiscretizedModel<Model, Integrator>>& models) {
    if ((int)models.size() != N_) throw std::runtime_error("Problem::SetDynamics: expected N models");
    for (int k = 0; k < N_; ++k) SetDynamics(models[k], k);
  }
  void SetConstraint(const constraints::ConstraintDesc& con, int k) {
    Range(k);
    cons_[k].push_back(con);
  }
  // the reference's signatures (problem.hpp:195-202): a pointer to the constraint class itself or to its typed base
  // (constraints::ConstraintPtr<ConType>); the descriptor is copied
  template <class ConstraintObject>
  void SetConstraint(std::shared_ptr<ConstraintObject> con, int k) {
    if (!con) throw std::runtime_error("Cannot pass a nullptr for the constraint.");
    SetConstraint(static_cast<const constraints::ConstraintDesc&>(*con), k);
  }
  // the constraints of knot k in the order the solver stacks them: equalities first, then inequalities,
  // insertion order within each (al_cost.hpp:267-272)
  std::vector<examples::ConstraintDesc> Constraints(int k) const {
    std::vector<examples::ConstraintDesc> v = cons_[k];
    std::stable_partition(v.begin(), v.end(), [](const examples::ConstraintDesc& c) { return c.IsEquality(); });
    return v;
  }
  int NumConstraints(int k) const {
    int p = 0;
    for (const auto& c : cons_[k]) p += c.OutputDimension();
    return p;
  }
  int NumConstraints() const {
    int p = 0;
    for (int k = 0; k <= N_; ++k) p += NumConstraints(k);
    return p;
  }
  // problem.hpp:213-240: what a knot was given, as the reference's base-class pointers (copies of the stored descriptors:
  // problem_test.cpp:32-60, ilqr_class_test.cpp:39-69).  An unset cost function is a nullptr, unset dynamics an error.
  std::shared_ptr<CostFunction> GetCostFunction(int k) const {
    Range(k);
    if (!has_cost_[k]) return nullptr;
    return std::make_shared<examples::QuadraticCost>(costs_[k]);
  }
  std::shared_ptr<DiscreteDynamics> GetDynamics(int k) const {
    Range(k);
    if (!has_dyn_[k]) throw std::runtime_error("Dynamics have not been defined at this knot point.");
    struct Stored final : DiscreteDynamics {
      int kind, n, m, index;
      std::vector<double> params;
      int Kind() const override { return kind; }
      int StateDimension() const override { return n; }
      int ControlDimension() const override { return m; }
      std::vector<double> Params() const override { return params; }
      int ModelIndex() const override { return index; }
    };
    auto d = std::make_shared<Stored>();
    d->kind = model_kind_;
    d->n = n_;
    d->m = m_;
    d->index = k < N_ ? knot_model_[k] : 0;
    d->params = model_params_;
    return d;
  }
  std::shared_ptr<std::vector<double>> GetInitialStatePointer() const {  // problem.hpp:242 (a copy)
    return std::make_shared<std::vector<double>>(x0_);
  }
  bool IsFullyDefined() const {  // problem.hpp:271-297
    for (int k = 0; k <= N_; ++k)
      if (!has_cost_[k] || !has_dyn_[k]) return false;
    return !x0_.empty();
  }

  // BuildAugLagProblem (al_problem.hpp:29-51) marks the copy whose constraints a solver folds into the cost;
  // a plain iLQR built from an unmarked problem ignores the constraints (ilqr.hpp:117-119, quirk Q10).
  void MarkAugLag(bool on) { auglag_ = on; }
  bool IsAugLag() const { return auglag_; }

  // Replay the definition through the C-ABI (consecutive knots with identical descriptors become
  // one [k_begin, k_end) call).
  void Apply(altro_handle h, bool with_constraints = true) const {
    using detail::Check;
    Check(h, altro_set_model(h, model_kind_, model_params_.empty() ? nullptr : model_params_.data(),
                             (int)model_params_.size()), "altro_set_model");
    bool per_knot_models = false;
    for (int k = 0; k < N_; ++k) per_knot_models = per_knot_models || knot_model_[k] != 0;
    if (per_knot_models) Check(h, altro_set_knot_models(h, knot_model_.data(), N_), "altro_set_knot_models");
    for (int k = 0; k <= N_;) {
      int e = k + 1;
      while (e <= N_ && costs_[e] == costs_[k]) ++e;
      const auto& c = costs_[k];
      if (c.user) {
        Check(h, altro_set_user_cost_type(h, c.user_type, k, e, c.user_params.data(), c.user_nparams,
                                          (int)c.user_params.size() > c.user_nparams ? 1 : 0), "altro_set_user_cost_type");
      } else {
        const int per = ((int)c.xref.size() > n_ ? 1 : 0) | ((int)c.uref.size() > m_ ? 2 : 0);
        Check(h, altro_set_lqr_cost(h, k, e, c.Q.data(), c.R.data(), c.xref.data(), c.uref.data(), per),
              "altro_set_lqr_cost");
      }
      k = e;
    }
    // insertion order within a knot is what matters (al_cost.hpp:267-272): emit constraint j of each knot
    size_t maxc = 0;
    if (with_constraints)
      for (const auto& v : cons_) maxc = std::max(maxc, v.size());
    for (size_t j = 0; j < maxc; ++j)
      for (int k = 0; k <= N_;) {
        if (cons_[k].size() <= j) {
          ++k;
          continue;
        }
        int e = k + 1;
        while (e <= N_ && cons_[e].size() > j && cons_[e][j] == cons_[k][j]) ++e;
        const auto& c = cons_[k][j];
        const int per = (int)c.params.size() > c.nparams ? 1 : 0;
        if (c.kind == ALTRO_CON_USER)
          Check(h, altro_add_user_constraint_type(h, c.user_type, k, e, c.params.data(), c.nparams, per),
                "altro_add_user_constraint_type");
        else
          Check(h, altro_add_constraint(h, c.kind, k, e, c.params.data(), c.nparams, per), "altro_add_constraint");
        k = e;
      }
    Check(h, altro_set_initial_state(h, x0_.data(), (int)x0_.size() > n_ ? 1 : 0), "altro_set_initial_state");
  }
  int StateDimension() const { return n_; }
  int ControlDimension() const { return m_; }

 private:
  void Range(int k) const {
    if (k < 0 || k > N_) throw std::runtime_error("Invalid knot point index.");
  }
  int N_, batch_ = 1, n_ = 0, m_ = 0, model_kind_ = 0;
  bool auglag_ = false;
  std::vector<double> model_params_, x0_;
  std::vector<examples::QuadraticCost> costs_;
  std::vector<bool> has_cost_;
  std::vector<std::vector<examples::ConstraintDesc>> cons_;
  std::vector<bool> has_dyn_;
  std::vector<int> knot_model_;  // per knot: index of its model in the user source's ALTRO_USER_MODELS list
};
}  // namespace problem

namespace augmented_lagrangian {
// BuildAugLagProblem (altro/augmented_lagrangian/al_problem.hpp:29-51): the copy of the problem whose
// constraints are folded into ALCost functions.  On this facade the device evaluates the AL terms itself, so
// the copy only carries the mark that tells a solver to register the constraints.
template <int n, int m>
problem::Problem BuildAugLagProblem(const problem::Problem& prob) {
  problem::Problem al = prob;
  al.MarkAugLag(true);
  return al;
}
}  // namespace augmented_lagrangian

namespace ilqr {

constexpr int kHistoryBatchLimit = 64;   // batches up to this size record the per-iteration history by default
constexpr int kHistoryCapacity = 302;    // rows kept per instance at least (default max_iterations_total + the initial row)

namespace detail_ilqr {
// State shared by an iLQR solver, the AL solver built on it and the knot-point views: the C-ABI handle, the
// options / statistics objects of the reference, the caller's trajectory, and a cache of downloaded arrays
// that is invalidated by every compute call (`epoch`).
template <int n, int m>
struct Core {
  altro_handle h = nullptr;
  int N = 0, B = 1;
  SolverOptions opts;
  SolverStats stats;
  std::shared_ptr<Trajectory<n, m>> traj;
  bool pushed = false;
  bool record_history = false;
  int hist_cap = 0;  // rows of the device-side history per instance
  int stats_instance = 0;
  unsigned epoch = 0;
  std::vector<std::vector<examples::ConstraintDesc>> cons;  // per knot, solver order (empty without AL)
  // cache
  unsigned gains_epoch = ~0u, ctg_epoch = ~0u;
  std::vector<double> K, d, P, p;
  // sums of the solves' device timings while SolverOptions::profiler_enable is set; written at destruction unless
  // PrintTimings() was called (the reference's Timer: timer.cpp:10-14, solver_stats.cpp:60-77)
  altro_timing prof{};
  int prof_solves = 0;
  bool prof_printed = false;
  ~Core() {
    if (h && prof_solves > 0 && opts.profiler_enable && !prof_printed) {
      FILE* f = stdout;
      if (opts.profiler_output_to_file) {
        const std::string path = (opts.log_directory.empty() ? std::string() : opts.log_directory + "/") + opts.profile_filename;
        f = std::fopen(path.c_str(), "w");
      }
      if (f) {
        detail::PrintTimingTree(f, prof);
        if (f != stdout) std::fclose(f);
      }
    }
    if (h) altro_destroy(h);
  }
  Core() = default;
  Core(const Core&) = delete;
  Core& operator=(const Core&) = delete;
};

// CostExpansion (altro/ilqr/cost_expansion.hpp:24-116): the blocks of the quadratic cost expansion of one knot
struct CostExpansion {
  std::vector<double> xx, xu, uu, x, u;  // column-major n x n, n x m, m x m; n; m
  const std::vector<double>& dxdx() const { return xx; }
  const std::vector<double>& dxdu() const { return xu; }
  const std::vector<double>& dudu() const { return uu; }
  const std::vector<double>& dx() const { return x; }
  const std::vector<double>& du() const { return u; }
};
}  // namespace detail_ilqr

// altro/ilqr/knot_point_function_type.hpp:243-268: read-only view of one knot of one instance.  The gain and
// cost-to-go arrays are downloaded once per compute call and shared by all views (a loop over knots costs one
// device copy, not one per knot).
template <int n, int m>
class KnotPointFunctions {
  using Core = detail_ilqr::Core<n, m>;

 public:
  KnotPointFunctions(std::shared_ptr<Core> c, int k, int b) : c_(std::move(c)), k_(k), b_(b) {}
  std::vector<double> GetFeedbackGain() const {  // m x n, column-major
    Gains();
    return Slice(c_->K, ((size_t)b_ * c_->N + k_) * m * n, m * n);
  }
  std::vector<double> GetFeedforwardGain() const {
    Gains();
    return Slice(c_->d, ((size_t)b_ * c_->N + k_) * m, m);
  }
  std::vector<double> GetCostToGoHessian() const {
    Ctg();
    return Slice(c_->P, ((size_t)b_ * (c_->N + 1) + k_) * n * n, n * n);
  }
  std::vector<double> GetCostToGoGradient() const {
    Ctg();
    return Slice(c_->p, ((size_t)b_ * (c_->N + 1) + k_) * n, n);
  }
  std::vector<double> GetDynamicsExpansion() const {  // [A|B], n x (n+m), column-major
    std::vector<double> AB((size_t)c_->B * n * (n + m));
    detail::Check(c_->h, altro_get_expansion(c_->h, k_, AB.data(), nullptr, nullptr, nullptr, nullptr, nullptr),
                  "altro_get_expansion");
    return Slice(AB, (size_t)b_ * n * (n + m), n * (n + m));
  }
  detail_ilqr::CostExpansion GetCostExpansion() const {  // knot_point_function_type.hpp:249
    const size_t B = c_->B;
    std::vector<double> xx(B * n * n), xu(B * n * m), uu(B * m * m), x(B * n), u(B * m);
    const bool stage = k_ < c_->N;
    detail::Check(c_->h, altro_get_expansion(c_->h, k_, nullptr, xx.data(), stage ? xu.data() : nullptr,
                                             stage ? uu.data() : nullptr, x.data(), stage ? u.data() : nullptr),
                  "altro_get_expansion");
    detail_ilqr::CostExpansion e;
    e.xx = Slice(xx, (size_t)b_ * n * n, n * n);
    e.xu = stage ? Slice(xu, (size_t)b_ * n * m, n * m) : std::vector<double>((size_t)n * m, 0.0);
    e.uu = stage ? Slice(uu, (size_t)b_ * m * m, m * m) : std::vector<double>((size_t)m * m, 0.0);
    e.x = Slice(x, (size_t)b_ * n, n);
    e.u = stage ? Slice(u, (size_t)b_ * m, m) : std::vector<double>((size_t)m, 0.0);
    return e;
  }

 private:
  void Gains() const {
    if (c_->gains_epoch == c_->epoch) return;
    c_->K.resize((size_t)c_->B * c_->N * m * n);
    c_->d.resize((size_t)c_->B * c_->N * m);
    detail::Check(c_->h, altro_get_gains(c_->h, c_->K.data(), c_->d.data()), "altro_get_gains");
    c_->gains_epoch = c_->epoch;
  }
  void Ctg() const {
    if (c_->ctg_epoch == c_->epoch) return;
    c_->P.resize((size_t)c_->B * (c_->N + 1) * n * n);
    c_->p.resize((size_t)c_->B * (c_->N + 1) * n);
    // (behind a Solve() that did not record them -- the default: the persistent kernel keeps P, p in registers -- the library
    //  runs the last iteration's backward pass once more with the records on: altro_get_ctg, Engine::ReplayCtg)
    detail::Check(c_->h, altro_get_ctg(c_->h, c_->P.data(), c_->p.data()), "altro_get_ctg");
    c_->ctg_epoch = c_->epoch;
  }
  static std::vector<double> Slice(const std::vector<double>& v, size_t off, size_t len) {
    return std::vector<double>(v.begin() + off, v.begin() + off + len);
  }
  std::shared_ptr<Core> c_;
  int k_, b_;
};

// altro/ilqr/ilqr.hpp:47-813 (the algorithm methods; thread-pool and logging members omitted)
template <int n, int m>
class iLQR {
  using Core = detail_ilqr::Core<n, m>;

 public:
  explicit iLQR(int N) : c_(std::make_shared<Core>()) { c_->N = N; }  // ilqr.hpp:50
  explicit iLQR(const problem::Problem& prob, int dtype = ALTRO_F64, int device_id = 0)  // ilqr.hpp:51-54
      : c_(std::make_shared<Core>()) {
    c_->N = prob.NumSegments();
    InitializeFromProblem(prob, dtype, device_id);
  }
  iLQR(const iLQR&) = delete;
  iLQR& operator=(const iLQR&) = delete;
  iLQR(iLQR&&) noexcept = default;

  // ilqr.hpp:98-134.  A problem marked by BuildAugLagProblem brings its constraints (AL cost); a plain problem
  // contributes costs and dynamics only, like the reference (ilqr.hpp:117-119).
  void InitializeFromProblem(const problem::Problem& prob, int dtype = ALTRO_F64, int device_id = 0) {
    if (prob.NumSegments() != c_->N) throw std::runtime_error("Number of segments in problem isn't consistent with solver.");
    if (prob.StateDimension() != n || prob.ControlDimension() != m)
      throw std::runtime_error("Inconsistent state / control dimension.");
    if (!prob.IsFullyDefined()) throw std::runtime_error("Expected problem to be fully defined.");
    if (c_->h) throw std::runtime_error("The solver has already been initialized from a problem.");
    c_->B = prob.BatchSize();
    altro_desc d{n, m, c_->N, c_->B, dtype, device_id};
    altro_status st = altro_create(&d, &c_->h);
    if (st != ALTRO_OK) throw std::runtime_error(std::string("altro_create failed: ") + altro_last_error(nullptr));
    prob.Apply(c_->h, prob.IsAugLag());
    c_->cons.assign(c_->N + 1, {});
    if (prob.IsAugLag())
      for (int k = 0; k <= c_->N; ++k) c_->cons[k] = prob.Constraints(k);
    SetRecordHistory(c_->B <= kHistoryBatchLimit);
    ctg_auto_ = c_->B <= kHistoryBatchLimit;  // (see SetRecordCostToGo)
  }
  // ilqr.hpp:97-124: the reference copies knot points [k_start, k_stop) and is always called with the whole range
  // (ilqr.hpp:131, al_solver.hpp:246, ilqr_class_test.cpp:76); the device engine is built from a whole problem
  template <int n2 = n, int m2 = m>
  void CopyFromProblem(const problem::Problem& prob, int k_start, int k_stop, int dtype = ALTRO_F64, int device_id = 0) {
    if (k_start < 0 || k_start > c_->N) throw std::runtime_error("Start index must be in the interval [0,N]");
    if (k_stop < 0 || k_stop > c_->N + 1) throw std::runtime_error("Stop index must be in the interval [0,N+1]");
    if (k_start != 0 || k_stop != c_->N + 1)
      throw std::runtime_error("CopyFromProblem: the device engine takes a problem whole -- CopyFromProblem(prob, 0, N + 1)");
    InitializeFromProblem(prob, dtype, device_id);
  }
  bool IsInitialized() const { return c_->h != nullptr; }
  // ilqr.hpp:163: the per-knot costs of the last cost evaluation (of the selected instance), k = 0 .. N
  std::vector<double> GetCosts() {
    std::vector<double> all((size_t)c_->B * (c_->N + 1));
    detail::Check(Need(), altro_get_knot_costs(c_->h, all.data()), "altro_get_knot_costs");
    const size_t at = (size_t)c_->stats_instance * (c_->N + 1);
    return std::vector<double>(all.begin() + at, all.begin() + at + c_->N + 1);
  }

  int NumSegments() const { return c_->N; }
  int BatchSize() const { return c_->B; }
  SolverOptions& GetOptions() { return c_->opts; }
  SolverStats& GetStats() { return c_->stats; }
  SolverStatus GetStatus() const { return status_; }
  double GetRegularization() const { return c_->stats.instances.empty() ? 0.0 : c_->stats.instances[c_->stats_instance].regularization; }
  altro_handle Handle() const { return Need(); }
  // the instance whose counters / vectors GetStats() shows (default 0); takes effect at the next compute call
  void SelectInstance(int b) { c_->stats_instance = b; }
  // per-iteration vectors of SolverStats (solver_stats.hpp:56-63) need the device to log every iteration
  void SetRecordHistory(bool on) {
    c_->hist_cap = on ? HistoryRowsNeeded() : 0;
    detail::Check(Need(), altro_set_record_history(c_->h, c_->hist_cap), "altro_set_record_history");
    c_->record_history = on;
  }
  // rows the SolverStats vectors can reach under the current options (one per iteration + the initial row)
  int HistoryRowsNeeded() const { return std::max(kHistoryCapacity, c_->opts.max_iterations_total + 2); }

  // KnotPointFunctions::GetCostToGoHessian / Gradient need P, p of every knot in memory; a solve only needs them in
  // registers, and the persistent tail kernel -- the fast path of exactly the small batches a facade user solves -- only
  // runs without the recording.  Default for batches of up to kHistoryBatchLimit instances: the STEP-LEVEL BackwardPass()
  // records (that is where the reference's tests read the cost-to-go: test/ilqr/unicycle_ilqr_test.cpp:39-54), Solve()
  // does not -- and P, p are readable behind it all the same, as in the reference: the first read makes the library run the
  // last iteration's backward pass once more with the records on (same kernels, same inputs, same values; round 5).
  // SetRecordCostToGo(true) records during every solve (which then takes the batched kernels only), (false) never.
  void SetRecordCostToGo(bool on) {
    ctg_auto_ = false;
    ApplyRecordCtg(on);
  }
  void ApplyRecordCtg(bool on) {
    if (on == record_ctg_) return;
    detail::Check(Need(), altro_set_record_ctg(c_->h, on ? 1 : 0), "altro_set_record_ctg");
    record_ctg_ = on;
  }
  // called in front of every whole solve (also by AugmentedLagrangianiLQR::Solve): recording policy and a history
  // buffer large enough for the iteration caps in force
  void PrepareSolve() {
    if (ctg_auto_) ApplyRecordCtg(false);
    if (c_->record_history && HistoryRowsNeeded() > c_->hist_cap) SetRecordHistory(true);
  }

  std::shared_ptr<Trajectory<n, m>> GetTrajectory() { return c_->traj; }
  void SetTrajectory(std::shared_ptr<Trajectory<n, m>> traj) {  // ilqr.hpp:231-235
    c_->traj = std::move(traj);
    c_->pushed = false;
  }
  std::shared_ptr<Trajectory<n, m>> MakeTrajectory(float dt) {  // ilqr.hpp:216-221
    auto Z = std::make_shared<Trajectory<n, m>>(c_->N, c_->B);
    Z->SetUniformStep(dt);
    SetTrajectory(Z);
    return Z;
  }
  KnotPointFunctions<n, m> GetKnotPointFunction(int k, int b = 0) {
    Need();
    return KnotPointFunctions<n, m>(c_, k, b);
  }

  void Solve() {  // ilqr.hpp:284-316
    Push();
    PrepareSolve();
    detail::Check(c_->h, altro_solve_ilqr(c_->h), "altro_solve_ilqr");
    Pull(true, true);
    AfterSolve();
  }
  // What SolverOptions::verbose asks for, after the fact: the rows the reference's logger prints while it iterates
  // (solver_stats.cpp:80-116: iters, cost, viol, dJ | grad | alpha | reg, z | pen by level) of the instance GetStats()
  // shows.  Below kInner only the last row (the state at the end of the solve), from kInner on every recorded iteration.
  void PrintLog(FILE* f = stdout) const {
    const SolverStats& S = c_->stats;
    const int lvl = static_cast<int>(c_->opts.verbose);
    if (lvl <= 0 || S.cost.empty()) return;
    const size_t rows = S.cost.size();
    auto header = [&]() {
      std::fprintf(f, "%6s %12s %11s %10s", "iters", "cost", "viol", "dJ");
      if (lvl >= 2) std::fprintf(f, " %10s", "grad");
      if (lvl >= 3) std::fprintf(f, " %6s", "alpha");
      if (lvl >= 4) std::fprintf(f, " %8s %7s", "reg", "z");
      if (lvl >= 5) std::fprintf(f, " %8s", "pen");
      std::fprintf(f, "\n");
    };
    const int freq = std::max(c_->opts.header_frequency, 1);
    int printed = 0;
    for (size_t i = lvl >= 3 ? 0 : rows - 1; i < rows; ++i, ++printed) {
      if (printed % freq == 0) header();
      std::fprintf(f, "%6zu %12.4g %11.3e %10.2e", i, S.cost[i], S.violations[i], S.cost_decrease[i]);
      if (lvl >= 2) std::fprintf(f, " %10.2e", S.gradient[i]);
      if (lvl >= 3) std::fprintf(f, " %6.2f", S.alpha[i]);
      if (lvl >= 4) std::fprintf(f, " %8.1e %7.3f", S.regularization[i], S.improvement_ratio[i]);
      if (lvl >= 5) std::fprintf(f, " %8.1e", S.max_penalty[i]);
      std::fprintf(f, "\n");
    }
  }
  // bookkeeping behind every whole solve: the log (verbose) and the profile sums (profiler_enable)
  void AfterSolve() {
    if (c_->opts.verbose != LogLevel::kSilent) PrintLog();
    if (c_->opts.profiler_enable) {
      altro_timing t;
      if (altro_get_timing(c_->h, &t) == ALTRO_OK) {
        altro_timing& a = c_->prof;
        a.total_ms += t.total_ms; a.init_ms += t.init_ms; a.expansions_ms += t.expansions_ms;
        a.backward_pass_ms += t.backward_pass_ms; a.forward_pass_ms += t.forward_pass_ms; a.fused_ms += t.fused_ms;
        a.sweeps += t.sweeps; a.fused_sweeps += t.fused_sweeps; a.launches += t.launches;
        a.sweep_launches += t.sweep_launches; a.instance_iterations += t.instance_iterations;
        c_->prof_solves++;
      }
    }
  }
  void SolveSetup() {  // ilqr.hpp:629-645
    Push();
    detail::Check(c_->h, altro_solve_setup(c_->h), "altro_solve_setup");
    ++c_->epoch;
  }
  void Rollout() {  // ilqr.hpp:453-459
    Push();
    detail::Check(c_->h, altro_rollout(c_->h), "altro_rollout");
    Pull(true, false);
  }
  double Cost(int b = 0) {  // ilqr.hpp:326-334
    std::vector<double> J(c_->B);
    Push();
    detail::Check(c_->h, altro_cost(c_->h, J.data()), "altro_cost");
    return J[b];
  }
  void UpdateExpansions() {  // ilqr.hpp:350-358
    Push();
    detail::Check(c_->h, altro_update_expansions(c_->h), "altro_update_expansions");
    ++c_->epoch;
  }
  void BackwardPass() {  // ilqr.hpp:385-445
    PushOptions();
    if (ctg_auto_) ApplyRecordCtg(true);
    detail::Check(c_->h, altro_backward_pass(c_->h), "altro_backward_pass");
    ++c_->epoch;
    PullStats();
  }
  void ForwardPass() {  // ilqr.hpp:512-558
    PushOptions();
    detail::Check(c_->h, altro_forward_pass(c_->h), "altro_forward_pass");
    Pull(true, true);
  }
  void UpdateConvergenceStatistics() {  // ilqr.hpp:568-587
    PushOptions();
    detail::Check(c_->h, altro_update_convergence_statistics(c_->h), "altro_update_convergence_statistics");
    PullStats();
  }
  // solver.GetStats().Reset() of the reference (solver_stats.cpp:31-45), on the host mirror and on the device
  void ResetStats() {
    detail::Check(Need(), altro_reset_stats(c_->h), "altro_reset_stats");
    c_->stats.Reset();
  }
  // the reference's task decomposition of UpdateExpansions (ilqr.hpp:183-214) has no counterpart: every
  // (instance, knot) pair is its own GPU thread
  int NumThreads() const { return 1; }
  int NumTasks() const { return 1; }

  // ---- plumbing shared with AugmentedLagrangianiLQR ---------------------------------------------------
  void PushOptions() {
    o_ = c_->opts.ToC();
    detail::Check(Need(), altro_set_options(c_->h, &o_), "altro_set_options");
  }
  // make the device see the caller's options and trajectory (the trajectory object is shared, so the caller
  // may have edited or replaced it since the last call: ilqr.hpp:223-235)
  void Push() {
    PushOptions();
    if (c_->traj && !c_->pushed) {
      auto& Z = *c_->traj;
      if (Z.BatchSize() != c_->B || Z.NumSegments() != c_->N) throw std::runtime_error("Trajectory size isn't consistent with the solver.");
      if (Z.IsUniformStep()) {
        if (Z.GetStep(0) > 0.0f) detail::Check(c_->h, altro_set_uniform_step(c_->h, Z.GetStep(0)), "altro_set_uniform_step");
      } else if (Z.GetStep(0) > 0.0f) {  // Trajectory::SetStep / SetTime per knot (trajectory.hpp:119-120)
        detail::Check(c_->h, altro_set_steps(c_->h, Z.Steps().data(), Z.NumSegments()), "altro_set_steps");
        detail::Check(c_->h, altro_set_times(c_->h, Z.Times().data(), Z.NumSegments() + 1), "altro_set_times");
      }
      detail::Check(c_->h, altro_set_trajectory(c_->h, Z.States().data(), Z.Controls().data(), 1), "altro_set_trajectory");
      c_->pushed = true;
    }
  }
  void MarkTrajectoryDirty() { c_->pushed = false; }
  void Pull(bool with_traj, bool with_stats) {
    ++c_->epoch;
    if (with_traj && c_->traj) {
      auto& Z = *c_->traj;
      detail::Check(c_->h, altro_get_trajectory(c_->h, Z.States().data(), Z.Controls().data()), "altro_get_trajectory");
    }
    if (with_stats) PullStats();
  }
  void PullStats() {
    SolverStats& S = c_->stats;
    S.instances.resize(c_->B);
    detail::Check(c_->h, altro_get_stats(c_->h, S.instances.data()), "altro_get_stats");
    const int b = std::min(std::max(c_->stats_instance, 0), c_->B - 1);
    const altro_stats& s = S.instances[b];
    status_ = static_cast<SolverStatus>(s.status_ilqr);
    status_al_ = static_cast<SolverStatus>(s.status);
    S.initial_cost = s.initial_cost;
    S.iterations_inner = s.iterations_inner;
    S.iterations_outer = s.iterations_outer;
    S.iterations_total = s.iterations_total;
    // field order of altro_get_history: cost, alpha, improvement_ratio, gradient, cost_decrease, regularization,
    // violations, max_penalty.  History rows = the rows closed by NewIteration; the latest row is the open one.
    std::vector<double>* vec[8] = {&S.cost, &S.alpha, &S.improvement_ratio, &S.gradient, &S.cost_decrease,
                                   &S.regularization, &S.violations, &S.max_penalty};
    const double last[8] = {s.cost, s.alpha, s.improvement_ratio, s.gradient, s.cost_decrease, s.regularization,
                            s.violation, s.max_penalty};
    const int cap = std::max(c_->hist_cap, 1);
    std::vector<double> buf((size_t)8 * cap);
    const int cnt = c_->record_history ? altro_get_history_all(c_->h, b, buf.data(), cap) : 0;  // one call, one sync
    for (int f = 0; f < 8; ++f) {
      vec[f]->clear();
      if (cnt > 0) vec[f]->assign(buf.begin() + (size_t)f * cap, buf.begin() + (size_t)f * cap + cnt);
      vec[f]->push_back(last[f]);
    }
  }
  SolverStatus StatusAL() const { return status_al_; }
  std::shared_ptr<Core> CorePtr() { return c_; }

 private:
  altro_handle Need() const {
    if (!c_->h) throw std::runtime_error("The solver has not been initialized with a problem.");
    return c_->h;
  }
  std::shared_ptr<Core> c_;
  SolverStatus status_ = SolverStatus::kUnsolved;
  SolverStatus status_al_ = SolverStatus::kUnsolved;
  altro_options o_{};
  bool record_ctg_ = false;
  bool ctg_auto_ = false;
};
}  // namespace ilqr

namespace constraints {
// altro/constraints/constraint_values.hpp:33-300: the values a constraint carries inside the AL cost -- here a LIVE view of
// one constraint of one knot of instance `b` on the device: every getter downloads what the solver holds now (the
// reference hands out the object itself; auglag_test.cpp:256-270 reads its duals again after UpdateDuals()).
template <int n, int m, class ConType>
class ConstraintValues {
 public:
  ConstraintValues(std::shared_ptr<ilqr::detail_ilqr::Core<n, m>> core, int row0, int p, int rows, int b, std::string label)
      : c_(std::move(core)), row0_(row0), p_(p), rows_(rows), b_(b), label_(std::move(label)) {}
  int OutputDimension() const { return p_; }           // constraint_values.hpp:100
  const std::string& GetLabel() const { return label_; }
  std::vector<double> GetDuals() const { return Rows(&altro_get_duals, "altro_get_duals"); }                      // lambda_
  std::vector<double> GetPenalty() const { return Rows(&altro_get_penalties, "altro_get_penalties"); }            // penalty_
  std::vector<double> GetConstraintValue() const { return Rows(&altro_get_constraint_values, "altro_get_constraint_values"); }  // c_

 private:
  std::vector<double> Rows(altro_status (*get)(altro_handle, double*), const char* what) const {
    std::vector<double> all((size_t)c_->B * rows_);
    if (!all.empty()) detail::Check(c_->h, get(c_->h, all.data()), what);
    const size_t at = (size_t)b_ * rows_ + row0_;
    return std::vector<double>(all.begin() + at, all.begin() + at + p_);
  }
  std::shared_ptr<ilqr::detail_ilqr::Core<n, m>> c_;
  int row0_, p_, rows_, b_;
  std::string label_;
};
}  // namespace constraints

namespace augmented_lagrangian {
// altro/augmented_lagrangian/al_cost.hpp:41-120: the AL cost of one knot, as far as callers look into it -- its constraint
// values by cone, in the order the constraints were added (example_unicycle_test.cpp:66,105; auglag_test.cpp:256-258)
template <int n, int m>
class ALCost {
 public:
  using EqVals = std::shared_ptr<constraints::ConstraintValues<n, m, constraints::Equality>>;
  using IneqVals = std::shared_ptr<constraints::ConstraintValues<n, m, constraints::Inequality>>;
  const std::vector<EqVals>& GetEqualityConstraints() const { return eq_; }        // al_cost.hpp:100
  const std::vector<IneqVals>& GetInequalityConstraints() const { return ineq_; }  // al_cost.hpp:103
  int NumConstraints() const { return rows_; }                                      // al_cost.hpp:96
  std::vector<EqVals> eq_;
  std::vector<IneqVals> ineq_;
  int rows_ = 0;
};
// altro/augmented_lagrangian/al_solver.hpp:28-224
template <int n, int m>
class AugmentedLagrangianiLQR {
 public:
  explicit AugmentedLagrangianiLQR(int N) : ilqr_solver_(N) {}  // al_solver.hpp:35
  explicit AugmentedLagrangianiLQR(const problem::Problem& prob, int dtype = ALTRO_F64, int device_id = 0)
      : ilqr_solver_(prob.NumSegments()) {  // al_solver.hpp:231-237
    InitializeFromProblem(prob, dtype, device_id);
  }
  AugmentedLagrangianiLQR(const AugmentedLagrangianiLQR&) = delete;
  AugmentedLagrangianiLQR& operator=(const AugmentedLagrangianiLQR&) = delete;
  AugmentedLagrangianiLQR(AugmentedLagrangianiLQR&&) noexcept = default;  // returned by value by MakeALSolver (unicycle.hpp:111-121)

  void InitializeFromProblem(const problem::Problem& prob, int dtype = ALTRO_F64, int device_id = 0) {  // al_solver.hpp:239-251
    ilqr_solver_.InitializeFromProblem(BuildAugLagProblem<n, m>(prob), dtype, device_id);
  }

  SolverStats& GetStats() { return ilqr_solver_.GetStats(); }
  SolverOptions& GetOptions() { return ilqr_solver_.GetOptions(); }
  SolverStatus GetStatus() const { return status_; }
  ilqr::iLQR<n, m>& GetiLQRSolver() { return ilqr_solver_; }
  int NumSegments() const { return ilqr_solver_.NumSegments(); }
  int BatchSize() const { return ilqr_solver_.BatchSize(); }
  altro_handle Handle() { return ilqr_solver_.Handle(); }
  // al_solver.hpp:253-271 ("Cannot query the number of constraints before initializing the solver with a problem.")
  int NumConstraints() const { return altro_num_constraints(ilqr_solver_.Handle()); }
  int NumConstraints(int k) const { return altro_num_constraints_at(ilqr_solver_.Handle(), k); }

  void SetTrajectory(std::shared_ptr<Trajectory<n, m>> traj) { ilqr_solver_.SetTrajectory(std::move(traj)); }
  void SetPenalty(double rho) { detail::Check(Handle(), altro_set_penalty(Handle(), rho), "altro_set_penalty"); }
  void SetPenaltyScaling(double phi) { detail::Check(Handle(), altro_set_penalty_scaling(Handle(), phi), "altro_set_penalty_scaling"); }

  void Solve() {  // al_solver.hpp:304-334
    ilqr_solver_.MarkTrajectoryDirty();  // the caller may have refilled the shared trajectory (auglag_test.cpp:366)
    ilqr_solver_.Push();
    ilqr_solver_.PrepareSolve();
    detail::Check(Handle(), altro_solve_al(Handle()), "altro_solve_al");
    ilqr_solver_.Pull(true, true);
    status_ = ilqr_solver_.StatusAL();
    ilqr_solver_.AfterSolve();
  }
  // al_solver.hpp:287-302: duals and penalties reset as the options say, statistics reset, "viol" and "pen" logged
  void Init() {
    ilqr_solver_.Push();  // (options and the caller's trajectory)
    detail::Check(Handle(), altro_al_init(Handle()), "altro_al_init");
  }
  // al_solver.hpp:215, al_cost.hpp:372-379: every multiplier of every instance back to zero
  void ResetDualVariables() {
    std::vector<double> zero((size_t)BatchSize() * NumConstraints(), 0.0);
    if (!zero.empty()) detail::Check(Handle(), altro_set_duals(Handle(), zero.data()), "altro_set_duals");
  }
  // al_solver.hpp:46: the AL cost of knot k (of instance b): live views of its constraint values by cone
  std::shared_ptr<ALCost<n, m>> GetALCost(int k, int b = 0) {
    auto core = ilqr_solver_.CorePtr();
    const auto& cons = core->cons;
    if (k < 0 || k >= (int)cons.size()) throw std::out_of_range("GetALCost: knot index");
    const int R = NumConstraints();
    int row = 0;
    for (int j = 0; j < k; ++j)
      for (const examples::ConstraintDesc& cd : cons[j]) row += cd.OutputDimension();
    auto cost = std::make_shared<ALCost<n, m>>();
    for (const examples::ConstraintDesc& cd : cons[k]) {
      const int p = cd.OutputDimension();
      if (cd.IsEquality())
        cost->eq_.push_back(std::make_shared<constraints::ConstraintValues<n, m, constraints::Equality>>(core, row, p, R, b, cd.label));
      else
        cost->ineq_.push_back(std::make_shared<constraints::ConstraintValues<n, m, constraints::Inequality>>(core, row, p, R, b, cd.label));
      row += p;
      cost->rows_ += p;
    }
    return cost;
  }
  void UpdateDuals() { detail::Check(Handle(), altro_update_duals(Handle()), "altro_update_duals"); }
  void UpdatePenalties() { detail::Check(Handle(), altro_update_penalties(Handle()), "altro_update_penalties"); }
  double MaxViolation(int b = 0) {  // al_solver.hpp:403-408: evaluates the cost first
    ilqr_solver_.Push();
    std::vector<double> v(BatchSize());
    detail::Check(Handle(), altro_max_violation(Handle(), v.data()), "altro_max_violation");
    return v[b];
  }
  double GetMaxViolation(int b = 0) {
    std::vector<double> v(BatchSize());
    detail::Check(Handle(), altro_get_max_violation(Handle(), v.data()), "altro_get_max_violation");
    return v[b];
  }
  double GetMaxPenalty(int b = 0) {
    std::vector<double> v(BatchSize());
    detail::Check(Handle(), altro_get_max_penalty(Handle(), v.data()), "altro_get_max_penalty");
    return v[b];
  }
  std::vector<double> GetDuals() {
    std::vector<double> lam((size_t)BatchSize() * NumConstraints());
    if (!lam.empty()) detail::Check(Handle(), altro_get_duals(Handle(), lam.data()), "altro_get_duals");
    return lam;
  }

  // al_solver.hpp:83-104: label, knot index and violation vector c - Pi_K(c) of every constraint of instance b,
  // from the constraint values the last evaluation left on the device; optionally sorted by max violation.
  std::vector<constraints::ConstraintInfo> GetConstraintInfo(bool should_sort = false, int b = 0) {
    const int R = NumConstraints();
    std::vector<double> c((size_t)BatchSize() * R);
    if (!c.empty()) detail::Check(Handle(), altro_get_constraint_values(Handle(), c.data()), "altro_get_constraint_values");
    std::vector<constraints::ConstraintInfo> coninfo;
    const auto& cons = ilqr_solver_.CorePtr()->cons;
    size_t row = (size_t)b * R;
    for (int k = 0; k <= NumSegments(); ++k)
      for (const examples::ConstraintDesc& cd : cons[k]) {
        constraints::ConstraintInfo info;
        info.label = cd.label;
        info.index = k;
        info.type = cd.GetConstraintType();
        const int p = cd.OutputDimension();
        for (int i = 0; i < p; ++i, ++row) info.violation.push_back(cd.IsEquality() ? c[row] : std::max(c[row], 0.0));
        coninfo.push_back(std::move(info));
      }
    if (should_sort)
      std::stable_sort(coninfo.begin(), coninfo.end(), [](const constraints::ConstraintInfo& a, const constraints::ConstraintInfo& b2) {
        return a.MaxViolation() > b2.MaxViolation();
      });
    return coninfo;
  }
  void PrintViolations(bool should_sort = false, int precision = 4, FILE* f = stdout) {  // al_solver.hpp:68-81
    const std::vector<constraints::ConstraintInfo> coninfo = GetConstraintInfo(should_sort);
    std::fprintf(f, "Got %zu constraints\n", coninfo.size());
    for (const constraints::ConstraintInfo& info : coninfo) std::fprintf(f, "%s\n", info.ToString(precision).c_str());
  }

  // Non-blocking Solve() for the MPC pattern (docs/Overview.dox:48-54): SolveAsync() returns at once, the
  // caller prepares the next problem, Wait() blocks and refreshes the trajectory and the statistics.
  void SolveAsync() {
    ilqr_solver_.MarkTrajectoryDirty();
    ilqr_solver_.Push();
    ilqr_solver_.PrepareSolve();
    detail::Check(Handle(), altro_solve_al_async(Handle()), "altro_solve_al_async");
  }
  bool Poll() {
    int done = 0;
    detail::Check(Handle(), altro_solve_poll(Handle(), &done), "altro_solve_poll");
    return done != 0;
  }
  void Wait() {
    detail::Check(Handle(), altro_wait(Handle()), "altro_wait");
    ilqr_solver_.Pull(true, true);
    status_ = ilqr_solver_.StatusAL();
    ilqr_solver_.AfterSolve();
  }
  altro_timing GetTiming() {
    altro_timing t;
    detail::Check(Handle(), altro_get_timing(Handle(), &t), "altro_get_timing");
    return t;
  }
  // Prints the device timing of the last solve (needs SolverOptions::profiler_enable) as a tree in the
  // layout of the reference's Timer (altro/common/timer.cpp:24-94, profile_entry.cpp:36-66; sample:
  // perf/profiler_unicycle.out), with the reference's section names.  "sweep_fused" is this build's
  // persistent tail launch: whole iterations (expansions + backward_pass + forward_pass) of the straggler
  // instances.  forward_pass includes the reference's cost / rollout / stats / dual_update /
  // penalty_update / convergence_check work, which the forward kernel performs.
  void PrintTimings(FILE* f = stdout) {
    detail::PrintTimingTree(f, GetTiming());
    ilqr_solver_.CorePtr()->prof_printed = true;  // (the destructor prints the sums only if nobody asked before: timer.cpp:10-14)
  }

 private:
  ilqr::iLQR<n, m> ilqr_solver_;
  SolverStatus status_ = SolverStatus::kUnsolved;
};
}  // namespace augmented_lagrangian

}  // namespace altro
