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

#include <cmath>
#include <cstddef>
#include <limits>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
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

// altro/common/solver_options.hpp:19-57 (same field names and defaults; console fields omitted)
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
  bool profiler_enable = false;

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

// altro/common/solver_stats.hpp:44-63.  Scalars and `.back()` values of instance 0 after a solve;
// AllInstances() gives the per-instance records of the batch.
struct SolverStats {
  double initial_cost = 0.0;
  int iterations_inner = 0;
  int iterations_outer = 0;
  int iterations_total = 0;
  std::vector<double> cost, alpha, improvement_ratio, gradient, cost_decrease, regularization, violations,
      max_penalty;
  std::vector<altro_stats> instances;
  const std::vector<altro_stats>& AllInstances() const { return instances; }
};

namespace detail {
inline void Check(altro_handle h, altro_status st, const char* what) {
  if (st != ALTRO_OK) {
    const char* msg = altro_last_error(h);
    throw std::runtime_error(std::string(what) + " failed (" + std::to_string((int)st) + "): " + (msg ? msg : ""));
  }
}
}  // namespace detail

// altro/common/trajectory.hpp:24-160.  Host storage of the states, controls and the (float) step of
// `batch` instances; it is the initial guess AND the output of a solve (altro/ilqr/ilqr.hpp:223-235).
template <int n, int m>
class Trajectory {
 public:
  explicit Trajectory(int N, int batch = 1)
      : N_(N), B_(batch), X_((size_t)batch * (N + 1) * n, 0.0), U_((size_t)batch * N * m, 0.0), h_(0.0f) {}
  int NumSegments() const { return N_; }
  int BatchSize() const { return B_; }
  double* State(int k, int b = 0) { return &X_[((size_t)b * (N_ + 1) + k) * n]; }
  double* Control(int k, int b = 0) { return &U_[((size_t)b * N_ + k) * m]; }
  const double* State(int k, int b = 0) const { return &X_[((size_t)b * (N_ + 1) + k) * n]; }
  const double* Control(int k, int b = 0) const { return &U_[((size_t)b * N_ + k) * m]; }
  void SetUniformStep(float h) { h_ = h; }                        // trajectory.hpp:122-130
  float GetStep(int k) const { return k < N_ ? h_ : 0.0f; }       // terminal knot has h = 0
  float GetTime(int k) const { return k < N_ ? static_cast<float>(k) * h_ : h_ * N_; }
  void SetZero() {
    std::fill(X_.begin(), X_.end(), 0.0);
    std::fill(U_.begin(), U_.end(), 0.0);
  }
  std::vector<double>& States() { return X_; }
  std::vector<double>& Controls() { return U_; }

 private:
  int N_, B_;
  std::vector<double> X_, U_;
  float h_;
};

// ---- descriptor types with the reference's class names --------------------------------------------
namespace examples {
struct Unicycle {  // examples/unicycle.hpp
  static constexpr int kind = ALTRO_MODEL_UNICYCLE;
  int StateDimension() const { return 3; }
  int ControlDimension() const { return 2; }
  std::vector<double> Params() const { return {}; }
};
struct TripleIntegrator {  // examples/triple_integrator.hpp
  static constexpr int kind = ALTRO_MODEL_TRIPLE_INTEGRATOR;
  explicit TripleIntegrator(int dof = 1) : dof_(dof) {}
  int StateDimension() const { return 3 * dof_; }
  int ControlDimension() const { return dof_; }
  std::vector<double> Params() const { return {static_cast<double>(dof_)}; }
  int dof_;
};
struct Quadrotor12 {  // build-defined model of BASELINE config 5
  static constexpr int kind = ALTRO_MODEL_QUADROTOR12;
  int StateDimension() const { return 12; }
  int ControlDimension() const { return 4; }
  std::vector<double> Params() const { return {}; }
};

// examples/quadratic_cost.hpp:29-39.  xref may hold one reference or `batch` references.
struct QuadraticCost {
  std::vector<double> Q, R, xref, uref;
  bool terminal = false;
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
    return Q == o.Q && R == o.R && xref == o.xref && uref == o.uref && terminal == o.terminal;
  }
};

struct ConstraintDesc {
  int kind = 0;
  std::vector<double> params;  // one instance's block, or batch blocks back to back
  int nparams = 0;             // length of one instance's block
  std::string label;
  bool operator==(const ConstraintDesc& o) const { return kind == o.kind && params == o.params && nparams == o.nparams; }
};
// examples/basic_constraints.hpp:15-40
struct GoalConstraint : ConstraintDesc {
  explicit GoalConstraint(const std::vector<double>& xf, int n = -1) {
    kind = ALTRO_CON_GOAL;
    params = xf;
    nparams = n > 0 ? n : static_cast<int>(xf.size());
    label = "Goal Constraint";
  }
};
// examples/basic_constraints.hpp:42-151
struct ControlBound : ConstraintDesc {
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
struct CircleConstraint : ConstraintDesc {
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
// problem/discretized_model.hpp:24-65: RK4 is the only integrator of the device path.
template <class Model>
struct DiscretizedModel {
  explicit DiscretizedModel(const Model& m) : model(m) {}
  Model model;
};

// altro/problem/problem.hpp:65-307
class Problem {
 public:
  explicit Problem(int N) : N_(N), costs_(N + 1), has_cost_(N + 1, false), cons_(N + 1), has_dyn_(N + 1, false) {}
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
  template <class Model>
  void SetDynamics(const DiscretizedModel<Model>& dm, int k) {
    Range(k);
    if (k >= N_) throw std::runtime_error("dynamics are set on knots 0..N-1");
    model_kind_ = Model::kind;
    model_params_ = dm.model.Params();
    n_ = dm.model.StateDimension();
    m_ = dm.model.ControlDimension();
    has_dyn_[k] = true;
    if (k == N_ - 1) has_dyn_[N_] = true;  // IdentityDynamics at the terminal knot (problem.hpp:161-164)
  }
  void SetConstraint(const examples::ConstraintDesc& con, int k) {
    Range(k);
    cons_[k].push_back(con);
  }
  int NumConstraints(int k) const {
    int p = 0;
    for (const auto& c : cons_[k]) p += Rows(c);
    return p;
  }
  int NumConstraints() const {
    int p = 0;
    for (int k = 0; k <= N_; ++k) p += NumConstraints(k);
    return p;
  }
  bool IsFullyDefined() const {  // problem.hpp:271-297
    for (int k = 0; k <= N_; ++k)
      if (!has_cost_[k] || !has_dyn_[k]) return false;
    return !x0_.empty();
  }

  // Replay the definition through the C-ABI (consecutive knots with identical descriptors become
  // one [k_begin, k_end) call).
  void Apply(altro_handle h) const {
    using detail::Check;
    Check(h, altro_set_model(h, model_kind_, model_params_.empty() ? nullptr : model_params_.data(),
                             (int)model_params_.size()), "altro_set_model");
    for (int k = 0; k <= N_;) {
      int e = k + 1;
      while (e <= N_ && costs_[e] == costs_[k]) ++e;
      const auto& c = costs_[k];
      const int per = ((int)c.xref.size() > n_ ? 1 : 0) | ((int)c.uref.size() > m_ ? 2 : 0);
      Check(h, altro_set_lqr_cost(h, k, e, c.Q.data(), c.R.data(), c.xref.data(), c.uref.data(), per),
            "altro_set_lqr_cost");
      k = e;
    }
    // insertion order within a knot is what matters (al_cost.hpp:267-272): emit constraint j of each knot
    size_t maxc = 0;
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
  static int Rows(const examples::ConstraintDesc& c) {
    if (c.kind == ALTRO_CON_GOAL) return c.nparams;
    if (c.kind == ALTRO_CON_CIRCLE) return c.nparams / 3;
    int p = 0;
    for (int i = 0; i < c.nparams; ++i)
      if (std::abs(c.params[i]) < std::numeric_limits<double>::max()) ++p;
    return p;
  }
  int N_, batch_ = 1, n_ = 0, m_ = 0, model_kind_ = 0;
  std::vector<double> model_params_, x0_;
  std::vector<examples::QuadraticCost> costs_;
  std::vector<bool> has_cost_;
  std::vector<std::vector<examples::ConstraintDesc>> cons_;
  std::vector<bool> has_dyn_;
};
}  // namespace problem

namespace ilqr {
// altro/ilqr/knot_point_function_type.hpp:243-268: read-only view of one knot of one instance.
template <int n, int m>
class KnotPointFunctions {
 public:
  KnotPointFunctions(altro_handle h, int N, int batch, int k, int b) : h_(h), N_(N), B_(batch), k_(k), b_(b) {}
  std::vector<double> GetFeedbackGain() const {  // m x n, column-major
    std::vector<double> K((size_t)B_ * N_ * m * n);
    detail::Check(h_, altro_get_gains(h_, K.data(), nullptr), "altro_get_gains");
    return Slice(K, ((size_t)b_ * N_ + k_) * m * n, m * n);
  }
  std::vector<double> GetFeedforwardGain() const {
    std::vector<double> d((size_t)B_ * N_ * m);
    detail::Check(h_, altro_get_gains(h_, nullptr, d.data()), "altro_get_gains");
    return Slice(d, ((size_t)b_ * N_ + k_) * m, m);
  }
  std::vector<double> GetCostToGoHessian() const {
    std::vector<double> P((size_t)B_ * (N_ + 1) * n * n);
    detail::Check(h_, altro_get_ctg(h_, P.data(), nullptr), "altro_get_ctg");
    return Slice(P, ((size_t)b_ * (N_ + 1) + k_) * n * n, n * n);
  }
  std::vector<double> GetCostToGoGradient() const {
    std::vector<double> p((size_t)B_ * (N_ + 1) * n);
    detail::Check(h_, altro_get_ctg(h_, nullptr, p.data()), "altro_get_ctg");
    return Slice(p, ((size_t)b_ * (N_ + 1) + k_) * n, n);
  }
  std::vector<double> GetDynamicsExpansion() const {  // [A|B], n x (n+m), column-major
    std::vector<double> AB((size_t)B_ * n * (n + m));
    detail::Check(h_, altro_get_expansion(h_, k_, AB.data(), nullptr, nullptr, nullptr, nullptr, nullptr),
                  "altro_get_expansion");
    return Slice(AB, (size_t)b_ * n * (n + m), n * (n + m));
  }

 private:
  static std::vector<double> Slice(const std::vector<double>& v, size_t off, size_t len) {
    return std::vector<double>(v.begin() + off, v.begin() + off + len);
  }
  altro_handle h_;
  int N_, B_, k_, b_;
};

// altro/ilqr/ilqr.hpp:47-813 (the algorithm methods; thread-pool and logging members omitted)
template <int n, int m>
class iLQR {
 public:
  iLQR(altro_handle h, int N, int batch, SolverOptions* opts, SolverStats* stats,
       std::shared_ptr<Trajectory<n, m>>* traj)
      : h_(h), N_(N), B_(batch), opts_(opts), stats_(stats), traj_(traj) {}
  int NumSegments() const { return N_; }
  SolverOptions& GetOptions() { return *opts_; }
  SolverStats& GetStats() { return *stats_; }
  SolverStatus GetStatus() const { return status_; }
  std::shared_ptr<Trajectory<n, m>> GetTrajectory() { return *traj_; }
  KnotPointFunctions<n, m> GetKnotPointFunction(int k, int b = 0) {
    detail::Check(h_, altro_set_record_ctg(h_, 1), "altro_set_record_ctg");
    return KnotPointFunctions<n, m>(h_, N_, B_, k, b);
  }
  void Solve() {
    Push();
    detail::Check(h_, altro_solve_ilqr(h_), "altro_solve_ilqr");
    Pull(true);
  }
  void Rollout() {
    Push();
    detail::Check(h_, altro_rollout(h_), "altro_rollout");
    Pull(false);
  }
  double Cost(int b = 0) {
    std::vector<double> J(B_);
    Push();
    detail::Check(h_, altro_cost(h_, J.data()), "altro_cost");
    return J[b];
  }
  void UpdateExpansions() {
    Push();
    detail::Check(h_, altro_update_expansions(h_), "altro_update_expansions");
  }
  void BackwardPass() {
    detail::Check(h_, altro_set_options(h_, &(o_ = opts_->ToC())), "altro_set_options");
    detail::Check(h_, altro_backward_pass(h_), "altro_backward_pass");
  }
  void ForwardPass() {
    detail::Check(h_, altro_forward_pass(h_), "altro_forward_pass");
    Pull(false);
  }
  // make the device see the caller's trajectory / options (the trajectory object is shared, so the
  // caller may have edited it since the last call: ilqr.hpp:223-235)
  void Push() {
    detail::Check(h_, altro_set_options(h_, &(o_ = opts_->ToC())), "altro_set_options");
    if (*traj_ && !pushed_) {
      auto& Z = **traj_;
      if (!step_set_) {  // the step is part of the problem definition: fixed after the first push
        detail::Check(h_, altro_set_uniform_step(h_, Z.GetStep(0)), "altro_set_uniform_step");
        step_set_ = true;
      }
      detail::Check(h_, altro_set_trajectory(h_, Z.States().data(), Z.Controls().data(), 1), "altro_set_trajectory");
      pushed_ = true;
    }
  }
  void MarkTrajectoryDirty() { pushed_ = false; }
  void Pull(bool with_stats) {
    if (*traj_) {
      auto& Z = **traj_;
      detail::Check(h_, altro_get_trajectory(h_, Z.States().data(), Z.Controls().data()), "altro_get_trajectory");
    }
    if (with_stats) {
      stats_->instances.resize(B_);
      detail::Check(h_, altro_get_stats(h_, stats_->instances.data()), "altro_get_stats");
      const altro_stats& s = stats_->instances[0];
      status_ = static_cast<SolverStatus>(s.status_ilqr);
      stats_->initial_cost = s.initial_cost;
      stats_->iterations_inner = s.iterations_inner;
      stats_->iterations_outer = s.iterations_outer;
      stats_->iterations_total = s.iterations_total;
      auto put = [](std::vector<double>& v, double x) { v.assign(1, x); };
      put(stats_->cost, s.cost);
      put(stats_->alpha, s.alpha);
      put(stats_->improvement_ratio, s.improvement_ratio);
      put(stats_->gradient, s.gradient);
      put(stats_->cost_decrease, s.cost_decrease);
      put(stats_->regularization, s.regularization);
      put(stats_->violations, s.violation);
      put(stats_->max_penalty, s.max_penalty);
    }
  }

 private:
  altro_handle h_;
  int N_, B_;
  SolverOptions* opts_;
  SolverStats* stats_;
  std::shared_ptr<Trajectory<n, m>>* traj_;
  SolverStatus status_ = SolverStatus::kUnsolved;
  altro_options o_{};
  bool pushed_ = false;
  bool step_set_ = false;
};
}  // namespace ilqr

namespace augmented_lagrangian {
// altro/augmented_lagrangian/al_solver.hpp:28-224
template <int n, int m>
class AugmentedLagrangianiLQR {
 public:
  explicit AugmentedLagrangianiLQR(const problem::Problem& prob, int dtype = ALTRO_F64, int device_id = 0)
      : N_(prob.NumSegments()), B_(prob.BatchSize()) {
    if (prob.StateDimension() != n || prob.ControlDimension() != m)
      throw std::runtime_error("Inconsistent state / control dimension.");
    if (!prob.IsFullyDefined()) throw std::runtime_error("Expected problem to be fully defined.");
    altro_desc d{n, m, N_, B_, dtype, device_id};
    altro_status st = altro_create(&d, &h_);
    if (st != ALTRO_OK) throw std::runtime_error(std::string("altro_create failed: ") + altro_last_error(nullptr));
    prob.Apply(h_);
    ilqr_.reset(new ilqr::iLQR<n, m>(h_, N_, B_, &opts_, &stats_, &traj_));
  }
  ~AugmentedLagrangianiLQR() { altro_destroy(h_); }
  AugmentedLagrangianiLQR(const AugmentedLagrangianiLQR&) = delete;
  AugmentedLagrangianiLQR& operator=(const AugmentedLagrangianiLQR&) = delete;

  SolverStats& GetStats() { return stats_; }
  SolverOptions& GetOptions() { return opts_; }
  SolverStatus GetStatus() const { return status_; }
  ilqr::iLQR<n, m>& GetiLQRSolver() { return *ilqr_; }
  int NumSegments() const { return N_; }
  int NumConstraints() const { return altro_num_constraints(h_); }
  int NumConstraints(int k) const { return altro_num_constraints_at(h_, k); }
  altro_handle Handle() { return h_; }

  void SetTrajectory(std::shared_ptr<Trajectory<n, m>> traj) {
    traj_ = std::move(traj);
    ilqr_->MarkTrajectoryDirty();
  }
  void SetPenalty(double rho) { detail::Check(h_, altro_set_penalty(h_, rho), "altro_set_penalty"); }
  void SetPenaltyScaling(double phi) { detail::Check(h_, altro_set_penalty_scaling(h_, phi), "altro_set_penalty_scaling"); }

  void Solve() {  // al_solver.hpp:304-334
    ilqr_->MarkTrajectoryDirty();
    ilqr_->Push();
    detail::Check(h_, altro_solve_al(h_), "altro_solve_al");
    ilqr_->Pull(true);
    status_ = static_cast<SolverStatus>(stats_.instances[0].status);
  }
  void UpdateDuals() { detail::Check(h_, altro_update_duals(h_), "altro_update_duals"); }
  void UpdatePenalties() { detail::Check(h_, altro_update_penalties(h_), "altro_update_penalties"); }
  double MaxViolation(int b = 0) {
    std::vector<double> v(B_);
    detail::Check(h_, altro_max_violation(h_, v.data()), "altro_max_violation");
    return v[b];
  }
  double GetMaxViolation(int b = 0) {
    std::vector<double> v(B_);
    detail::Check(h_, altro_get_max_violation(h_, v.data()), "altro_get_max_violation");
    return v[b];
  }
  double GetMaxPenalty(int b = 0) {
    std::vector<double> v(B_);
    detail::Check(h_, altro_get_max_penalty(h_, v.data()), "altro_get_max_penalty");
    return v[b];
  }
  std::vector<double> GetDuals() {
    std::vector<double> lam((size_t)B_ * NumConstraints());
    if (!lam.empty()) detail::Check(h_, altro_get_duals(h_, lam.data()), "altro_get_duals");
    return lam;
  }
  // Non-blocking Solve() for the MPC pattern (docs/Overview.dox:48-54): SolveAsync() returns at once, the
  // caller prepares the next problem, Wait() blocks and refreshes the trajectory and the statistics.
  void SolveAsync() {
    ilqr_->MarkTrajectoryDirty();
    ilqr_->Push();
    detail::Check(h_, altro_solve_al_async(h_), "altro_solve_al_async");
  }
  bool Poll() {
    int done = 0;
    detail::Check(h_, altro_solve_poll(h_, &done), "altro_solve_poll");
    return done != 0;
  }
  void Wait() {
    detail::Check(h_, altro_wait(h_), "altro_wait");
    ilqr_->Pull(true);
    status_ = static_cast<SolverStatus>(stats_.instances[0].status);
  }
  altro_timing GetTiming() {
    altro_timing t;
    detail::Check(h_, altro_get_timing(h_, &t), "altro_get_timing");
    return t;
  }
  // Prints the device timing of the last solve (needs SolverOptions::profiler_enable) as a tree in the
  // layout of the reference's Timer (altro/common/timer.cpp:24-94, profile_entry.cpp:36-66; sample:
  // perf/profiler_unicycle.out), with the reference's section names.  "sweep_fused" is this build's
  // persistent tail launch: whole iterations (expansions + backward_pass + forward_pass) of the straggler
  // instances.  forward_pass includes the reference's cost / rollout / stats / dual_update /
  // penalty_update / convergence_check work, which the forward kernel performs.
  void PrintTimings(FILE* f = stdout) {
    const altro_timing t = GetTiming();
    const double total = t.total_ms * 1e3;
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
    std::fprintf(f, "sweeps %d (tail iterations in the fused launch: %d), kernel launches %d, instance-iterations %lld\n",
                 t.sweeps, t.fused_sweeps, t.launches, t.instance_iterations);
  }

 private:
  int N_, B_;
  altro_handle h_ = nullptr;
  SolverOptions opts_;
  SolverStats stats_;
  SolverStatus status_ = SolverStatus::kUnsolved;
  std::shared_ptr<Trajectory<n, m>> traj_;
  std::unique_ptr<ilqr::iLQR<n, m>> ilqr_;
};
}  // namespace augmented_lagrangian

}  // namespace altro
