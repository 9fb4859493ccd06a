// altro_common.hpp — host-side types shared by the C-ABI front end (altro_capi.cpp) and the
// templated device engines (altro_engine.hpp).  No HIP types in here.
#pragma once

#include <atomic>
#include <string>
#include <vector>

#include "../../include/altro_hip.h"

namespace altro_hip {

// ---- compile-time limits of the closed registry ---------------------------------------------------
constexpr int kMaxConPerKnot = 4;  // constraints attached to one knot point
constexpr int kMaxClasses = 8;     // distinct (cost, constraint list) combinations over the knots
constexpr int kMaxCostGroups = 8;
#ifndef ALTRO_LS_LANES
#define ALTRO_LS_LANES 20  // (experimental builds: 10 -- six instances per wavefront, only with line_search_max_iterations <= 10)
#endif
constexpr int kLineSearchLanes = ALTRO_LS_LANES;  // speculative line-search trials evaluated side by side
constexpr int kHistFields = 8;
constexpr int kMaxRuns = 16;  // maximal runs of consecutive knots sharing one class
constexpr int kMaxFastCircles = 3;  // circles of one constraint that the specialised cost-wave layouts keep in registers
constexpr int kMaxSharedPool = 128;  // shared parameters that travel to the forward kernel as kernel arguments

// ---- dtype-independent problem specification (what the altro::problem::Problem setters record) ---
struct CostSpec {
  int k_begin, k_end;
  std::vector<double> Q, R, xref, uref;
  int per_instance;  // bit0 xref, bit1 uref (user cost: params are per instance)
  int user = 0;      // 0: LQR cost; 1 + t: the t-th user cost type of the model's source (ALTRO_USER_COSTS) instead
  std::vector<double> params;  // user cost: [nparams] or [B][nparams]
};
struct ConSpec {
  int kind, k_begin, k_end, nparams, per_instance;
  std::vector<double> params;
  int user_type = 0;  // ALTRO_CON_USER: index of the constraint type in the model's source (ALTRO_USER_CONSTRAINTS)
};
struct ProblemSpec {
  altro_desc desc{};
  int model_kind = 0;
  int dof = 0;
  float hstep = 0.0f;
  // Trajectory::SetStep(k, h) / SetTime(k, t) (trajectory.hpp:119-120): per-knot steps hk[N] and times tk[N + 1]; empty =
  // the uniform step above with t_k = float(k) * h, t_N = h * N (trajectory.hpp:122-130)
  std::vector<float> hk, tk;
  // Problem::SetDynamics(model, k) with different models along the horizon (problem.hpp:155-166): knot k uses model
  // knot_model[k] of the user source's ALTRO_USER_MODELS list; empty = model 0 everywhere
  std::vector<int> knot_model;
  std::vector<CostSpec> costs;
  std::vector<ConSpec> cons;
  std::vector<double> x0;  // [n] or [B][n]
  int x0_per_instance = 0;
  std::vector<double> X, U;  // host layout, instance-major
  // after the upload a new trajectory goes from the caller's buffers straight to the device: views for that one call
  const double* X_view = nullptr;
  const double* U_view = nullptr;
  bool has_X = false, has_U = false;
  int traj_per_instance = 0;
  double penalty = -1.0;  // SetPenalty issued before the device state exists
  double phi = -1.0;
};

// ---- device-visible problem description (lives in global memory, read with scalar loads) ---------
struct ConDesc {
  int kind;          // altro_constraint_kind
  int type;          // 0 equality (dual cone = identity), 1 inequality (negative orthant)
  int p;             // rows
  int per_instance;  // params in the per-instance pool ([slot][Bp]) instead of the shared pool
  int param_off;     // first slot / element of this constraint's parameters
  int row_off;       // row offset inside the knot
  unsigned lo_mask;  // CONTROL_BOUND: controls with a finite lower bound (basic_constraints.hpp:138-145); USER: index of the type
  unsigned hi_mask;  // CONTROL_BOUND: controls with a finite upper bound
};
struct KnotClass {
  int cost_group;
  int ncon;
  int nrows;
  int pad;
  ConDesc con[kMaxConPerKnot];
};
struct CostGroupDesc {
  int Q_off, R_off;          // shared pool, column-major n x n and m x m
  int q_off, r_off, c_off;   // pool element (shared) or slot (per instance)
  int q_pi, r_pi, c_pi;      // per-instance flags
  int q_diag, r_diag;        // Q / R are diagonal (every off-diagonal entry is exactly zero)
  int user, u_off, u_pi;     // user cost (altro_set_user_cost): 1 + index of its type; parameters at u_off (pool element / first slot)
};
// A run of consecutive knot points [k_begin, k_end) with the same class: rows of knot k start at
// rowbase + (k - k_begin) * nrows(cls).  Lets the serial kernels keep the class in scalar registers.
// `fast` selects a compile-time-specialised constraint layout for the serial rollout loop.
enum FastKind {
  kFastGeneric = 0,  // anything: table-driven evaluation
  kFastNone = 1,     // no constraint on these knots
  kFastB = 2,        // [CONTROL_BOUND with every lower and upper bound finite]
  kFastCB = 3,       // [CIRCLE, CONTROL_BOUND(full)]
  kFastBC = 4,       // [CONTROL_BOUND(full), CIRCLE]
  kFastC = 5         // [CIRCLE]
};
struct KnotRun {
  int k_begin, k_end, cls, rowbase;
  int fast, pad0, pad1, pad2;
};
struct ProblemDesc {
  int n, m, N, B, Bp;
  int nclass, ngroups, total_rows;
  int nruns, nslots;  // nslots: per-instance parameter slots (ipool rows)
  float hstep;        // uniform step (trajectory.hpp:122-130); the terminal knot's step is unused
  int npool;          // elements in the shared parameter pool
  KnotRun runs[kMaxRuns];
  KnotClass cls[kMaxClasses];
  CostGroupDesc grp[kMaxCostGroups];
};

// Options in the form the kernels consume (copied from altro_options at every launch).
struct DevOpts {
  int max_iterations_total, max_iterations_outer, max_iterations_inner;
  int bp_reg_fail_threshold, check_forwardpass_bounds, line_search_max_iterations, reset_duals;
  int fast_forward_stalls;  // opt-in (ALTRO_HIP_FAST_FORWARD_STALLS): see k_sweep_fused
  double cost_tolerance, gradient_tolerance;
  double bp_reg_increase_factor, bp_reg_initial, bp_reg_max, bp_reg_min;
  double state_max, control_max;
  double line_search_lower_bound, line_search_upper_bound, line_search_decrease_factor;
  double constraint_tolerance, maximum_penalty, initial_penalty;
};

// Engines of this process that run their sweeps as chains on streams of their own, PER DEVICE (hardware queues are a
// device's).  The count lives in libaltro_hip.so; a user-model plugin carries its own copy of this header, so the
// library hands every plugin a pointer to ITS counter function at load time (altro_user_set_chain_hook): built-in and
// plugin engines then share one book.  delta = +1 / -1 / 0 (query); returns the count after the change.
inline int ChainClaimLocal(int device, int delta) {
  static std::atomic<int> n[64];
  std::atomic<int>& c = n[device >= 0 && device < 64 ? device : 63];
  if (delta == 0) return c.load();
  return c.fetch_add(delta) + delta;
}
using ChainClaimFn = int (*)(int, int);
inline ChainClaimFn& ChainClaimHook() {
  static ChainClaimFn fn = &ChainClaimLocal;
  return fn;
}
inline int ChainClaim(int device, int delta) { return ChainClaimHook()(device, delta); }

// ---- engine interface ----------------------------------------------------------------------------
class EngineBase {
 public:
  virtual ~EngineBase() {}
  virtual altro_status Upload(const ProblemSpec& spec, std::string* err) = 0;
  virtual altro_status SetInitialState(const ProblemSpec& spec, std::string* err) = 0;
  virtual altro_status SetTrajectory(const ProblemSpec& spec, std::string* err) = 0;
  virtual altro_status SetStep(float hstep) = 0;
  // per-knot steps / times of the spec (or back to the uniform step when both are empty)
  virtual altro_status SetKnotTimes(const ProblemSpec& spec, std::string* err) = 0;
  virtual altro_status ResetTrajectory() = 0;
  virtual altro_status ResetStats() = 0;
  virtual altro_status SetPenalty(double rho) = 0;
  virtual altro_status SetPenaltyScaling(double phi) = 0;
  virtual altro_status SolveAL(const altro_options& o) = 0;
  virtual altro_status SolveILQR(const altro_options& o) = 0;
  virtual altro_status AlInit(const altro_options& o) = 0;
  virtual altro_status SolveSetup(const altro_options& o) = 0;
  virtual altro_status Rollout(const altro_options& o) = 0;
  virtual altro_status Cost(const altro_options& o, double* J) = 0;
  virtual altro_status UpdateExpansions(const altro_options& o) = 0;
  virtual altro_status BackwardPass(const altro_options& o) = 0;
  virtual altro_status ForwardPass(const altro_options& o) = 0;
  virtual altro_status UpdateConvergenceStatistics(const altro_options& o) = 0;
  virtual altro_status UpdateDuals(const altro_options& o) = 0;
  virtual altro_status UpdatePenalties(const altro_options& o) = 0;
  virtual altro_status GetMaxViolation(double* out) = 0;
  virtual altro_status GetMaxPenalty(double* out) = 0;
  virtual altro_status GetTrajectory(double* X, double* U) = 0;
  virtual altro_status GetGains(double* K, double* d) = 0;
  virtual altro_status SetRecordCtg(int enable) = 0;
  virtual altro_status GetCtg(double* P, double* p) = 0;
  virtual altro_status GetExpansion(int k, double* AB, double* lxx, double* lxu, double* luu,
                                    double* lx, double* lu) = 0;
  virtual altro_status GetKnotCosts(double* costs) = 0;
  virtual int NumRows() = 0;
  virtual int NumRowsAt(int k) = 0;
  virtual altro_status GetRows(int which /*0 lam,1 pen,2 cval*/, double* out) = 0;
  virtual altro_status SetDuals(const double* lam) = 0;
  virtual altro_status GetStats(altro_stats* st, bool ilqr_mode) = 0;
  virtual altro_status GetTiming(altro_timing* t) = 0;
  virtual altro_status SetRecordHistory(int capacity) = 0;
  virtual int GetHistory(int instance, int field, double* out, int cap) = 0;
  virtual int GetHistoryAll(int instance, double* out /*[kHistFields][cap]*/, int cap) = 0;
  virtual altro_status PackResultsDevice(void* dst) = 0;
  virtual altro_status PackTrajectoryDevice(double* X, double* U) = 0;
  virtual altro_status DeviceInfo(char* name, int name_len, int* cu_count) = 0;
  virtual const char* LastError() = 0;
};

// One factory per (dtype, model) translation unit (inst_*.hip); returns nullptr if dims mismatch.
EngineBase* MakeEngineUnicycleF64(const altro_desc& d, std::string* err);
EngineBase* MakeEngineUnicycleF32(const altro_desc& d, std::string* err);
EngineBase* MakeEngineTripleInt2F64(const altro_desc& d, std::string* err);
EngineBase* MakeEngineTripleInt2F32(const altro_desc& d, std::string* err);
EngineBase* MakeEngineQuad12F64(const altro_desc& d, std::string* err);
EngineBase* MakeEngineQuad12F32(const altro_desc& d, std::string* err);

}  // namespace altro_hip
