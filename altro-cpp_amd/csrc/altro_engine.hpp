// altro_engine.hpp — host orchestration of the batched AL-iLQR kernels for one (dtype, model).
//
// Owns the device state of one handle (one device, one HIP stream) and drives the sweeps:
//   AL solve  = k_al_init, k_solve_setup, k_rollout, then { k_expansions, k_backward, k_forward }
//               until no instance is active.  The host runs ONE SWEEP AHEAD of the device: sweep
//               i+1 is enqueued before the active-instance counter of sweep i is read back, so the
//               stream never drains (an all-idle sweep costs three empty launches).
// Instantiated once per translation unit (inst_*.hip).
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "altro_kernels.hpp"

namespace altro_hip {

#define ALTRO_HIP_CHECK(expr)                                                                  \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      char buf_[512];                                                                          \
      snprintf(buf_, sizeof(buf_), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,               \
               hipGetErrorString(e_));                                                         \
      err_ = buf_;                                                                             \
      return ALTRO_HIP_ERROR;                                                                  \
    }                                                                                          \
  } while (0)

inline DevOpts ToDevOpts(const altro_options& o) {
  DevOpts d{};
  d.max_iterations_total = o.max_iterations_total;
  d.max_iterations_outer = o.max_iterations_outer;
  d.max_iterations_inner = o.max_iterations_inner;
  d.bp_reg_fail_threshold = o.bp_reg_fail_threshold;
  d.check_forwardpass_bounds = o.check_forwardpass_bounds;
  d.line_search_max_iterations = o.line_search_max_iterations;
  d.reset_duals = o.reset_duals;
  d.cost_tolerance = o.cost_tolerance;
  d.gradient_tolerance = o.gradient_tolerance;
  d.bp_reg_increase_factor = o.bp_reg_increase_factor;
  d.bp_reg_initial = o.bp_reg_initial;
  d.bp_reg_max = o.bp_reg_max;
  d.bp_reg_min = o.bp_reg_min;
  d.state_max = o.state_max;
  d.control_max = o.control_max;
  d.line_search_lower_bound = o.line_search_lower_bound;
  d.line_search_upper_bound = o.line_search_upper_bound;
  d.line_search_decrease_factor = o.line_search_decrease_factor;
  d.constraint_tolerance = o.constraint_tolerance;
  d.maximum_penalty = o.maximum_penalty;
  d.initial_penalty = o.initial_penalty;
  return d;
}

template <class T, class M>
class Engine final : public EngineBase {
  static constexpr int n = M::n, m = M::m, nm = n + m;
  using R = Rec<T, n, m>;
  using RS = rec_scalar_t<T, M>;  // storage type of the expansion and gain records (float under WithRec32<>)
  using RR = Rec<RS, n, m>;

 public:
  explicit Engine(const altro_desc& d) : desc_(d) {}
  ~Engine() override { Release(); }
  const char* LastError() override { return err_.c_str(); }

  altro_status Init() {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    ALTRO_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    ALTRO_HIP_CHECK(hipEventCreateWithFlags(&start_ev_, hipEventDisableTiming));
    ALTRO_HIP_CHECK(hipEventCreateWithFlags(&tail_ev_, hipEventDisableTiming | hipEventBlockingSync));
    cur_ = stream_;
    ALTRO_HIP_CHECK(hipDeviceGetAttribute(&num_cus_, hipDeviceAttributeMultiprocessorCount, desc_.device_id));
    // (round 3, tail iteration at 39 us: hand-over at 1x .. 8x the CUs measured, 2x .. 4x best by ~1 %: workgroups beyond
    //  the CUs queue up behind the first round; from 6x on the persistent kernel loses to the batched sweeps)
    persist_at_ = 2 * num_cus_;
    fwd_single_at_ = 2 * num_cus_;  // (measured 1x .. 8x the CUs: config 3 -1.2 % at 2x, config 2 indifferent)
    if (const char* e = std::getenv("ALTRO_HIP_DEBUG_POISON")) {
      poison_on_ = true;
      poison_pattern_ = (unsigned)strtoul(e, nullptr, 16);
      poison_mix_ = std::strchr(e, ',') != nullptr;
    }
    if (const char* e = std::getenv("ALTRO_HIP_PERSIST_AT")) persist_at_ = atoi(e);
    return ReserveCounters(1024);
  }
  // one counter per sweep (device: filled by the forward kernel's atomics; host: pinned + mapped,
  // written by the next sweep's first kernel), so no memset / copy sits between the sweeps
  altro_status ReserveCounters(int count) {
    if (count <= counter_cap_) return ALTRO_OK;
    if (d_counter_) hipFree(d_counter_);
    if (h_counter_) hipHostFree((void*)h_counter_);
    d_counter_ = nullptr;
    h_counter_ = nullptr;
    counter_cap_ = 0;
    ALTRO_HIP_CHECK(hipHostMalloc((void**)&h_counter_, (size_t)count * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    ALTRO_HIP_CHECK(hipHostGetDevicePointer((void**)&h_counter_dev_, (void*)h_counter_, 0));
    ALTRO_HIP_CHECK(hipMalloc((void**)&d_counter_, (size_t)count * sizeof(int)));
    counter_cap_ = count;
    return ALTRO_OK;
  }

  // ---- problem upload -------------------------------------------------------------------------
  altro_status Upload(const ProblemSpec& spec, std::string* err) override {
    altro_status st;
    if (uploaded_) {
      err_ = "problem definition changed after the device state was created; create a new handle";
      st = ALTRO_NOT_READY;
    } else {
      st = UploadImpl(spec);
      // EVERY failed upload -- a rejected argument, a failed allocation, a HIP error half-way -- leaves the engine as it
      // was before it: buffers, chain streams and the chain booking of the partial upload go, so that the retry neither
      // leaks them nor meets its own ChainClaim
      if (st != ALTRO_OK) DropUpload();
    }
    if (st != ALTRO_OK && err) *err = err_;
    return st;
  }
  altro_status SetInitialState(const ProblemSpec& spec, std::string* err) override {
    ctg_replayable_ = false;  // (what a replayed backward pass would read no longer belongs to the last whole solve: ADVICE r5)
    altro_status st = SetInitialStateImpl(spec);
    if (st != ALTRO_OK && err) *err = err_;
    return st;
  }
  altro_status SetTrajectory(const ProblemSpec& spec, std::string* err) override {
    ctg_replayable_ = false;
    altro_status st = SetTrajectoryImpl(spec);
    if (st != ALTRO_OK && err) *err = err_;
    return st;
  }

  // The integration step belongs to the trajectory (trajectory.hpp:122-130), which the caller may replace
  // between solves: it can be set at any time, and the calls that integrate refuse to run without it.
  altro_status SetStep(float hstep) override {
    if (!uploaded_) return ALTRO_OK;  // Upload takes it from the spec
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    altro_status st = Sync();
    if (st != ALTRO_OK) return st;
    pd_.hstep = hstep;
    ALTRO_HIP_CHECK(CopySync(d_pd_, &pd_, sizeof(pd_), hipMemcpyHostToDevice));
    return ALTRO_OK;
  }
  // Trajectory::SetStep(k, h) / SetTime(k, t) (trajectory.hpp:119-120) and time-varying user models
  // (ContinuousDynamics::Evaluate(x, u, t, xdot), dynamics.hpp:59-95).  The hot kernels are built around one step for the
  // whole horizon (ProblemDesc::hstep: a loop invariant of the rollout wave, of the fused RK4 and of the persistent
  // kernel); a trajectory with its own steps or a model that reads the time takes the general path instead -- the
  // expansions / initial rollout read h[k], t[k] per knot, the forward pass runs on k_forward (one wave per three
  // instances, inputs from global memory) and the persistent tail kernel is not used.  Same schedule, same results as
  // the oracle (tests/test_knot_times_gpu.py); slower per iteration.
  altro_status SetKnotTimes(const ProblemSpec& s, std::string* err) override {
    altro_status st = SetKnotTimesImpl(s);
    if (st != ALTRO_OK && err) *err = err_;
    return st;
  }
  altro_status SetKnotTimesImpl(const ProblemSpec& s) {
    if (!uploaded_) return ALTRO_OK;  // Upload takes them from the spec
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    altro_status st = Sync();
    if (st != ALTRO_OK) return st;
    const bool per_knot = !s.hk.empty() || kTimeVarying;
    // (a time-varying model whose trajectory has no step yet -- the facade uploads the problem before it hands the
    //  trajectory over -- stays without knot times: StepOk() refuses every integrating call until altro_set_uniform_step /
    //  altro_set_steps come, and both end up here again)
    if (!per_knot || (s.hk.empty() && !(s.hstep > 0.0f))) {
      A_.hk = nullptr;
      A_.tk = nullptr;
      return ALTRO_OK;
    }
    std::vector<float> hk(N_ + 1, 0.0f), tk(N_ + 1, 0.0f);
    for (int k = 0; k < N_; ++k) hk[k] = s.hk.empty() ? s.hstep : s.hk[k];
    if (!s.tk.empty()) {
      for (int k = 0; k <= N_; ++k) tk[k] = s.tk[k];
    } else if (s.hk.empty()) {  // SetUniformStep (trajectory.hpp:122-130)
      for (int k = 0; k < N_; ++k) tk[k] = static_cast<float>(k) * s.hstep;
      tk[N_] = s.hstep * N_;
    }  // (steps without times: the times stay zero, as Trajectory::SetStep leaves them)
    for (int k = 0; k < N_; ++k)
      if (!(hk[k] > 0.0f)) {
        err_ = "the integration step of knot " + std::to_string(k) + " is not set (Trajectory::SetStep / SetUniformStep)";
        return ALTRO_NOT_READY;
      }
    if (!d_hk_) {
      altro_status as = Alloc(&d_hk_, (size_t)N_ + 1);
      if (as != ALTRO_OK) return as;
      as = Alloc(&d_tk_, (size_t)N_ + 1);
      if (as != ALTRO_OK) return as;
    }
    ALTRO_HIP_CHECK(CopySync(d_hk_, hk.data(), hk.size() * sizeof(float), hipMemcpyHostToDevice));
    ALTRO_HIP_CHECK(CopySync(d_tk_, tk.data(), tk.size() * sizeof(float), hipMemcpyHostToDevice));
    A_.hk = d_hk_;
    A_.tk = d_tk_;
    return ALTRO_OK;
  }
  altro_status ResetTrajectory() override {
    ctg_replayable_ = false;
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    ALTRO_HIP_CHECK(hipMemcpyAsync(A_.X, X_init_, (size_t)(N_ + 1) * R::nP * Bp_ * sizeof(T), hipMemcpyDeviceToDevice, stream_));
    ALTRO_HIP_CHECK(hipMemcpyAsync(A_.U, U_init_, (size_t)N_ * R::mP * Bp_ * sizeof(T), hipMemcpyDeviceToDevice, stream_));
    return ALTRO_OK;
  }
  altro_status ResetStats() override {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    hipLaunchKernelGGL(k_reset_stats<T>, GridB(), dim3(kBlock), 0, stream_, A_);
    return ALTRO_OK;
  }
  altro_status SetPenalty(double rho) override {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    hipLaunchKernelGGL(k_set_rows<T>, GridB(), dim3(kBlock), 0, stream_, A_, d_pd_, 0, 1, T(rho));
    return Sync();
  }
  altro_status SetPenaltyScaling(double phi) override {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    std::vector<double> v(kMaxClasses * kMaxConPerKnot, phi);
    ALTRO_HIP_CHECK(CopySync(d_phi_, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
    return ALTRO_OK;
  }

  // ---- solves -----------------------------------------------------------------------------------
  altro_status SolveAL(const altro_options& o) override { return Solve(o, kFwdAL); }
  altro_status SolveILQR(const altro_options& o) override { return Solve(o, kFwdILQR); }

  altro_status AlInit(const altro_options& o) override {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    const DevOpts d = ToDevOpts(o);
    hipLaunchKernelGGL(k_al_init<T>, GridAlInit(), dim3(kBlock), 0, stream_, A_, d_pd_, d);
    hipLaunchKernelGGL((k_knot_costs<T, M>), GridBK(), dim3(kBlock), 0, stream_, A_, d_pd_);
    hipLaunchKernelGGL(k_log_viol_pen<T>, GridB(), dim3(kBlock), 0, stream_, A_, d_pd_);
    return Sync();
  }
  altro_status SolveSetup(const altro_options& o) override {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    hipLaunchKernelGGL(k_solve_setup<T>, GridB(), dim3(kBlock), 0, stream_, A_, ToDevOpts(o), 0);
    return Sync();
  }
  altro_status Rollout(const altro_options&) override {
    if (!StepOk()) return ALTRO_NOT_READY;
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    hipLaunchKernelGGL((k_rollout<T, M>), GridB(), dim3(kBlock), 0, stream_, A_, d_pd_, 1);
    return Sync();
  }
  altro_status Cost(const altro_options&, double* J) override {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    hipLaunchKernelGGL((k_knot_costs<T, M>), GridBK(), dim3(kBlock), 0, stream_, A_, d_pd_);
    hipLaunchKernelGGL(k_sum_costs<T>, GridB(), dim3(kBlock), 0, stream_, A_, d_tmp_);
    altro_status st = Sync();
    if (st != ALTRO_OK) return st;
    if (J) return DownloadVec(d_tmp_, J);
    return ALTRO_OK;
  }
  altro_status UpdateExpansions(const altro_options&) override {
    ctg_replayable_ = false;  // (the records a replayed backward pass would read are about to change)
    if (!StepOk()) return ALTRO_NOT_READY;
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    hipLaunchKernelGGL((k_expansions<T, M>), GridBK(), dim3(kBlock), 0, stream_, A_, d_pd_, 1, (int*)nullptr, (int*)nullptr);
    return Sync();
  }
  altro_status BackwardPass(const altro_options& o) override {
    cur_ = stream_;
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    LaunchBackward(A_, ToDevOpts(o), 1, B_);
    ctg_fresh_ = A_.record_ctg != 0;
    ctg_replayable_ = false;
    return Sync();
  }
  // the backward pass of the last iteration of a whole solve, once more, with the cost-to-go records on (see GetCtg)
  altro_status ReplayCtg() {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    altro_status st = Sync();
    if (st != ALTRO_OK) return st;
    double* tmpT = nullptr;
    int* tmpI = nullptr;
    const size_t nT = (size_t)kNumScalarT * Bp_ * sizeof(double), nI = (size_t)kNumScalarI * Bp_ * sizeof(int);
    ALTRO_HIP_CHECK(hipMalloc((void**)&tmpT, nT));
    if (hipMalloc((void**)&tmpI, nI) != hipSuccess) {
      hipFree(tmpT);
      err_ = "ReplayCtg: out of device memory";
      return ALTRO_HIP_ERROR;
    }
    hipError_t e = hipMemcpyAsync(tmpT, d_scalarT_, nT, hipMemcpyDeviceToDevice, stream_);
    if (e == hipSuccess) e = hipMemcpyAsync(tmpI, d_scalarI_, nI, hipMemcpyDeviceToDevice, stream_);
    // the regularisation the last backward pass used (stats_.Log("reg", rho_)) is what this one starts from
    if (e == hipSuccess) e = hipMemcpyAsync(A_.rho_reg, A_.reg_log, (size_t)Bp_ * sizeof(double), hipMemcpyDeviceToDevice, stream_);
    if (e == hipSuccess) {
      DevArrays<T> A = A_;
      A.record_ctg = 1;
      cur_ = stream_;
      LaunchBackward(A, last_opts_, 1, B_);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(d_scalarT_, tmpT, nT, hipMemcpyDeviceToDevice, stream_);
    if (e == hipSuccess) e = hipMemcpyAsync(d_scalarI_, tmpI, nI, hipMemcpyDeviceToDevice, stream_);
    if (e == hipSuccess) e = hipStreamSynchronize(stream_);
    hipFree(tmpT);
    hipFree(tmpI);
    ALTRO_HIP_CHECK(e);
    ctg_fresh_ = true;
    return ALTRO_OK;
  }
  altro_status ForwardPass(const altro_options& o) override {
    cur_ = stream_;
    ctg_replayable_ = false;  // (a step-level forward pass moves the trajectory)
    if (!StepOk()) return ALTRO_NOT_READY;
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    LaunchForward(A_, ToDevOpts(o), (int)kFwdStepOnly, 1, B_);
    return Sync();
  }
  altro_status UpdateConvergenceStatistics(const altro_options& o) override {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    hipLaunchKernelGGL((k_conv_stats<T, M>), GridB(), dim3(kBlock), 0, stream_, A_, d_pd_, ToDevOpts(o));
    return Sync();
  }
  altro_status UpdateDuals(const altro_options&) override {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    hipLaunchKernelGGL(k_update_duals<T>, GridB(), dim3(kBlock), 0, stream_, A_, d_pd_);
    return Sync();
  }
  altro_status UpdatePenalties(const altro_options&) override {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    hipLaunchKernelGGL(k_update_penalties<T>, GridB(), dim3(kBlock), 0, stream_, A_, d_pd_);
    return Sync();
  }
  altro_status GetMaxViolation(double* out) override {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    hipLaunchKernelGGL(k_max_viol_pen<T>, GridB(), dim3(kBlock), 0, stream_, A_, d_pd_, d_tmp_, (double*)nullptr);
    altro_status st = Sync();
    if (st != ALTRO_OK) return st;
    return DownloadVec(d_tmp_, out);
  }
  altro_status GetMaxPenalty(double* out) override {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    hipLaunchKernelGGL(k_max_viol_pen<T>, GridB(), dim3(kBlock), 0, stream_, A_, d_pd_, (double*)nullptr, d_tmp_);
    altro_status st = Sync();
    if (st != ALTRO_OK) return st;
    return DownloadVec(d_tmp_, out);
  }

  // ---- results ----------------------------------------------------------------------------------
  altro_status GetTrajectory(double* X, double* U) override {
    if (X) {
      altro_status st = DownloadRec(A_.X, N_ + 1, R::nP, 0, n, X);
      if (st != ALTRO_OK) return st;
    }
    if (U) return DownloadRec(A_.U, N_, R::mP, 0, m, U);
    return ALTRO_OK;
  }
  altro_status GetGains(double* K, double* d) override {
    if (K) {
      altro_status st = DownloadRec((const RS*)A_.KD, N_, RR::KP, RR::oK, m * n, K);
      if (st != ALTRO_OK) return st;
    }
    if (d) return DownloadRec((const RS*)A_.KD, N_, RR::KP, RR::oD, m, d);
    return ALTRO_OK;
  }
  altro_status SetRecordCtg(int enable) override {
    A_.record_ctg = enable ? 1 : 0;
    return ALTRO_OK;
  }
  // KnotPointFunctions::GetCostToGoHessian / Gradient (knot_point_function_type.hpp:243-268): P, p of the LAST backward
  // pass.  The solve itself keeps them in registers unless altro_set_record_ctg asked for the records (the persistent kernel
  // never stores them).  The reference's stay readable after Solve(); here a read behind a solve that did not record runs
  // that backward pass ONCE MORE with the recording on -- the expansions of the last iteration are still in memory, the
  // regularisation it used is logged (reg_log) -- and puts the solver state back (ReplayCtg): same kernels on the same
  // inputs, the same P, p.
  altro_status GetCtg(double* P, double* p) override {
    if (!ctg_fresh_) {
      if (!ctg_replayable_) {
        err_ = "no cost-to-go to read: it belongs to a backward pass -- altro_solve_*, or altro_backward_pass after "
               "altro_set_record_ctg(h, 1) -- and the trajectory or the expansions have changed since the last one";
        return ALTRO_NOT_READY;
      }
      const altro_status rs = ReplayCtg();
      if (rs != ALTRO_OK) return rs;
    }
    if (P) {
      altro_status st = DownloadRec(A_.CTG, N_ + 1, R::CP, R::oP, n * n, P);
      if (st != ALTRO_OK) return st;
    }
    if (p) return DownloadRec(A_.CTG, N_ + 1, R::CP, R::op, n, p);
    return ALTRO_OK;
  }
  altro_status GetExpansion(int k, double* AB, double* lxx, double* lxu, double* luu, double* lx,
                            double* lu) override {
    if (k < 0 || k > N_) return ALTRO_INVALID_ARG;
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    altro_status sst = Sync();
    if (sst != ALTRO_OK) return sst;
    std::vector<RS> h((size_t)RR::EP * Bp_);
    ALTRO_HIP_CHECK(CopySync(h.data(), (const RS*)A_.EXP + (size_t)k * Bp_ * RR::EP, h.size() * sizeof(RS), hipMemcpyDeviceToHost));
    auto take = [&](double* out, int off, int E, bool stage_only) {
      if (!out || (stage_only && k >= N_)) return;
      for (int b = 0; b < B_; ++b)
        for (int e = 0; e < E; ++e) out[(size_t)b * E + e] = (double)h[(size_t)b * RR::EP + off + e];
    };
    take(AB, R::oAB, n * nm, true);
    take(lxx, R::oLxx, n * n, false);
    take(lxu, R::oLxu, n * m, true);
    take(luu, R::oLuu, m * m, true);
    take(lx, R::oLx, n, false);
    take(lu, R::oLu, m, true);
    return ALTRO_OK;
  }
  altro_status GetKnotCosts(double* costs) override { return DownloadRec(A_.costs, N_ + 1, 1, 0, 1, costs); }
  int NumRows() override { return pd_.total_rows; }
  int NumRowsAt(int k) override {
    if (k < 0 || k > N_) return -1;
    return pd_.cls[knot_class_[k]].nrows;
  }
  altro_status GetRows(int which, double* out) override {
    T* src = which == 0 ? A_.lam : (which == 1 ? A_.pen : A_.cval);
    const int R = pd_.total_rows;
    if (R == 0) return ALTRO_OK;
    std::vector<T> h((size_t)R * Bp_);
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    {
      altro_status st = Sync();
      if (st != ALTRO_OK) return st;
    }
    ALTRO_HIP_CHECK(CopySync(h.data(), src, h.size() * sizeof(T), hipMemcpyDeviceToHost));
    for (int b = 0; b < B_; ++b)
      for (int r = 0; r < R; ++r) out[(size_t)b * R + r] = (double)h[(size_t)r * Bp_ + b];
    return ALTRO_OK;
  }
  altro_status SetDuals(const double* lam) override {
    ctg_replayable_ = false;
    const int R = pd_.total_rows;
    if (R == 0) return ALTRO_OK;
    std::vector<T> h((size_t)R * Bp_, T(0));
    for (int b = 0; b < B_; ++b)
      for (int r = 0; r < R; ++r) h[(size_t)r * Bp_ + b] = T(lam[(size_t)b * R + r]);
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    {
      altro_status st = Sync();
      if (st != ALTRO_OK) return st;
    }
    ALTRO_HIP_CHECK(CopySync(A_.lam, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return ALTRO_OK;
  }
  altro_status GetStats(altro_stats* st, bool ilqr_mode) override {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    {
      altro_status sst = Sync();
      if (sst != ALTRO_OK) return sst;
    }
    std::vector<double> f((size_t)kNumScalarT * Bp_);
    std::vector<int> iv((size_t)kNumScalarI * Bp_);
    ALTRO_HIP_CHECK(CopySync(f.data(), d_scalarT_, f.size() * sizeof(double), hipMemcpyDeviceToHost));
    ALTRO_HIP_CHECK(CopySync(iv.data(), d_scalarI_, iv.size() * sizeof(int), hipMemcpyDeviceToHost));
    auto F = [&](double* p, int b) { return f[(size_t)(p - d_scalarT_) + b]; };
    auto I = [&](int* p, int b) { return iv[(size_t)(p - d_scalarI_) + b]; };
    for (int b = 0; b < B_; ++b) {
      altro_stats& s = st[b];
      s.status_ilqr = I(A_.status, b);
      s.status = ilqr_mode ? s.status_ilqr : I(A_.status_al, b);
      s.iterations_inner = I(A_.it_inner, b);
      s.iterations_outer = I(A_.it_outer, b);
      s.iterations_total = I(A_.it_total, b);
      s.reserved = 0;
      s.cost = F(A_.cost_cur, b);
      s.initial_cost = F(A_.initial_cost, b);
      s.cost_decrease = F(A_.dJ, b);
      s.gradient = F(A_.grad, b);
      s.violation = F(A_.viol, b);
      s.max_penalty = F(A_.penmax, b);
      s.alpha = F(A_.alpha, b);
      s.regularization = F(A_.reg_log, b);
      s.improvement_ratio = F(A_.z, b);
    }
    return ALTRO_OK;
  }
  altro_status GetTiming(altro_timing* t) override {
    if (timing_.instance_iterations < 0 && uploaded_) {
      ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
      altro_status sst = Sync();
      if (sst != ALTRO_OK) return sst;
      std::vector<int> it(Bp_);
      ALTRO_HIP_CHECK(CopySync(it.data(), A_.it_total, (size_t)Bp_ * sizeof(int), hipMemcpyDeviceToHost));
      long long tot = 0;
      for (int b = 0; b < B_; ++b) tot += it[b];
      timing_.instance_iterations = tot;
    }
    *t = timing_;
    return ALTRO_OK;
  }
  altro_status SetRecordHistory(int capacity) override {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    if (A_.hist) {
      hipFree(A_.hist);
      hipFree(A_.hist_len);
      A_.hist = nullptr;
      A_.hist_len = nullptr;
    }
    A_.hist_cap = 0;
    if (capacity > 0) {
      ALTRO_HIP_CHECK(hipMalloc((void**)&A_.hist, (size_t)kHistFields * capacity * Bp_ * sizeof(double)));
      ALTRO_HIP_CHECK(hipMalloc((void**)&A_.hist_len, (size_t)Bp_ * sizeof(int)));
      ALTRO_HIP_CHECK(ZeroSync(A_.hist_len, 0, (size_t)Bp_ * sizeof(int)));
      A_.hist_cap = capacity;
    }
    return ALTRO_OK;
  }
  int GetHistory(int instance, int field, double* out, int cap) override {
    if (!A_.hist || instance < 0 || instance >= B_ || field < 0 || field >= kHistFields) return -1;
    if (hipSetDevice(desc_.device_id) != hipSuccess) return -1;
    if (hipStreamSynchronize(stream_) != hipSuccess) return -1;
    int len = 0;
    if (CopySync(&len, A_.hist_len + instance, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    // the reference's vectors also hold the row opened by the last NewIteration: a copy of the last
    const int cnt = std::min(std::min(len, A_.hist_cap), cap);
    if (cnt <= 0) return 0;
    // column `instance` of the [field][row][Bp] block: one strided copy
    if (hipMemcpy2DAsync(out, sizeof(double), A_.hist + (size_t)field * A_.hist_cap * Bp_ + instance, (size_t)Bp_ * sizeof(double),
                         sizeof(double), (size_t)cnt, hipMemcpyDeviceToHost, stream_) != hipSuccess ||
        hipStreamSynchronize(stream_) != hipSuccess)
      return -1;
    return cnt;
  }
  // every field of one instance's history at once: one synchronisation in front, the eight column copies enqueued back
  // to back, one synchronisation behind (the per-field call costs three per field)
  int GetHistoryAll(int instance, double* out, int cap) override {
    if (!A_.hist || instance < 0 || instance >= B_ || cap <= 0) return -1;
    if (hipSetDevice(desc_.device_id) != hipSuccess) return -1;
    int len = 0;
    if (CopySync(&len, A_.hist_len + instance, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    const int cnt = std::min(std::min(len, A_.hist_cap), cap);
    if (cnt <= 0) return 0;
    for (int f = 0; f < kHistFields; ++f)
      if (hipMemcpy2DAsync(out + (size_t)f * cap, sizeof(double), A_.hist + (size_t)f * A_.hist_cap * Bp_ + instance,
                           (size_t)Bp_ * sizeof(double), sizeof(double), (size_t)cnt, hipMemcpyDeviceToHost, stream_) != hipSuccess)
        return -1;
    if (hipStreamSynchronize(stream_) != hipSuccess) return -1;
    return cnt;
  }
  altro_status DeviceInfo(char* name, int name_len, int* cu_count) override {
    hipDeviceProp_t p;
    ALTRO_HIP_CHECK(hipGetDeviceProperties(&p, desc_.device_id));
    if (name && name_len > 0) {
      std::strncpy(name, p.name, name_len - 1);
      name[name_len - 1] = 0;
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    return ALTRO_OK;
  }
  altro_status PackResultsDevice(void* dst) override {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    hipLaunchKernelGGL(k_pack_results<T>, GridB(), dim3(kBlock), 0, stream_, A_, (double*)dst, last_mode_ilqr_ ? 1 : 0);
    return Sync();
  }
  // records [knots][Bp][nP] -> the caller's rows [B][knots][n] straight into the caller's DEVICE buffers
  altro_status PackTrajectoryDevice(double* X, double* U) override {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    if (X)
      hipLaunchKernelGGL((k_rec_to_rows<T>), dim3((B_ + kBlock - 1) / kBlock, N_ + 1), dim3(kBlock), 0, stream_, (const T*)A_.X, X,
                         N_ + 1, R::nP, 0, n, B_, Bp_);
    if (U)
      hipLaunchKernelGGL((k_rec_to_rows<T>), dim3((B_ + kBlock - 1) / kBlock, N_), dim3(kBlock), 0, stream_, (const T*)A_.U, U, N_,
                         R::mP, 0, m, B_, Bp_);
    return Sync();
  }

 private:
  // ---- helpers ----------------------------------------------------------------------------------
  dim3 GridB() const { return dim3((B_ + kBlock - 1) / kBlock); }
  dim3 GridBK() const { return dim3((B_ + kBlock - 1) / kBlock, N_ + 1); }
  dim3 GridAlInit() const { return dim3((B_ + kBlock - 1) / kBlock, std::max(1, (pd_.total_rows + kAlInitRows - 1) / kAlInitRows)); }
  // Backward pass launch: fp64 unicycle-sized problems run on the matrix cores (4 instances per
  // wavefront), everything else on the one-lane-per-instance VALU kernel.
  // The MFMA backward pass computes in fp64 whatever the storage type of the engine is
  static constexpr bool kMfmaBackward = n <= 3 && m <= 2;  // (round 4: every model that fits the 4 x 4 tiles, not only the unicycle's shape)
  // larger models: one instance per wavefront -- on the 16x16x4 fp64 matrix cores (k_backward_mfma16), or with
  // the matrices in LDS and the products on the vector ALUs (k_backward_coop: ALTRO_HIP_BACKWARD=coop)
  // (n = 4 -- the cart-pole class of user models -- takes the 16x16 tile too while the launch has few instances: one
  //  instance per wavefront at ~0.4 us per knot against ~1.5 us for the one-lane-per-instance kernel, which only wins on
  //  throughput once the instances outnumber the SIMDs several times: kMfma16SmallMax.  Decided by the handle's batch,
  //  not by the instances still active, so that an instance sees the same arithmetic in every sweep of every solve)
  static constexpr bool kMfma16Backward = n >= 4 && n <= 12 && m <= 4;
  static constexpr int kMfma16SmallMax = 4096;
  // gain records big enough (m n + m >= 14 elements) that reading K from global memory can pay: see kdg_
  static constexpr bool kKdgEligible = n * m >= 12;
  // rollout inputs from global memory, two knots ahead in three register sets: small records only
  // (round 6: ... and the large models, for which it competes with kSrcKdg -- see the LDS plan in UploadImpl)
  static constexpr bool kRgEligible = (!kKdgEligible && R::KP + R::nP + R::mP <= 16) || kKdgEligible;
  static constexpr bool kCoopBackward = !kMfmaBackward && n >= 6;
  // ALTRO_HIP_DEBUG_POISON=<hex pattern>[,mix]: before every kernel of a solve, fill the LDS of every CU with the pattern
  // (and the line-search candidates once, at upload), to flush out reads of memory the solve has not written.
  void PoisonLds() {
    if (!poison_on_) return;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_poison_lds<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((k_poison_lds<0>), dim3(num_cus_ * 8), dim3(256), 160 * 1024, cur_, poison_pattern_, 160 * 1024 / 4,
                       poison_mix_, (int*)nullptr);
  }
  void LaunchBackward(const DevArrays<T>& A, const DevOpts& d, int all, int ninst) {
    PoisonLds();
    if constexpr (kMfmaBackward) {
      if (!force_valu_backward_ && mfma_offsets_ok_) {
        if (A.record_ctg)
          hipLaunchKernelGGL((k_backward_mfma<T, M, true>), dim3((ninst + 3) / 4), dim3(kBlock), 0, cur_, A, d, all);
        else
          hipLaunchKernelGGL((k_backward_mfma<T, M, false>), dim3((ninst + 3) / 4), dim3(kBlock), 0, cur_, A, d, all);
        return;
      }
    }
    if constexpr (kMfma16Backward) {
      if (!force_valu_backward_ && !force_coop_backward_ && mfma_offsets_ok_ && (n > 4 || B_ <= kMfma16SmallMax)) {
        if (A.record_ctg)
          hipLaunchKernelGGL((k_backward_mfma16<T, M, true>), dim3(ninst), dim3(kBlock), 0, cur_, A, d, all);
        else
          hipLaunchKernelGGL((k_backward_mfma16<T, M, false>), dim3(ninst), dim3(kBlock), 0, cur_, A, d, all);
        return;
      }
    }
    if constexpr (kCoopBackward) {
      if (!force_valu_backward_) {
        hipLaunchKernelGGL((k_backward_coop<T, M>), dim3(ninst), dim3(kBlock), 0, cur_, A, d, all);
        return;
      }
    }
    hipLaunchKernelGGL((k_backward<T, M>), dim3((ninst + kBlock - 1) / kBlock), dim3(kBlock), 0, cur_, A, d, all);
  }
  // The fused sweep kernel needs the MFMA backward pass, the LDS-staged forward pass with one instance
  // per workgroup, no cost-to-go recording and at most 20 line-search trials.
  bool FusedOk(const DevOpts& d) const {
    if constexpr (!kMfmaBackward) return false;
    return !force_valu_backward_ && !no_fused_ && mfma_offsets_ok_ && fwd_lds_bytes_ > 0 && !A_.record_ctg && !A_.hk &&
           d.line_search_max_iterations <= kLineSearchLanes && fused_lds_bytes_ <= 160 * 1024;
  }
  // Forward pass launch: instances per wavefront and the LDS-staged variant are chosen from the size
  // of one instance's read-only block (X, U, K, d, lambda, rho); see k_forward.
  void LaunchForward(const DevArrays<T>& A, const DevOpts& d, int mode, int all, int ninst, int ninst_all_chains = -1) {
    PoisonLds();
    // (per-knot steps / times / models: the three-wave kernel takes them for every model without a hand-fused RK4 --
    //  user models, the triple integrator, the 12-state model; the unicycle's fused rollout keeps ONE step as a loop
    //  invariant and goes to k_forward)
    if (fwd_lds_bytes_ > 0 && d.line_search_max_iterations <= kLineSearchLanes && (!A.hk || !M::kHasFusedRk4)) {
      // two-wave pipeline (rollout wave + cost wave), inputs staged in LDS.  When the instances left
      // would not even fill the CUs one by one, each gets a workgroup of its own: the prologue and the
      // epilogue of the kernel (staging, winner copy) shrink with the instances per workgroup.
      const int per_wave = (std::max(ninst, ninst_all_chains) <= fwd_single_at_) ? 1 : fwd_per_wave_;
      const size_t lds = fwd_shared_bytes_ + (size_t)per_wave * fwd_per_inst_bytes_;
      const dim3 grid2((ninst + per_wave - 1) / per_wave);
      if (kdg_) {
        if constexpr (kKdgEligible)
          hipLaunchKernelGGL((k_forward2<T, M, kSrcKdg>), grid2, dim3(kFwdWaves * kBlock), lds, cur_, A, d_pd_, pd_, d, mode,
                             all, per_wave);
      } else if (rg_) {
        if constexpr (kRgEligible)
          hipLaunchKernelGGL((k_forward2<T, M, kSrcGlb>), grid2, dim3(kFwdWaves * kBlock), lds, cur_, A, d_pd_, pd_, d, mode,
                             all, per_wave);
      } else {
        hipLaunchKernelGGL((k_forward2<T, M, kSrcLds>), grid2, dim3(kFwdWaves * kBlock), lds, cur_, A, d_pd_, pd_, d, mode,
                           all, per_wave);
      }
      return;
    }
    const dim3 grid((ninst + fwd_per_wave_ - 1) / fwd_per_wave_);
    {
      // fallback: single wave, reads from HBM (staged block larger than LDS, > 20 line-search trials, or per-knot
      // steps / times: SetKnotTimes)
      hipLaunchKernelGGL((k_forward<T, M>), grid, dim3(kBlock), 0, cur_, A, d_pd_, d, mode, all, fwd_per_wave_);
    }
  }
  static void CpuRelax() {  // one spin-wait hint, whatever the host CPU
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield");
#else
    std::this_thread::yield();
#endif
  }
  bool StepOk() {
    if (pd_.hstep > 0.0f || A_.hk) return true;
    err_ = "the integration step is not set (altro_set_uniform_step / Trajectory::SetUniformStep)";
    return false;
  }
  altro_status Sync() {
    ALTRO_HIP_CHECK(hipGetLastError());
    ALTRO_HIP_CHECK(hipStreamSynchronize(stream_));
    return ALTRO_OK;
  }
  // Every copy and fill of the engine runs on the engine's own stream and is waited for: that stream is non-blocking, so
  // work on the null stream (where a plain hipMemcpy / hipMemset goes -- and a device-to-device hipMemcpy or a hipMemset
  // may return before it has run) would not be ordered against the kernels and the asynchronous copies of the solves.
  hipError_t CopySync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, stream_);
    return e != hipSuccess ? e : hipStreamSynchronize(stream_);
  }
  hipError_t ZeroSync(void* dst, int value, size_t bytes) {
    const hipError_t e = hipMemsetAsync(dst, value, bytes, stream_);
    return e != hipSuccess ? e : hipStreamSynchronize(stream_);
  }
  template <class U>
  altro_status Alloc(U** p, size_t count) {
    ALTRO_HIP_CHECK(hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(U)));
    ALTRO_HIP_CHECK(ZeroSync(*p, 0, std::max<size_t>(count, 1) * sizeof(U)));
    allocs_.push_back((void*)*p);
    return ALTRO_OK;
  }
  // undo a (partial) UploadImpl: device buffers, chain streams and the chain booking go; the engine's own stream stays
  void DropUpload() {
    uploaded_ = false;
    if (hipSetDevice(desc_.device_id) != hipSuccess) return;
    hipStreamSynchronize(stream_);
    for (void* p : allocs_) hipFree(p);
    allocs_.clear();
    d_hk_ = d_tk_ = nullptr;
    if (counted_chained_) ChainClaim(desc_.device_id, -1);
    counted_chained_ = false;
    for (int c = 1; c < kMaxChains; ++c) {
      if (chain_stream_[c]) hipStreamDestroy(chain_stream_[c]);
      if (chain_ev_[c]) hipEventDestroy(chain_ev_[c]);
      chain_stream_[c] = nullptr;
      chain_ev_[c] = nullptr;
    }
    chains_ = 1;
    // (the history buffers and the recording switches belong to the handle, not to the upload)
    double* const hist = A_.hist;
    int* const hist_len = A_.hist_len;
    const int hist_cap = A_.hist_cap, record_ctg = A_.record_ctg;
    std::memset(&A_, 0, sizeof(A_));
    A_.hist = hist;
    A_.hist_len = hist_len;
    A_.hist_cap = hist_cap;
    A_.record_ctg = record_ctg;
    d_tmp_ = nullptr;
    d_twin_box_ = nullptr;
    twin_cap_ = 0;
    d_seg_cursor_ = nullptr;
    seg_total_ = 0;
    d_list_[0] = d_list_[1] = nullptr;
    d_iota_ = d_merged_ = nullptr;
    d_loop_win_ = d_loop_ctl_ = d_loop_tail_ = nullptr;
    loop_groups_ = 0;
    X_init_ = U_init_ = nullptr;
    d_phi_ = nullptr;
    d_pd_ = nullptr;
    d_scalarT_ = nullptr;
    d_scalarI_ = nullptr;
  }
  void Release() {
    if (hipSetDevice(desc_.device_id) != hipSuccess) return;
    for (void* p : allocs_) hipFree(p);
    allocs_.clear();
    if (A_.hist) hipFree(A_.hist);
    if (A_.hist_len) hipFree(A_.hist_len);
    if (d_stage_) hipFree(d_stage_);
    d_stage_ = nullptr;
    stage_cap_ = 0;
    if (d_counter_) hipFree(d_counter_);
    if (h_counter_) hipHostFree((void*)h_counter_);
    for (auto& e : prof_ev_) hipEventDestroy(e);
    if (counted_chained_) ChainClaim(desc_.device_id, -1);
    counted_chained_ = false;
    for (int c = 1; c < kMaxChains; ++c) {
      if (chain_stream_[c]) hipStreamDestroy(chain_stream_[c]);
      if (chain_ev_[c]) hipEventDestroy(chain_ev_[c]);
      chain_stream_[c] = nullptr;
      chain_ev_[c] = nullptr;
    }
    if (start_ev_) hipEventDestroy(start_ev_);
    if (tail_ev_) hipEventDestroy(tail_ev_);
    if (stream_) hipStreamDestroy(stream_);
  }
  altro_status DownloadVec(const double* dev, double* out) {
    ALTRO_HIP_CHECK(CopySync(out, dev, (size_t)B_ * sizeof(double), hipMemcpyDeviceToHost));
    return ALTRO_OK;
  }
  // Staging buffer of the host boundary: the caller's row layout on the device (see k_rec_to_rows / k_rows_to_rec).
  altro_status EnsureStage(size_t doubles) {
    if (doubles <= stage_cap_) return ALTRO_OK;
    if (d_stage_) ALTRO_HIP_CHECK(hipFree(d_stage_));
    d_stage_ = nullptr;
    stage_cap_ = 0;
    ALTRO_HIP_CHECK(hipMalloc((void**)&d_stage_, doubles * sizeof(double)));
    stage_cap_ = doubles;
    return ALTRO_OK;
  }
  // device records [knots][Bp][EP] (fields at off..off+E) -> host [B][knots][E]: converted on the device, one contiguous
  // copy straight into the caller's buffer
  template <class E_>
  altro_status DownloadRec(const E_* dev, int knots, int EP, int off, int E, double* out) {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    const size_t count = (size_t)B_ * knots * E;
    altro_status sst = EnsureStage(count);
    if (sst != ALTRO_OK) return sst;
    hipLaunchKernelGGL((k_rec_to_rows<E_>), dim3((B_ + kBlock - 1) / kBlock, knots), dim3(kBlock), 0, stream_, dev, d_stage_, knots,
                       EP, off, E, B_, Bp_);
    ALTRO_HIP_CHECK(hipGetLastError());
    ALTRO_HIP_CHECK(CopySync(out, d_stage_, count * sizeof(double), hipMemcpyDeviceToHost));
    return ALTRO_OK;
  }
  // host [B][knots][E] (or shared [knots][E]; nullptr: zeros) -> device records [knots][Bp][EP] (padding zeroed)
  altro_status UploadRec(T* dev, int knots, int EP, int E, const double* src, bool per_instance) {
    const size_t count = (size_t)(per_instance ? B_ : 1) * knots * E;
    if (src) {
      altro_status sst = EnsureStage(count);
      if (sst != ALTRO_OK) return sst;
      ALTRO_HIP_CHECK(hipMemcpyAsync(d_stage_, src, count * sizeof(double), hipMemcpyHostToDevice, stream_));
    }
    hipLaunchKernelGGL((k_rows_to_rec<T>), dim3(Bp_ / kBlock, knots), dim3(kBlock), 0, stream_, src ? (const double*)d_stage_ : nullptr,
                       dev, knots, EP, E, B_, Bp_, per_instance ? 1 : 0);
    ALTRO_HIP_CHECK(hipGetLastError());
    return Sync();
  }

  altro_status SetInitialStateImpl(const ProblemSpec& s) {
    if (!uploaded_) return ALTRO_OK;
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    {
      altro_status st = Sync();
      if (st != ALTRO_OK) return st;
    }
    // x0 records [b][nP]: one "knot" with Bp instances
    if (s.x0.empty()) return UploadRec(A_.x0, 1, R::nP, n, nullptr, false);
    return UploadRec(A_.x0, 1, R::nP, n, s.x0.data(), s.x0_per_instance != 0);
  }
  altro_status SetTrajectoryImpl(const ProblemSpec& s) {
    if (!uploaded_) return ALTRO_OK;
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    {
      altro_status st0 = Sync();  // ResetTrajectory may still be copying on the (non-blocking) stream
      if (st0 != ALTRO_OK) return st0;
    }
    const double* Xh = !s.has_X ? nullptr : (s.X_view ? s.X_view : s.X.data());
    const double* Uh = !s.has_U ? nullptr : (s.U_view ? s.U_view : s.U.data());
    altro_status st = UploadRec(A_.X, N_ + 1, R::nP, n, Xh, s.traj_per_instance != 0);
    if (st != ALTRO_OK) return st;
    st = UploadRec(A_.U, N_, R::mP, m, Uh, s.traj_per_instance != 0);
    if (st != ALTRO_OK) return st;
    ALTRO_HIP_CHECK(CopySync(X_init_, A_.X, (size_t)(N_ + 1) * R::nP * Bp_ * sizeof(T), hipMemcpyDeviceToDevice));
    ALTRO_HIP_CHECK(CopySync(U_init_, A_.U, (size_t)N_ * R::mP * Bp_ * sizeof(T), hipMemcpyDeviceToDevice));
    return ALTRO_OK;
  }

  // Build the device problem description from the recorded setter calls.
  altro_status UploadImpl(const ProblemSpec& s) {
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    // per-knot models (Problem::SetDynamics(model, k), problem.hpp:155-166): indices into the source's ALTRO_USER_MODELS,
    // checked before anything is allocated
    for (size_t k = 0; k < s.knot_model.size() && (int)k < desc_.N; ++k) {
      const int w = s.knot_model[k];
      if (w < 0 || w >= kModelCount) {
        err_ = "altro_set_knot_models: knot " + std::to_string(k) + " asks for model " + std::to_string(w) + ", this handle's model " +
               (kModelCount > 1 ? "source lists " + std::to_string(kModelCount) + " models (ALTRO_USER_MODELS)"
                                : std::string("is a single model (a user source with #define ALTRO_USER_MODELS A, B, ... holds several)"));
        return ALTRO_INVALID_ARG;
      }
    }
    B_ = desc_.batch;
    N_ = desc_.N;
    Bp_ = ((B_ + kBlock - 1) / kBlock) * kBlock;
    // Twin workgroups of the persistent kernel (TwinCtl, altro_kernels.hpp) clone an instance into a SHADOW COLUMN of the
    // per-instance arrays: every array is allocated twin_cap_ instances wider (the stride of all of them is Bp_), one column
    // per slot of a persistent launch.  ALTRO_HIP_TWIN=0: no columns, no twins.
    twin_cap_ = 0;
    if constexpr (kMfmaBackward) {
      const char* e = std::getenv("ALTRO_HIP_TWIN");
      if (!(e && atoi(e) == 0) && !fast_forward_ && !no_fused_) {
        // (a solve that has split rejection streaks hands over at 3/2 of persist_at_, see Solve)
        const int want = std::min(Bp_, ((persist_at_ * 3 / 2 + 8 + kBlock - 1) / kBlock) * kBlock);
        // (the MFMA backward pass addresses the expansion records with 32-bit byte offsets: only widen while that holds)
        const size_t bytes = ((size_t)kBwdFrontPad + (size_t)N_ + 2) * RR::EP * (size_t)(Bp_ + want) * sizeof(RS);
        if (bytes < (size_t)0xffffffffu) twin_cap_ = want;
      }
    }
    // Segments of rejection streaks in the batched sweeps (DevArrays::seg_*): shadow columns [Bpad, Bpad + seg_total_),
    // one slice per chain of sweeps, then the twin columns.  Only for batches that run batched sweeps at all; ALTRO_HIP_SEGMENTS=0 switches it off.
    seg_total_ = 0;
    seg_col0_ = Bp_;
    if constexpr (kMfmaBackward) {
      const char* e = std::getenv("ALTRO_HIP_SEGMENTS");
      if (!(e && atoi(e) == 0) && !fast_forward_ && B_ > persist_at_) {
        // (5/8 of the batch: most of the plateau of config 3 -- a quarter of its instances, three clones each -- fits, and
        //  what the wider arrays cost a batch that never splits stays around 0.5 %: config 2 runs 0.3 % / 1 % / 2 % slower
        //  with 2048 / 3520 / 4096 unused columns behind its 4672; config 3 is as fast with 2560 as with 4096;
        //  profiles/r05_experiments.txt #3)
        int want = (Bp_ * 5 / 8) / kBlock * kBlock;
        const size_t bytes = ((size_t)kBwdFrontPad + (size_t)N_ + 2) * RR::EP * (size_t)(Bp_ + want + twin_cap_) * sizeof(RS);
        if (bytes < (size_t)0xffffffffu) seg_total_ = want;
      }
    }
    seg_parts_ = 4;
    Bp_ += seg_total_ + twin_cap_;
    std::memset(&pd_, 0, sizeof(pd_));
    pd_.n = n;
    pd_.m = m;
    pd_.N = N_;
    pd_.B = B_;
    pd_.Bp = Bp_;

    std::vector<T> pool;             // shared parameters
    std::vector<std::vector<T>> ip;  // per-instance slots, each [B]
    auto new_slot = [&]() {
      ip.emplace_back(B_, T(0));
      return (int)ip.size() - 1;
    };

    // --- costs: the last SetCostFunction on a knot wins (problem.hpp:113-127) ---------------------
    std::vector<int> knot_cost(N_ + 1, -1);
    for (size_t ci = 0; ci < s.costs.size(); ++ci)
      for (int k = s.costs[ci].k_begin; k < s.costs[ci].k_end; ++k) knot_cost[k] = (int)ci;
    for (int k = 0; k <= N_; ++k)
      if (knot_cost[k] < 0) {
        err_ = "cost function missing at knot " + std::to_string(k) + " (Problem::IsFullyDefined)";
        return ALTRO_NOT_READY;
      }
    std::map<int, int> group_of_cost;
    for (int k = 0; k <= N_; ++k) {
      const int ci = knot_cost[k];
      if (group_of_cost.count(ci)) continue;
      if (pd_.ngroups >= kMaxCostGroups) {
        err_ = "too many distinct cost functions";
        return ALTRO_UNSUPPORTED;
      }
      const CostSpec& c = s.costs[ci];
      CostGroupDesc g{};
      if (c.user) {
        // the user model's UserCost (altro_set_user_cost): only its parameters travel
        if constexpr (!kHasUserCost) {
          err_ = "this model defines no UserCost (altro_set_user_cost needs a user model whose source defines ALTRO_USER_COST)";
          return ALTRO_INVALID_ARG;
        } else {
          const int NP = UserCostParams(c.user - 1);  // (-1: no such type in the source)
          if (NP < 0) {
            err_ = "user cost type " + std::to_string(c.user - 1) + ": the model's source defines " +
                   std::to_string(UserCostList::size) + " cost type(s) (ALTRO_USER_COSTS)";
            return ALTRO_INVALID_ARG;
          }
          if ((int)c.params.size() != NP * (c.per_instance ? B_ : 1)) {
            err_ = "user cost type " + std::to_string(c.user - 1) + ": expected " + std::to_string(NP) + " parameters" +
                   (c.per_instance ? " per instance" : "");
            return ALTRO_INVALID_ARG;
          }
          g.user = c.user;
          g.u_pi = c.per_instance ? 1 : 0;
          // (the quadratic fields stay valid, all-zero pool entries: every generic read is in bounds)
          g.Q_off = g.R_off = g.q_off = g.r_off = g.c_off = (int)pool.size();
          for (int e = 0; e < n * n + m * m; ++e) pool.push_back(T(0));
          if (!g.u_pi) {
            g.u_off = (int)pool.size();
            for (int e = 0; e < NP; ++e) pool.push_back(T(c.params[e]));
          } else {
            g.u_off = (int)ip.size();
            for (int e = 0; e < NP; ++e) {
              const int sl = new_slot();
              for (int b = 0; b < B_; ++b) ip[sl][b] = T(c.params[(size_t)b * NP + e]);
            }
          }
          group_of_cost[ci] = pd_.ngroups;
          pd_.grp[pd_.ngroups++] = g;
          continue;
        }
      }
      // QuadraticCost::LQRCost (examples/quadratic_cost.hpp:29-39), evaluated in T like the oracle
      g.Q_off = (int)pool.size();
      for (int e = 0; e < n * n; ++e) pool.push_back(T(c.Q[e]));
      g.R_off = (int)pool.size();
      for (int e = 0; e < m * m; ++e) pool.push_back(T(c.R[e]));
      const T* Q = &pool[g.Q_off];
      const T* R = &pool[g.R_off];
      g.q_diag = g.r_diag = 1;
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i)
          if (i != j && Q[i + j * n] != T(0)) g.q_diag = 0;
      for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i)
          if (i != j && R[i + j * m] != T(0)) g.r_diag = 0;
      const bool xpi = (c.per_instance & 1) != 0, upi = (c.per_instance & 2) != 0;
      g.q_pi = xpi;
      g.r_pi = upi;
      g.c_pi = xpi || upi;
      const int ninst_q = xpi ? B_ : 1, ninst_r = upi ? B_ : 1, ninst_c = g.c_pi ? B_ : 1;
      std::vector<T> q((size_t)ninst_q * n), r((size_t)ninst_r * m), cc(ninst_c), xa((size_t)ninst_q), ub_((size_t)ninst_r);
      std::vector<T> xQx(ninst_q), uRu(ninst_r);
      for (int b = 0; b < ninst_q; ++b) {
        T xr[n], Qx[n];
        for (int i = 0; i < n; ++i) xr[i] = T(c.xref[(size_t)b * n + i]);
        T acc = T(0);
        for (int i = 0; i < n; ++i) {
          T sacc = T(0);
          for (int j = 0; j < n; ++j) sacc += Q[i + j * n] * xr[j];
          Qx[i] = sacc;
          q[(size_t)b * n + i] = -sacc;
        }
        for (int i = 0; i < n; ++i) acc += xr[i] * Qx[i];
        xQx[b] = acc;
      }
      for (int b = 0; b < ninst_r; ++b) {
        T ur[m], Ru[m];
        for (int i = 0; i < m; ++i) ur[i] = T(c.uref[(size_t)b * m + i]);
        T acc = T(0);
        for (int i = 0; i < m; ++i) {
          T sacc = T(0);
          for (int j = 0; j < m; ++j) sacc += R[i + j * m] * ur[j];
          Ru[i] = sacc;
          r[(size_t)b * m + i] = -sacc;
        }
        for (int i = 0; i < m; ++i) acc += ur[i] * Ru[i];
        uRu[b] = acc;
      }
      for (int b = 0; b < ninst_c; ++b) cc[b] = T(0.5) * xQx[xpi ? b : 0] + T(0.5) * uRu[upi ? b : 0];
      auto put = [&](bool pi, const std::vector<T>& v, int E) {
        if (!pi) {
          int off = (int)pool.size();
          for (int e = 0; e < E; ++e) pool.push_back(v[e]);
          return off;
        }
        int first = -1;
        for (int e = 0; e < E; ++e) {
          int sl = new_slot();
          if (e == 0) first = sl;
          for (int b = 0; b < B_; ++b) ip[sl][b] = v[(size_t)b * E + e];
        }
        return first;
      };
      g.q_off = put(xpi, q, n);
      g.r_off = put(upi, r, m);
      g.c_off = put(g.c_pi, cc, 1);
      group_of_cost[ci] = pd_.ngroups;
      pd_.grp[pd_.ngroups++] = g;
    }

    // --- constraints: per knot, equalities first then inequalities, insertion order kept ----------
    struct ConBuilt {
      ConDesc d;
    };
    std::vector<ConBuilt> built(s.cons.size());
    for (size_t i = 0; i < s.cons.size(); ++i) {
      const ConSpec& c = s.cons[i];
      ConDesc d{};
      d.kind = c.kind;
      d.per_instance = c.per_instance ? 1 : 0;
      if (c.kind == ALTRO_CON_GOAL) {
        if (c.nparams != n) {
          err_ = "goal constraint needs n parameters";
          return ALTRO_INVALID_ARG;
        }
        d.type = 0;
        d.p = n;
      } else if (c.kind == ALTRO_CON_CONTROL_BOUND) {
        if (c.nparams != 2 * m || c.per_instance) {
          err_ = "control bound needs 2m shared parameters";
          return ALTRO_INVALID_ARG;
        }
        d.type = 1;
        for (int j = 0; j < m; ++j) {  // GetFiniteIndices, basic_constraints.hpp:138-145
          if (std::abs(c.params[j]) < std::numeric_limits<double>::max()) d.lo_mask |= 1u << j;
          if (std::abs(c.params[m + j]) < std::numeric_limits<double>::max()) d.hi_mask |= 1u << j;
        }
        d.p = __builtin_popcount(d.lo_mask) + __builtin_popcount(d.hi_mask);
      } else if (c.kind == ALTRO_CON_CIRCLE) {
        if (c.nparams % 3 != 0 || c.nparams == 0 || n < 2) {
          err_ = "circle constraint needs (cx, cy, r) triples";
          return ALTRO_INVALID_ARG;
        }
        d.type = 1;
        d.p = c.nparams / 3;
      } else if (c.kind == ALTRO_CON_USER) {
        // the user model's UserConstraint: OutputDimension and cone come from its source
        if constexpr (!kHasUserCon) {
          err_ = "this model defines no UserConstraint (ALTRO_CON_USER needs a user model whose source defines ALTRO_USER_CONSTRAINT)";
          return ALTRO_INVALID_ARG;
        } else {
          int unp = -1, up = -1, ueq = -1;
          UserConInfo(c.user_type, &unp, &up, &ueq);
          if (unp < 0) {
            err_ = "user constraint type " + std::to_string(c.user_type) + ": the model's source defines " +
                   std::to_string(UserConList::size) + " constraint type(s) (ALTRO_USER_CONSTRAINTS)";
            return ALTRO_INVALID_ARG;
          }
          if (c.nparams != unp) {
            err_ = "user constraint type " + std::to_string(c.user_type) + ": expected " + std::to_string(unp) + " parameters";
            return ALTRO_INVALID_ARG;
          }
          d.type = ueq ? 0 : 1;
          d.p = up;
          d.lo_mask = (unsigned)c.user_type;  // (the device dispatches on it: user_con_auglag)
        }
      } else {
        err_ = "unknown constraint kind";
        return ALTRO_INVALID_ARG;
      }
      if (d.kind == ALTRO_CON_CONTROL_BOUND) {
        d.param_off = (int)pool.size();
        for (int j = 0; j < m; ++j)
          if ((d.lo_mask >> j) & 1u) pool.push_back(T(c.params[j]));
        for (int j = 0; j < m; ++j)
          if ((d.hi_mask >> j) & 1u) pool.push_back(T(c.params[m + j]));
      } else if (!d.per_instance) {
        d.param_off = (int)pool.size();
        for (int e = 0; e < c.nparams; ++e) pool.push_back(T(c.params[e]));
      } else {
        int first = -1;
        for (int e = 0; e < c.nparams; ++e) {
          int sl = new_slot();
          if (e == 0) first = sl;
          for (int b = 0; b < B_; ++b) ip[sl][b] = T(c.params[(size_t)b * c.nparams + e]);
        }
        d.param_off = first;
      }
      built[i].d = d;
    }
    std::vector<std::vector<int>> knot_cons(N_ + 1);
    for (size_t i = 0; i < s.cons.size(); ++i)
      for (int k = s.cons[i].k_begin; k < s.cons[i].k_end; ++k) knot_cons[k].push_back((int)i);
    for (auto& v : knot_cons)
      std::stable_partition(v.begin(), v.end(), [&](int i) { return built[i].d.type == 0; });

    // --- knot classes --------------------------------------------------------------------------------
    std::map<std::vector<int>, int> class_of;
    knot_class_.assign(N_ + 1, 0);
    knot_rowbase_.assign(N_ + 1, 0);
    int rows = 0;
    for (int k = 0; k <= N_; ++k) {
      std::vector<int> key;
      key.push_back(group_of_cost[knot_cost[k]]);
      for (int i : knot_cons[k]) key.push_back(i);
      auto it = class_of.find(key);
      int cls;
      if (it == class_of.end()) {
        if (pd_.nclass >= kMaxClasses) {
          err_ = "too many distinct knot-point classes";
          return ALTRO_UNSUPPORTED;
        }
        if ((int)knot_cons[k].size() > kMaxConPerKnot) {
          err_ = "too many constraints on one knot point";
          return ALTRO_UNSUPPORTED;
        }
        cls = pd_.nclass++;
        KnotClass& kc = pd_.cls[cls];
        kc.cost_group = key[0];
        kc.ncon = (int)knot_cons[k].size();
        int ro = 0;
        for (int c = 0; c < kc.ncon; ++c) {
          kc.con[c] = built[knot_cons[k][c]].d;
          kc.con[c].row_off = ro;
          ro += kc.con[c].p;
        }
        kc.nrows = ro;
        class_of[key] = cls;
      } else {
        cls = it->second;
      }
      knot_class_[k] = cls;
      knot_rowbase_[k] = rows;
      rows += pd_.cls[cls].nrows;
    }
    pd_.total_rows = rows;
    pd_.nslots = (int)ip.size();
    pd_.npool = (int)pool.size();
    pd_.hstep = s.hstep;
    // runs of consecutive knots sharing a class (scalar-register friendly serial loops)
    pd_.nruns = 0;
    for (int k = 0; k <= N_; ++k) {
      if (pd_.nruns > 0 && pd_.runs[pd_.nruns - 1].cls == knot_class_[k]) {
        pd_.runs[pd_.nruns - 1].k_end = k + 1;
        continue;
      }
      if (pd_.nruns >= kMaxRuns) {
        err_ = "too many runs of distinct knot-point classes";
        return ALTRO_UNSUPPORTED;
      }
      KnotRun run{};
      run.k_begin = k;
      run.k_end = k + 1;
      run.cls = knot_class_[k];
      run.rowbase = knot_rowbase_[k];
      {
        const KnotClass& kc = pd_.cls[run.cls];
        const unsigned full = (1u << m) - 1u;
        auto is_full_bound = [&](const ConDesc& c) {
          return c.kind == ALTRO_CON_CONTROL_BOUND && c.lo_mask == full && c.hi_mask == full;
        };
        auto is_circle = [&](const ConDesc& c) { return c.kind == ALTRO_CON_CIRCLE && c.p <= kMaxFastCircles; };  // (cost_consumer_run keeps them in registers)
        run.fast = kFastGeneric;
        if (pd_.grp[kc.cost_group].q_diag && pd_.grp[kc.cost_group].r_diag) {
          if (kc.ncon == 0) run.fast = kFastNone;
          else if (kc.ncon == 1 && is_full_bound(kc.con[0])) run.fast = kFastB;
          else if (kc.ncon == 1 && is_circle(kc.con[0])) run.fast = kFastC;
          else if (kc.ncon == 2 && is_circle(kc.con[0]) && is_full_bound(kc.con[1])) run.fast = kFastCB;
          else if (kc.ncon == 2 && is_full_bound(kc.con[0]) && is_circle(kc.con[1])) run.fast = kFastBC;
        }
      }
      pd_.runs[pd_.nruns++] = run;
    }
    {
      const double lim = 2147483647.0;
      auto biggest = [&]() {
        return std::max({(double)(N_ + 1) * RR::EP * Bp_, (double)(N_ + 1) * nm * kLineSearchLanes * Bp_, (double)std::max(rows, 1) * Bp_});
      };
      // (the shadow columns widen every array; a batch that fits 32-bit indexing without them must not be refused
      //  because of them -- drop the segments' columns first, then the twins')
      if (biggest() > lim && seg_total_ > 0) {
        Bp_ -= seg_total_;
        seg_total_ = 0;
      }
      if (biggest() > lim && twin_cap_ > 0) {
        Bp_ -= twin_cap_;
        twin_cap_ = 0;
      }
      pd_.Bp = Bp_;
      if (biggest() > lim) {
        err_ = "problem too large for 32-bit device indexing (split the batch over several handles)";
        return ALTRO_UNSUPPORTED;
      }
    }

    // --- device allocations --------------------------------------------------------------------------
    std::memset(&A_, 0, sizeof(A_));
    A_.B = B_;
    A_.Bp = Bp_;
    A_.N = N_;
    const size_t bp = Bp_;
#define ALTRO_ALLOC(ptr, count)                     \
  do {                                              \
    altro_status st_ = Alloc(&(ptr), (count));      \
    if (st_ != ALTRO_OK) return st_;                \
  } while (0)
    ALTRO_ALLOC(A_.x0, (size_t)R::nP * bp);
    ALTRO_ALLOC(A_.X, (size_t)(N_ + 1) * R::nP * bp);
    ALTRO_ALLOC(A_.U, (size_t)N_ * R::mP * bp);
    {
      // k_backward_mfma reads a zeroed pad record behind the last knot and lets its prefetch run up to
      // kBwdFrontPad records below knot 0
      const size_t front = (size_t)kBwdFrontPad * RR::EP * bp;
      RS* exp = nullptr;
      ALTRO_ALLOC(exp, front + (size_t)(N_ + 1) * RR::EP * bp + RR::EP);
      A_.EXP = exp + front;
      mfma_offsets_ok_ = (front + (size_t)(N_ + 2) * RR::EP * bp) * sizeof(RS) < (size_t)0xffffffffu;
    }
    ALTRO_ALLOC(A_.costs, (size_t)(N_ + 1) * bp);
    {
      RS* kd = nullptr;
      ALTRO_ALLOC(kd, (size_t)N_ * RR::KP * bp);
      A_.KD = kd;
    }
    ALTRO_ALLOC(A_.CTG, (size_t)(N_ + 1) * R::CP * bp + kBlock);  // (+ the junk sink of the recording backward pass)
    ALTRO_HIP_CHECK(hipMalloc((void**)&A_.trial, (size_t)(N_ + 1) * nm * kLineSearchLanes * bp * sizeof(T)));
    allocs_.push_back((void*)A_.trial);
    if (poison_on_) {
      std::vector<unsigned> junk((size_t)(N_ + 1) * nm * kLineSearchLanes * bp * sizeof(T) / 4);
      for (size_t i = 0; i < junk.size(); ++i) junk[i] = poison_mix_ ? (poison_pattern_ ^ ((unsigned)i * 2654435761u)) : poison_pattern_;
      ALTRO_HIP_CHECK(CopySync(A_.trial, junk.data(), junk.size() * 4, hipMemcpyHostToDevice));
    }
    ALTRO_ALLOC(A_.lam, (size_t)rows * bp);
    ALTRO_ALLOC(A_.pen, (size_t)rows * bp);
    ALTRO_ALLOC(A_.cval, (size_t)rows * bp);
    ALTRO_ALLOC(d_tmp_, bp);
    if (twin_cap_ > 0) ALTRO_ALLOC(d_twin_box_, (size_t)twin_cap_ * (kTwWords + 1));  // mailboxes, then the state words
    if (seg_total_ > 0) {
      // (kept out of A_: only the launches of Solve that take part in the scheme see them, SegArrays)
      ALTRO_ALLOC(seg_.end, bp);
      ALTRO_ALLOC(seg_.next, bp);
      ALTRO_ALLOC(seg_.flag, bp);
      ALTRO_ALLOC(seg_.streak, bp);
      ALTRO_ALLOC(seg_.tot0, bp);
      ALTRO_ALLOC(seg_.rho0, bp);
      ALTRO_ALLOC(seg_.drho0, bp);
      ALTRO_ALLOC(d_seg_cursor_, kMaxChains);
    }
    ALTRO_ALLOC(d_list_[0], bp);
    ALTRO_ALLOC(d_list_[1], bp);
    {
      // chains of batched sweeps: four for a batch that fills the GPU several times over (measured on 4096 instances:
      // config 2 -5 %, config 3 -14 %; a quarter of a 1024-instance batch no longer fills the CUs)
      chains_ = B_ >= 2048 ? kDefaultChains : 1;
      // (the streams of a process share four hardware queues: a second engine with chains of its own would queue up
      //  behind this one's persistent kernel -- only the first large engine of a process gets them)
      if (chains_ > 1 && ChainClaim(desc_.device_id, 0) > 0) chains_ = 1;
      if (const char* e = std::getenv("ALTRO_HIP_CHAINS")) chains_ = std::max(1, std::min(kMaxChains, atoi(e)));
      for (;;) {
        chain_size_ = ((B_ + chains_ - 1) / chains_ + kBlock - 1) / kBlock * kBlock;
        if (chains_ == 1 || chain_size_ * (chains_ - 1) < B_) break;
        chains_--;  // (the last chain would be empty)
      }
      if (chains_ > 1) {
        ChainClaim(desc_.device_id, +1);
        counted_chained_ = true;
        ALTRO_ALLOC(d_iota_, bp);
        ALTRO_ALLOC(d_merged_, bp);
        std::vector<int> iota(bp);
        for (int i = 0; i < bp; ++i) iota[i] = i;
        ALTRO_HIP_CHECK(CopySync(d_iota_, iota.data(), (size_t)bp * sizeof(int), hipMemcpyHostToDevice));
        for (int c = 1; c < chains_; ++c) {
          ALTRO_HIP_CHECK(hipStreamCreateWithFlags(&chain_stream_[c], hipStreamNonBlocking));
          ALTRO_HIP_CHECK(hipEventCreateWithFlags(&chain_ev_[c], hipEventDisableTiming));
        }
      }
      if (chains_ > 1 && !std::getenv("ALTRO_HIP_CHAINS")) {
        // Do the chains' streams run side by side?  The streams of a process share a few hardware queues (four unless
        // GPU_MAX_HW_QUEUES says otherwise), handed out in order of creation: with other streams around (another
        // handle, a framework's stream pool, RCCL) two chains can land on one queue and take turns -- slower than one
        // chain.  One wavefront spinning for 200 us on every chain stream: side by side they take 200 us together.
        // Timed ON THE DEVICE: every spin kernel stamps its start and its end with the constant 100 MHz clock, and the
        // chains run side by side iff the intervals overlap (the latest start lies well before the earliest end).  Host
        // wall time around launches and synchronisations would also count launch latency and host jitter -- eight ranks
        // starting at once could silently lose their chains that way.
        const long long ticks = 20000;  // of the 100 MHz constant clock: 200 us
        ALTRO_HIP_CHECK(hipStreamSynchronize(stream_));
        long long* d_stamps = nullptr;
        ALTRO_HIP_CHECK(hipMalloc((void**)&d_stamps, 2 * kMaxChains * sizeof(long long)));
        long long h_stamps[2 * kMaxChains] = {0};
        long long best_overlap = -(1ll << 60);
        for (int rep = 0; rep < 2; ++rep) {  // (the first round also loads the kernel)
          for (int c = 0; c < chains_; ++c)
            hipLaunchKernelGGL((k_spin<0>), dim3(1), dim3(kBlock), 0, c == 0 ? stream_ : chain_stream_[c], ticks, d_stamps + 2 * c);
          for (int c = 0; c < chains_; ++c) ALTRO_HIP_CHECK(hipStreamSynchronize(c == 0 ? stream_ : chain_stream_[c]));
          ALTRO_HIP_CHECK(CopySync(h_stamps, d_stamps, sizeof(h_stamps), hipMemcpyDeviceToHost));
          long long last_start = h_stamps[0], first_end = h_stamps[1];
          for (int c = 1; c < chains_; ++c) {
            last_start = std::max(last_start, h_stamps[2 * c]);
            first_end = std::min(first_end, h_stamps[2 * c + 1]);
          }
          best_overlap = std::max(best_overlap, first_end - last_start);
        }
        hipFree(d_stamps);
        chain_overlap_ticks_ = best_overlap;
        if (best_overlap < ticks / 2) {  // two of them ran (mostly) one after the other
          for (int c = 1; c < chains_; ++c) {
            hipStreamDestroy(chain_stream_[c]);
            hipEventDestroy(chain_ev_[c]);
            chain_stream_[c] = nullptr;
            chain_ev_[c] = nullptr;
          }
          chains_ = 1;
          chain_size_ = Bp_;
          ChainClaim(desc_.device_id, -1);
          counted_chained_ = false;
        }
      }
    }
    ALTRO_ALLOC(X_init_, (size_t)(N_ + 1) * R::nP * bp);
    ALTRO_ALLOC(U_init_, (size_t)N_ * R::mP * bp);
    T* dpool = nullptr;
    T* dipool = nullptr;
    ALTRO_ALLOC(dpool, pool.size());
    ALTRO_ALLOC(dipool, ip.size() * bp);
    if (!pool.empty())
      ALTRO_HIP_CHECK(CopySync(dpool, pool.data(), pool.size() * sizeof(T), hipMemcpyHostToDevice));
    if (!ip.empty()) {
      std::vector<T> flat(ip.size() * bp, T(0));
      for (size_t sl = 0; sl < ip.size(); ++sl)
        for (int b = 0; b < B_; ++b) flat[sl * bp + b] = ip[sl][b];
      ALTRO_HIP_CHECK(CopySync(dipool, flat.data(), flat.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    A_.pool = dpool;
    A_.ipool = dipool;
    int *dkc = nullptr, *dkr = nullptr;
    ALTRO_ALLOC(dkc, N_ + 1);
    ALTRO_ALLOC(dkr, N_ + 1);
    ALTRO_HIP_CHECK(CopySync(dkc, knot_class_.data(), (N_ + 1) * sizeof(int), hipMemcpyHostToDevice));
    ALTRO_HIP_CHECK(CopySync(dkr, knot_rowbase_.data(), (N_ + 1) * sizeof(int), hipMemcpyHostToDevice));
    A_.knot_class = dkc;
    A_.knot_rowbase = dkr;
    ALTRO_ALLOC(d_phi_, kMaxClasses * kMaxConPerKnot);
    {
      std::vector<double> v(kMaxClasses * kMaxConPerKnot, s.phi >= 1.0 ? s.phi : 10.0);  // constraint_values.hpp:30
      ALTRO_HIP_CHECK(CopySync(d_phi_, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    A_.phi = d_phi_;
    ALTRO_ALLOC(d_pd_, 1);
    ALTRO_HIP_CHECK(CopySync(d_pd_, &pd_, sizeof(pd_), hipMemcpyHostToDevice));
    // per-instance scalar state, one slab each so GetStats is two copies
    ALTRO_ALLOC(d_scalarT_, (size_t)kNumScalarT * bp);
    ALTRO_ALLOC(d_scalarI_, (size_t)kNumScalarI * bp);
    double** tp[kNumScalarT] = {&A_.rho_reg, &A_.drho,     &A_.dV0,  &A_.dV1,  &A_.J0,
                           &A_.initial_cost, &A_.cost_cur, &A_.cost_prev, &A_.dJ, &A_.grad,
                           &A_.viol,    &A_.penmax,   &A_.alpha, &A_.z,   &A_.reg_log};
    for (int i = 0; i < kNumScalarT; ++i) *tp[i] = d_scalarT_ + (size_t)i * bp;
    int** ipn[kNumScalarI] = {&A_.status, &A_.status_al, &A_.it_inner, &A_.it_outer,
                              &A_.it_total, &A_.phase,   &A_.need_init_cost};
    for (int i = 0; i < kNumScalarI; ++i) *ipn[i] = d_scalarI_ + (size_t)i * bp;
    {
      std::vector<int> ones(bp, ALTRO_UNSOLVED);
      ALTRO_HIP_CHECK(CopySync(A_.status, ones.data(), bp * sizeof(int), hipMemcpyHostToDevice));
      ALTRO_HIP_CHECK(CopySync(A_.status_al, ones.data(), bp * sizeof(int), hipMemcpyHostToDevice));
    }
    {
      // LDS plan of the forward pass: up to 3 instances per wavefront, at most 80 KiB per workgroup
      // (two workgroups per CU); if even one instance does not fit in 160 KiB, read from HBM instead.
      auto padv = [](size_t e) { return (e + R::V - 1) / R::V * R::V; };
      // (+ the optional padding between the instances' blocks of a workgroup: FwdLds::total)
      auto padded = [](size_t bytes) { return bytes + (size_t)fwd_block_pad_bytes((long long)bytes); };
      const size_t per_inst = padded(((size_t)(N_ + 1) * R::nP + (size_t)N_ * R::mP + (size_t)N_ * R::KP +
                                      2 * padv((size_t)rows) + padv(ip.size())) * sizeof(T));
      const int lanes_max = kBlock / kLineSearchLanes;
      fwd_per_wave_ = lanes_max;
      while (fwd_per_wave_ > 1 && fwd_per_wave_ * per_inst > 80 * 1024) fwd_per_wave_--;
      const size_t shared_bytes = (padv(pool.size()) + kFwdSlots * (size_t)nm * kBlock) * sizeof(T) + 2 * kBlock * sizeof(int) +
                                  kBlock * sizeof(double);
      while (fwd_per_wave_ > 1 && shared_bytes + fwd_per_wave_ * per_inst > 80 * 1024) fwd_per_wave_--;
      if (const char* e = std::getenv("ALTRO_HIP_FWD_PER_WAVE")) fwd_per_wave_ = std::max(1, std::min(fwd_per_wave_, atoi(e)));
      fwd_lds_bytes_ = shared_bytes + fwd_per_wave_ * per_inst;
      fwd_shared_bytes_ = shared_bytes;
      fwd_per_inst_bytes_ = per_inst;
      fused_lds_bytes_ = (shared_bytes + (2 * kSyncFused - kFwdSlots) * (size_t)nm * kBlock * sizeof(T) + per_inst + 15) / 16 * 16 +
                         (4 + 2 + kBlock + 2 + 16) * sizeof(double) +
                         (size_t)(N_ + 1) * kLineSearchLanes * nm * sizeof(T) +  // + the candidates of one instance
                         ((size_t)N_ * R::KP + kBlock) * sizeof(T) + 48 * sizeof(double) +  // + the speculative pass (gains, hand-over), step-length table, sequence words
                         ((size_t)N_ + 4) * sizeof(T) +                                    // + the knot costs of the expansion step
                         padv((size_t)rows) * sizeof(T);                                   // + the constraint values of the expansions computed ahead
      kdg_ = false;
      rg_ = false;
      if constexpr (kRgEligible && !kKdgEligible) {
        // Small models: the rollout wave reads (xbar, ubar, K, d) from global memory two knots ahead; LDS keeps only the
        // multipliers and the parameters, so that four workgroups (the register limit) instead of two share a CU.
        // Needs the winner-only gradient measure of phase 2, whose terms live in the hand-off slots.
        const size_t per_inst_g = padded((2 * padv((size_t)rows) + padv(ip.size())) * sizeof(T));
        // Taken when it lets more instances be resident on a CU than the fully staged variant: workgroups per CU =
        // min(LDS limit, register limit: waves per SIMD of the kernel * 4 SIMDs / 3 waves).  Config 3 (fp32 records: 152
        // VGPRs, three waves per SIMD): 6 -> 9 instances, -7 % per forward sweep; config 2 (fp64 records in flight: 171
        // VGPRs, two waves per SIMD): 6 -> 6, and the staged variant is 2 % faster.  ALTRO_HIP_FWD_SRC=lds / global
        // override the choice.
        auto resident = [&](const void* fn, size_t lds_wg) {
          hipFuncAttributes at{};
          if (hipFuncGetAttributes(&at, fn) != hipSuccess || at.numRegs <= 0) return 0;
          const int waves = std::min(8, 512 / ((at.numRegs + 7) / 8 * 8));  // per SIMD
          const int by_regs = waves * 4 / kFwdWaves, by_lds = (int)(160 * 1024 / lds_wg);
          return lanes_max * std::min(by_regs, by_lds);
        };
        const char* src = std::getenv("ALTRO_HIP_FWD_SRC");
        bool want = false;
        if (src) {
          want = std::string(src) == "global";
        } else if (fwd_per_wave_ == lanes_max) {
          want = resident(reinterpret_cast<const void*>(&k_forward2<T, M, kSrcGlb>), shared_bytes + lanes_max * per_inst_g) >
                 resident(reinterpret_cast<const void*>(&k_forward2<T, M, kSrcLds>), shared_bytes + lanes_max * per_inst);
        } else {
          want = true;  // the staged block does not even allow three instances per workgroup
        }
        if (want && 16 + lanes_max * N_ <= kFwdSlots * nm * kBlock && shared_bytes + lanes_max * per_inst_g <= 80 * 1024) {
          rg_ = true;
          fwd_per_wave_ = lanes_max;
          if (const char* e = std::getenv("ALTRO_HIP_FWD_PER_WAVE")) fwd_per_wave_ = std::max(1, std::min(fwd_per_wave_, atoi(e)));
          fwd_per_inst_bytes_ = per_inst_g;
          fwd_lds_bytes_ = shared_bytes + fwd_per_wave_ * per_inst_g;
        }
      }
      if constexpr (kKdgEligible) {
        // Models whose gain records fill the LDS (12-state model: 83 of 134 KB per instance -> one instance per CU):
        // keep only d in LDS, let the rollout wave read K from global memory one knot ahead, and put up to three
        // instances into one workgroup (one workgroup per CU, 160 KiB)
        const size_t per_inst_b = padded(((size_t)(N_ + 1) * R::nP + (size_t)N_ * R::mP + (size_t)N_ * R::mP +
                                          2 * padv((size_t)rows) + padv(ip.size())) * sizeof(T));
        int pw_b = lanes_max;
        while (pw_b > 1 && shared_bytes + pw_b * per_inst_b > 160 * 1024) pw_b--;
        if (shared_bytes + per_inst > 80 * 1024 && pw_b > 1) {
          kdg_ = true;
          fwd_per_wave_ = pw_b;
          fwd_per_inst_bytes_ = per_inst_b;
          fwd_lds_bytes_ = shared_bytes + pw_b * per_inst_b;
        }
        // Round 6: ... or NOTHING of the trajectory and the gains in LDS (k_forward2<.., kSrcGlb> for the large
        // models: the rollout wave reads xbar, ubar, K, d from global memory one knot ahead, the gain record in its storage
        // type; one barrier per knot, so two hand-off slots instead of four).  What is left -- the multipliers and the slots --
        // lets TWO workgroups share a CU's LDS, and the variant is compiled for two waves per SIMD (256 registers): config 4
        // keeps 4 instances on a CU instead of 2, its 512 forward workgroups run in ONE round instead of two.  Taken when it
        // raises the instances resident on a CU (ALTRO_HIP_FWD_SRC=kdg | global overrides).
        {
          const size_t shared_g = (padv(pool.size()) + 2 * (size_t)fwd_sync_batched<M, kSrcGlb>() * nm * kBlock) * sizeof(T) +
                                  2 * kBlock * sizeof(int) + kBlock * sizeof(double);
          const size_t per_inst_g = padded((2 * padv((size_t)rows) + padv(ip.size())) * sizeof(T));
          int pw_g = lanes_max;
          while (pw_g > 1 && shared_g + pw_g * per_inst_g > 80 * 1024) pw_g--;
          hipFuncAttributes at{};
          const bool have = hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_forward2<T, M, kSrcGlb>)) == hipSuccess && at.numRegs > 0;
          const int waves = have ? std::min(8, 512 / ((at.numRegs + 7) / 8 * 8)) : 0;  // per SIMD
          const int wgs_g = std::min(waves * 4 / kFwdWaves, (int)(160 * 1024 / (shared_g + pw_g * per_inst_g)));
          const int resident_g = wgs_g * pw_g, resident_now = kdg_ ? fwd_per_wave_ : (int)std::max<size_t>(1, 160 * 1024 / std::max<size_t>(1, fwd_lds_bytes_)) * fwd_per_wave_;
          const char* src = std::getenv("ALTRO_HIP_FWD_SRC");
          bool want = resident_g > resident_now;
          if (src) want = std::string(src) == "global";
          if (want && have && shared_g + pw_g * per_inst_g <= 160 * 1024 &&
              16 + pw_g * N_ <= 2 * fwd_sync_batched<M, kSrcGlb>() * nm * kBlock) {
            kdg_ = false;
            rg_ = true;
            fwd_per_wave_ = pw_g;
            fwd_shared_bytes_ = shared_g;
            fwd_per_inst_bytes_ = per_inst_g;
            fwd_lds_bytes_ = shared_g + pw_g * per_inst_g;
          }
        }
      }
      if (fwd_lds_bytes_ > 160 * 1024) {
        fwd_lds_bytes_ = 0;
        fwd_per_wave_ = lanes_max;
      } else if (fwd_lds_bytes_ > 64 * 1024) {
        if (kdg_) {
          if constexpr (kKdgEligible)
            ALTRO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_forward2<T, M, kSrcKdg>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds_bytes_));
        } else if (rg_) {
          if constexpr (kRgEligible)
            ALTRO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_forward2<T, M, kSrcGlb>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds_bytes_));
        } else {
          ALTRO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_forward2<T, M, kSrcLds>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds_bytes_));
        }
      }
      if constexpr (kMfmaBackward) {
        if (fused_lds_bytes_ > 64 * 1024 && fused_lds_bytes_ <= 160 * 1024)
        {
          // (+ the two variants that know the segments of rejection streaks: default speculation modes only, see Solve)
          const void* variants[] = {reinterpret_cast<const void*>(&k_sweep_fused<T, M, false, kSpecFree, true>),
                                     reinterpret_cast<const void*>(&k_sweep_fused<T, M, true, kSpecWave, true>),
                                     reinterpret_cast<const void*>(&k_sweep_fused<T, M, false, kSpecOff>),
                                     reinterpret_cast<const void*>(&k_sweep_fused<T, M, true, kSpecOff>),
                                     reinterpret_cast<const void*>(&k_sweep_fused<T, M, false, kSpecWave>),
                                     reinterpret_cast<const void*>(&k_sweep_fused<T, M, true, kSpecWave>),
                                     reinterpret_cast<const void*>(&k_sweep_fused<T, M, false, kSpecFree>),
                                     reinterpret_cast<const void*>(&k_sweep_fused<T, M, true, kSpecFree>)};
          for (const void* fn : variants)
            ALTRO_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds_bytes_));
        }
      }
    }
    // The device-side sweep loop (k_sweep_loop, round 6): persistent workgroups of the forward pass's shape that run
    // E -> B -> F for the instances in their slots; as many workgroups as the GPU holds at once (registers: two waves per
    // SIMD; LDS: the forward block, which the gain buffer of the backward pass shares), never more than the batch fills.
    loop_groups_ = 0;
    if constexpr (kMfmaBackward) {
      const char* e = std::getenv("ALTRO_HIP_SWEEP_LOOP");
      if (!(e && atoi(e) == 0) && fwd_lds_bytes_ > 0 && !kdg_ && fwd_per_wave_ == lanes_per_wave() && B_ > persist_at_) {
        const size_t bwd_bytes = ((size_t)kBwdChunk * 4 * R::KP + kBlock) * sizeof(double);
        loop_lds_bytes_ = std::max(fwd_lds_bytes_, bwd_bytes);
        const void* fn = rg_ ? LoopKernel<kSrcGlb>() : LoopKernel<kSrcLds>();
        if (fn && loop_lds_bytes_ <= 160 * 1024) {
          if (loop_lds_bytes_ > 64 * 1024)
            ALTRO_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)loop_lds_bytes_));
          int per_cu = 0;
          ALTRO_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kFwdWaves * kBlock, loop_lds_bytes_));
          if (const char* e2 = std::getenv("ALTRO_HIP_LOOP_PER_CU")) per_cu = std::max(1, std::min(per_cu, atoi(e2)));
          const int slots = lanes_per_wave();
          loop_groups_ = std::max(0, std::min(per_cu * num_cus_, (B_ + slots - 1) / slots));
          loop_per_cu_ = per_cu;
          // WHEN it takes the bulk phase (measured, profiles/r06_experiments.txt #1): a batch that fits the slots of the
          // resident workgroups (two per CU, three slots each: 1536 instances on an MI355X) runs as fast (kTurn90) or up to
          // 23 % faster (obstacles, 1024 instances) than the host-paced sweeps, with no host in the loop; a larger batch is
          // worked off in generations -- slots refilled as instances finish -- at 1536 instances in flight against the
          // sweeps' whole batch over four chains, and is 15 - 25 % slower (4096 kTurn90: 7.5 against 6.3 ms).  So: the
          // loop by default iff the batch fits the slots; ALTRO_HIP_SWEEP_LOOP=1 forces it for any batch, =0 never.
          if (!(e && atoi(e) != 0) && B_ > per_cu * num_cus_ * slots) loop_groups_ = 0;
        }
        if (loop_groups_ > 0) {
          ALTRO_ALLOC(d_loop_win_, (size_t)loop_groups_ * 4);
          ALTRO_ALLOC(d_loop_ctl_, (size_t)kLwWords + 8);  // (+ the words the persistent tail kernel reports)
          ALTRO_ALLOC(d_loop_tail_, (size_t)B_ + 16);
        }
      }
    }
    // candidates of the batched line search: the first 6 trials and the last live one own a slot, deeper winners are
    // replayed (CandLayout); ALTRO_HIP_CAND_FRONT=19 stores every trial (round 3), 0 replays every accepted trial (tests).
    // Measured (ms per solve, configs 2 / 3 / 4): front 19: 8.31 / 39.9 / 5.78; 12: 8.15 / 39.5 / 5.75; 8: 8.04 / 38.8 / 5.70;
    // 6: 7.91 / 38.7 / 5.65; 4: 7.96 / 38.8 / 5.62 (profiles/r04_experiments.txt)
    A_.cand_front = 6;
    if (const char* e = std::getenv("ALTRO_HIP_CAND_FRONT")) A_.cand_front = std::max(0, std::min(kLineSearchLanes - 1, atoi(e)));
    A_.xcd_remap = 1;  // (0: workgroup i takes slot block i, round 3)
    if (!s.knot_model.empty()) {
      bool any = false;
      for (int k = 0; k < N_; ++k) any = any || s.knot_model[k] != 0;
      if (any) {
        int* dkm = nullptr;
        ALTRO_ALLOC(dkm, (size_t)N_ + 1);
        std::vector<int> km(s.knot_model.begin(), s.knot_model.begin() + N_);
        km.push_back(0);
        ALTRO_HIP_CHECK(CopySync(dkm, km.data(), km.size() * sizeof(int), hipMemcpyHostToDevice));
        A_.knot_model = dkm;
      }
    }
    uploaded_ = true;
    // penalties start at one (constraint_values.hpp:44); an earlier SetPenalty overrides
    hipLaunchKernelGGL(k_set_rows<T>, GridB(), dim3(kBlock), 0, stream_, A_, d_pd_, 0, 1,
                       T(s.penalty >= 0 ? s.penalty : 1.0));
    altro_status st = Sync();
    if (st == ALTRO_OK) st = SetInitialStateImpl(s);
    if (st == ALTRO_OK) st = SetKnotTimesImpl(s);
    if (st == ALTRO_OK) st = SetTrajectoryImpl(s);
    // (a failed upload leaves the engine as it was before it -- Upload() drops whatever a failed UploadImpl built: the
    //  C-ABI keeps the handle "not uploaded" and the next call uploads again instead of meeting "problem definition changed")
    return st;
#undef ALTRO_ALLOC
  }

  // ALTRO_HIP_DEBUG_POISON: shadow columns (segments of the batched sweeps, twins of the persistent kernel) start every
  // solve full of NaN words (k_poison_columns)
  void PoisonShadowColumns() {
    if (!poison_on_ || seg_total_ + twin_cap_ <= 0) return;
    const unsigned col0 = (unsigned)(Bp_ - seg_total_ - twin_cap_), ncols = (unsigned)(seg_total_ + twin_cap_);
    auto fill = [&](void* arr, size_t rows, size_t bytes_per_column_row) {
      if (!arr || rows == 0) return;
      hipLaunchKernelGGL((k_poison_columns<0>), dim3(1024), dim3(256), 0, stream_, (unsigned*)arr, (unsigned)rows, (unsigned)Bp_, col0, ncols,
                         (unsigned)(bytes_per_column_row / 4), poison_pattern_, poison_mix_);
    };
    fill(A_.X, N_ + 1, R::nP * sizeof(T));
    fill(A_.U, N_, R::mP * sizeof(T));
    fill(A_.x0, 1, R::nP * sizeof(T));
    fill(A_.costs, N_ + 1, sizeof(T));
    fill(A_.lam, pd_.total_rows, sizeof(T));
    fill(A_.pen, pd_.total_rows, sizeof(T));
    fill(A_.cval, pd_.total_rows, sizeof(T));
    fill((void*)A_.EXP, N_ + 1, RR::EP * sizeof(RS));
    fill((void*)A_.KD, N_, RR::KP * sizeof(RS));
  }

  // ---- the sweep loop -----------------------------------------------------------------------------
  hipEvent_t ProfEvent(size_t i) {
    while (prof_ev_.size() <= i) {
      hipEvent_t e;
      hipEventCreate(&e);
      prof_ev_.push_back(e);
    }
    return prof_ev_[i];
  }

  // One chain of batched sweeps: the instances [lo, hi) of the batch, swept on a stream of their own.  A large batch runs
  // as up to four chains (chains_): the three kernels of a sweep follow each other on one stream -- a latency-bound
  // backward launch, as long as its unluckiest instance, between two throughput-bound ones, and a forward launch whose
  // last round of workgroups leaves most CUs idle -- and chains that drift out of phase fill each other's gaps.  The
  // persistent tail kernel takes the lists all chains leave behind in ONE launch.
  struct Chain {
    int lo = 0, hi = 0;
    hipStream_t st = nullptr;
    int* d_cnt = nullptr;            // this chain's block of device counters (one per sweep; cursors of the list rebuilds)
    volatile int* h_cnt = nullptr;   // ... of host-mapped words (the counts the sweeps publish)
    int* h_cnt_dev = nullptr;
    int sweeps = 0;                  // sweeps enqueued
    int known = 0;                   // newest count the host knows (an upper bound of what the next sweep works on)
    bool waiting = false, done = false, tail = false;
    bool last_split = false;         // the sweep enqueued last was allowed to split rejection streaks (its list may outgrow `known`)
    bool split_seen = false;         // ... some sweep of this chain was: its slice of shadow columns may hold instances
    std::vector<size_t> ev;          // profiler events: chain start, then three per sweep
  };

  altro_status Solve(const altro_options& o, int mode) {
    if (!StepOk()) return ALTRO_NOT_READY;
    ALTRO_HIP_CHECK(hipSetDevice(desc_.device_id));
    const auto t0 = std::chrono::steady_clock::now();
    DevOpts d = ToDevOpts(o);
    d.fast_forward_stalls = fast_forward_ ? 1 : 0;
    const bool prof = o.profiler_enable != 0;
    last_mode_ilqr_ = (mode == kFwdILQR);
    last_opts_ = d;
    ctg_fresh_ = A_.record_ctg != 0;  // (recorded by the batched backward kernels of this solve, or readable through ReplayCtg)
    ctg_replayable_ = true;
    std::memset(&timing_, 0, sizeof(timing_));
    size_t nev = 0;
    cur_ = stream_;
    PoisonShadowColumns();
    // upper bound on sweeps: every sweep advances every active instance by one inner iteration
    // (64-bit product, clamped: the counters of the sweeps live in a pinned array of this length)
    const long long inner_outer = (long long)std::max(1, o.max_iterations_inner) *
                                  (long long)std::max(1, (mode == kFwdAL) ? o.max_iterations_outer : 1);
    const int max_sweeps =
        (int)std::max<long long>(1, std::min<long long>({(long long)o.max_iterations_total, inner_outer, 1LL << 20})) + 2;
    // Sweep i of a chain works on the instances that its sweep i-1 left active: a dense list built by the forward
    // kernel (two list buffers, one counter per sweep).  Sweep i+1's first kernel publishes the
    // length of its list (= what sweep i left) to pinned host memory; the host polls that word,
    // stays one sweep ahead of the device, and sizes each grid with the newest count it knows --
    // counts only shrink, so it is an upper bound -- so tail sweeps launch a handful of workgroups.
    const int C = chains_;
    const int cstride = 2 * (max_sweeps + 12);  // per chain: counts [0, max_sweeps + 2), results of the persistent kernel
                                                // [max_sweeps + 2, + 10), cursors of the list rebuilds (dense sweeps) in the second half
    {
      altro_status rs = ReserveCounters(C * cstride + 16);
      if (rs != ALTRO_OK) return rs;
    }
    if (prof) hipEventRecord(ProfEvent(nev++), stream_);
    twin_box_clean_ = false;
    if (begin_merged_) {
      // ONE launch (k_begin_solve): AL Init, SolveSetup, activation, the open-loop rollout, and the clearing of the shadow
      // columns' flags (no shadow column starts a solve as an active instance, whatever the last solve left in it: the
      // dense expansion launches rebuild their lists from these flags), of the sweep counters and of the twins' mailboxes
      DevArrays<T> As = A_;
      if (seg_total_ > 0) SegArrays(As);  // (resets the bookkeeping of the segments, DevArrays::seg_*)
      ZeroJobs z{};
      int nz = 0;
      auto zero = [&](void* ptr, size_t words) {
        if (nz >= kZeroJobs || !ptr || words == 0 || words > 0xffffffffull) return false;  // (the caller falls back to a memset)
        z.p[nz] = reinterpret_cast<unsigned*>(ptr);
        z.n[nz++] = (unsigned)words;
        return true;
      };
      if (seg_total_ + twin_cap_ > 0 && !zero(A_.phase + (Bp_ - seg_total_ - twin_cap_), (size_t)(seg_total_ + twin_cap_)))
        ALTRO_HIP_CHECK(hipMemsetAsync(A_.phase + (Bp_ - seg_total_ - twin_cap_), 0, (size_t)(seg_total_ + twin_cap_) * sizeof(int), stream_));
      if (!zero(d_counter_, (size_t)(C * cstride + 16)))
        ALTRO_HIP_CHECK(hipMemsetAsync(d_counter_, 0, (size_t)(C * cstride + 16) * sizeof(int), stream_));
      if (twin_cap_ > 0) twin_box_clean_ = zero(d_twin_box_, (size_t)twin_cap_ * (kTwWords + 1) * 2);  // (else: cleared at the launch)
      size_t words = 0;
      for (int j = 0; j < nz; ++j) words += z.n[j];
      const int gx = (B_ + kBlock - 1) / kBlock;
      const int rows_y = (mode == kFwdAL) ? std::max(1, (pd_.total_rows + kAlInitRows - 1) / kAlInitRows) : 0;
      const int zero_y = (int)std::min<size_t>(32, std::max<size_t>(1, words / ((size_t)gx * kBlock * 16)));
      hipLaunchKernelGGL((k_begin_solve<T, M>), dim3(gx, 1 + rows_y + zero_y), dim3(kBlock), 0, stream_, As, d_pd_, d,
                         mode == kFwdAL ? 1 : 0, rows_y, z);
      timing_.launches += 1;
    } else {
      if (mode == kFwdAL) hipLaunchKernelGGL(k_al_init<T>, GridAlInit(), dim3(kBlock), 0, stream_, A_, d_pd_, d);
      if (seg_total_ + twin_cap_ > 0)
        ALTRO_HIP_CHECK(hipMemsetAsync(A_.phase + (Bp_ - seg_total_ - twin_cap_), 0, (size_t)(seg_total_ + twin_cap_) * sizeof(int), stream_));
      {
        DevArrays<T> As = A_;
        if (seg_total_ > 0) SegArrays(As);
        hipLaunchKernelGGL(k_solve_setup<T>, GridB(), dim3(kBlock), 0, stream_, As, d, 1);
      }
      hipLaunchKernelGGL((k_rollout<T, M>), GridB(), dim3(kBlock), 0, stream_, A_, d_pd_, 1);
      timing_.launches += (mode == kFwdAL) ? 3 : 2;
      ALTRO_HIP_CHECK(hipMemsetAsync(d_counter_, 0, (size_t)(C * cstride + 16) * sizeof(int), stream_));
    }
    if (prof) hipEventRecord(ProfEvent(nev++), stream_);
    ALTRO_HIP_CHECK(hipGetLastError());
    // the device-side sweep loop takes the bulk phase whenever the persistent tail kernel can follow it (FusedOk: MFMA
    // backward pass, staged forward pass, uniform step, <= 20 line-search trials, no cost-to-go records)
    const bool loop_on = loop_groups_ > 0 && FusedOk(d);
    // segments of rejection streaks: not with a recorded history (its rows are appended in iteration order)
    // ... nor with cost-to-go records: every shadow column writes P, p into its own CTG column, and neither
    // k_seg_fixup nor the twins' commit copy those back -- altro_get_ctg would return the records of the backward pass at which
    // the instance's OWN column retired
    const bool seg_on = seg_total_ > 0 && !A_.hist && !A_.record_ctg && !d.fast_forward_stalls && C * kBlock <= seg_total_ && !loop_on &&
                        this->spec_mode_ == kSpecAuto;  // (the persistent kernel's variants that know the segments: default modes only)
    const int seg_capc = seg_on ? (seg_total_ / C) / kBlock * kBlock : 0;  // shadow columns per chain
    if (seg_on) ALTRO_HIP_CHECK(hipMemsetAsync(d_seg_cursor_, 0, kMaxChains * sizeof(int), stream_));
    Chain chain[kMaxChains];
    for (int c = 0; c < C; ++c) {
      Chain& ch = chain[c];
      ch.lo = C > 1 ? std::min(B_, c * chain_size_) : 0;
      ch.hi = C > 1 ? std::min(B_, (c + 1) * chain_size_) : B_;
      ch.st = c == 0 ? stream_ : chain_stream_[c];
      ch.d_cnt = d_counter_ + (size_t)c * cstride;
      ch.h_cnt = h_counter_ + (size_t)c * cstride;
      ch.h_cnt_dev = h_counter_dev_ + (size_t)c * cstride;
      ch.known = ch.hi - ch.lo;
      for (int i = 0; i < max_sweeps + 2; ++i) ch.h_cnt[i] = -1;
    }
    if (C > 1) {
      // the other chains start behind the initialisation (and the counter reset) on the engine's stream
      hipEventRecord(start_ev_, stream_);
      for (int c = 1; c < C; ++c) hipStreamWaitEvent(chain[c].st, start_ev_, 0);
    }
    for (int c = 0; c < C; ++c) {
      if (!prof) break;
      if (c == 0) {
        chain[c].ev.push_back(nev - 1);
      } else {
        hipEventRecord(ProfEvent(nev), chain[c].st);
        chain[c].ev.push_back(nev++);
      }
    }
    const bool fused_ok = FusedOk(d);
    bool tail_mode = false;  // the persistent kernel takes over: every chain stops at its next sweep boundary
    auto total_known = [&]() {
      long long t = 0;
      for (int c = 0; c < C; ++c) t += chain[c].done ? 0 : chain[c].known;
      return (int)std::min<long long>(t, B_);
    };
    auto rec = [&](Chain& ch) {
      if (!prof) return;
      hipEventRecord(ProfEvent(nev), ch.st);
      ch.ev.push_back(nev++);
    };
    // lists of a chain: its own slice [lo, ...) of the two list buffers (a chain never holds more than hi - lo instances)
    // (a chain's slice of the two list buffers: its instances and its shadow columns -- every column is listed at most once)
    auto list_of = [&](int which, const Chain& ch) { return d_list_[which] + ch.lo + (int)(&ch - chain) * seg_capc; };
    // (the kernels only see the arrays -- and pay for the bookkeeping of streaks, a few dependent words per instance and
    //  sweep -- from sweep seg_from of a chain on: a batch whose sweeps are over by then, config 2, never does)
    const int seg_from = 24;
    auto seg_fields = [&](DevArrays<T>& A, const Chain& ch, int i) {
      if (!seg_on || i < seg_from) {
        A.seg_end = nullptr;
        return;
      }
      const int c = (int)(&ch - chain);
      SegArrays(A);
      A.seg_cursor = d_seg_cursor_ + c;
      A.seg_lo = seg_col0_ + c * seg_capc;
      A.seg_hi = A.seg_lo + seg_capc;
      A.seg_parts = seg_parts_;
    };
    // may sweep i of a chain split streaks?  Only every kSegSplitEvery-th does: the host sizes the grids of sweep i + 1 from
    // the count sweep i - 1 left, and a sweep that splits is the only thing that can make a list longer than that
    // ... and only while the batch has room: a streak split while most of the batch is still iterating only makes sweeps
    // longer that the GPU already fills (the gain is in packing the plateau of the long inner solves into fewer sweeps)
    const int seg_below = (int)((long long)B_ * seg_below_pct_ / 100);
    // ... and not any more once the hand-over to the persistent kernel is near (3 x persist_at_ instances left): the
    // launch takes whatever the lists hold by then, and a list that has just grown fourfold means rounds of workgroups
    // instead of one (config 3: 32.2 - 32.6 ms with this bound, 33 - 35 ms with 1.5 x).  Once a chain has split, the
    // hand-over itself moves to 1.5 x persist_at_: what is left when the segments retire are the long runners, whose
    // iterations the persistent kernel runs at a fifth of a batched sweep's latency.
    const int seg_above = std::min(3 * persist_at_, B_ * 3 / 8);  // (a batch of 2048 keeps a window below its 65 %)
    const int seg_persist_at = persist_at_ * 3 / 2;
    const int seg_every = kSegSplitEvery;
    auto splits_in = [&](int i) {
      return seg_on && i >= seg_from + 2 && (i % seg_every) == 0 && total_known() <= seg_below && total_known() > seg_above;
    };
    // upper bound of the instances a sweep may meet, given the newest count the host knows: a streak that splits adds
    // seg_parts - 1 columns per instance between that count and the sweep being enqueued
    auto bound_of = [&](const Chain& ch) {  // ... of the sweep about to be enqueued: what the one before it left
      const int known = std::max(1, ch.known);
      if (!ch.last_split) return known;
      return (int)std::min<long long>((long long)(ch.hi - ch.lo) + seg_capc, (long long)known * seg_parts_);
    };
    auto any_split = [&]() {
      bool any = false;
      for (int c = 0; c < C; ++c) any = any || chain[c].split_seen;
      return seg_on && any;
    };
    auto enqueue_sweep = [&](Chain& ch, int i) -> altro_status {
      DevArrays<T> A = A_;
      A.chain_lo = ch.lo;
      A.chain_hi = (C > 1 || seg_on) ? ch.hi : 0;
      seg_fields(A, ch, i);
      if (i == 0) {
        A.act_list = C > 1 ? d_iota_ + ch.lo : nullptr;
        A.act_count = nullptr;
        A.act_count_const = ch.hi - ch.lo;
        A.host_count = nullptr;
      } else {
        A.act_list = list_of(i % 2, ch);
        A.act_count = ch.d_cnt + (i - 1);
        A.host_count = ch.h_cnt_dev + (i - 1);
      }
      A.next_list = list_of((i + 1) % 2, ch);
      A.next_count = ch.d_cnt + i;
      const int ninst = i == 0 ? std::max(1, ch.known) : bound_of(ch);
      const bool may_split = splits_in(i);
      if (!may_split) A.seg_parts = 1;
      // (a batch that fits the CUs from the start -- the MPC case, one or a few dozen instances -- goes to the persistent
      //  kernel at once: its first sweep as three launches and a host round trip would only add latency)
      if (fused_ok && (i > 0 || (C == 1 && !no_fused_first_)) && (tail_mode || total_known() <= (any_split() ? seg_persist_at : persist_at_))) {
        // the tail: every instance left gets a workgroup that runs whole iterations (k_sweep_fused, launched below
        // for all chains together); this chain's list is the one sweep i would have worked on
        tail_mode = true;
        ch.tail = true;
        return ALTRO_OK;
      }
      cur_ = ch.st;
      const int span = ch.hi - ch.lo;
      const dim3 gridB((ninst + kBlock - 1) / kBlock);
      PoisonLds();
      // (small models only: their expansions are HBM-bound; the 12-state model's are compute-bound and pay for the idle
      //  lanes of a dense launch: config 4 +13 %)
      if (i > 0 && dense_expansions_ && !kKdgEligible && 4 * (long long)std::max(1, ch.known) >= span) {  // (thresholds 1/2 ... 1/16 measured: 1/4 ... 1/8 best)
        // a good part of the chain is still iterating: lane = instance (coalesced rows and records), and the list is
        // rebuilt in runs of neighbouring instances for the two kernels that follow (see k_expansions); the launch also
        // covers the chain's slice of shadow columns (instance_of_slot)
        hipLaunchKernelGGL((k_expansions<T, M>), dim3((span + (ch.split_seen ? seg_capc : 0) + kBlock - 1) / kBlock, N_ + 1), dim3(kBlock), 0, ch.st, A, d_pd_, 2,
                           list_of(i % 2, ch), ch.d_cnt + (max_sweeps + 12) + i);
      } else {
        hipLaunchKernelGGL((k_expansions<T, M>), dim3(gridB.x, N_ + 1), dim3(kBlock), 0, ch.st, A, d_pd_, 0, (int*)nullptr,
                           (int*)nullptr);
      }
      rec(ch);
      LaunchBackward(A, d, 0, ninst);
      rec(ch);
      LaunchForward(A, d, mode, 0, ninst, total_known());
      rec(ch);
      cur_ = stream_;
      ch.last_split = may_split;
      ch.split_seen = ch.split_seen || may_split;
      timing_.launches += 3;
      timing_.sweep_launches += 1;
      return ALTRO_OK;
    };
    bool loop_launched = false;
    size_t loop_ev = 0;
    const int loop_handover = std::min(B_, persist_at_);
    if (loop_on) {
      // ONE launch instead of the chains of sweeps: persistent workgroups pull instances, run E -> B -> F for them until
      // they finish, and leave what is unfinished when `loop_handover` instances are left to the persistent tail kernel,
      // which is enqueued right behind (the host never looks in between)
      if constexpr (kMfmaBackward) {
        ALTRO_HIP_CHECK(hipMemsetAsync(d_loop_ctl_, 0, (size_t)(kLwWords + 8) * sizeof(int), stream_));
        const int per_xcd = ((B_ + kLoopXcds - 1) / kLoopXcds + 15) / 16 * 16;
        const LoopCtl lc{d_loop_win_, d_loop_ctl_, d_loop_tail_, loop_handover, per_xcd};
        PoisonLds();
        if (prof) {
          loop_ev = nev;
          hipEventRecord(ProfEvent(nev++), stream_);
        }
        const dim3 gl(loop_groups_), bl(kFwdWaves * kBlock);
        if (rg_) {
          if constexpr (kRgEligible)
            hipLaunchKernelGGL((k_sweep_loop<T, M, kSrcGlb>), gl, bl, loop_lds_bytes_, stream_, A_, d_pd_, pd_, d, mode, lc);
        } else {
          hipLaunchKernelGGL((k_sweep_loop<T, M, kSrcLds>), gl, bl, loop_lds_bytes_, stream_, A_, d_pd_, pd_, d, mode, lc);
        }
        if (prof) hipEventRecord(ProfEvent(nev++), stream_);
        timing_.launches += 1;
        timing_.sweep_launches += 1;
        for (int c = 0; c < C; ++c) chain[c].done = true;
        tail_mode = true;
        loop_launched = true;
      }
    }
    for (int c = 0; c < C && !loop_launched; ++c) {
      altro_status st = enqueue_sweep(chain[c], 0);
      if (st != ALTRO_OK) return st;
      if (!chain[c].tail) chain[c].sweeps = 1;  // (tail already: the persistent kernel takes the batch from its first sweep)
    }
    // Every chain: enqueue sweep s, then wait for the count sweep s-1 left (published by sweep s, which is already
    // enqueued), then enqueue sweep s+1 ... -- the chains polled in turn, none of them ever blocking the others.
    // The host waits for words the device writes into pinned memory.  It is ONE SWEEP AHEAD -- the sweep it is about to
    // enqueue only has to be in the queue before the one in flight ends, a hundred microseconds or more away -- so it
    // need not burn a core on the poll: a short spin (the word usually lands within microseconds of a neighbouring
    // chain's), then naps of ~50 us (eight ranks of a node, or the worker threads of libaltro_group.so, share the
    // container's CPU quota with RCCL's proxy threads).  A small batch -- the latency path -- keeps spinning.
    const bool nap_ok = host_wait_backoff_ && B_ >= kHostNapMinBatch;
    bool nap = false;  // (only once the solve has run for kHostNapAfterUs: a 0.3 ms solve of 1024 small problems keeps the spin)
    unsigned spins = 0, naps = 0;
    for (;;) {
      bool any = false, progressed = false;
      for (int c = 0; c < C; ++c) {
        Chain& ch = chain[c];
        if (ch.done || ch.tail) continue;
        any = true;
        if (ch.waiting) {
          const int v = __atomic_load_n(&ch.h_cnt[ch.sweeps - 2], __ATOMIC_ACQUIRE);
          if (v < 0) continue;
          ch.known = v;
          ch.waiting = false;
          progressed = true;
          if (v == 0) {
            ch.done = true;
            continue;
          }
        }
        if (ch.sweeps >= max_sweeps) {  // the iteration caps bound the sweeps; nothing left to learn
          ch.done = true;
          progressed = true;
          continue;
        }
        altro_status st = enqueue_sweep(ch, ch.sweeps);  // publishes the count left by sweep (sweeps - 1)
        if (st != ALTRO_OK) return st;
        progressed = true;
        if (ch.tail) continue;  // (not enqueued: the persistent kernel comes instead)
        ch.sweeps++;
        ch.waiting = true;
      }
      if (!any) break;
      bool check_streams = false;
      if (nap_ok && !nap && (spins & 0xff) == 0)
        nap = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > kHostNapAfterUs;
      if (progressed) {
        spins = 0;
        naps = 0;
      } else if (nap && spins >= kHostSpinsBeforeNap) {
        std::this_thread::sleep_for(std::chrono::microseconds(20));  // (timer slack makes it ~70 us)
        timing_.host_naps += 1;
        check_streams = (++naps & 0xf) == 0;
      } else {
        CpuRelax();
        check_streams = (++spins & 0x3ff) == 0 && !nap;
      }
      if (check_streams) {
        for (int c = 0; c < C; ++c) {
          Chain& ch = chain[c];
          if (ch.done || ch.tail || !ch.waiting) continue;
          const hipError_t q = hipStreamQuery(ch.st);
          if (q == hipSuccess) {  // stream drained: the word must be there, or the launch failed
            if (__atomic_load_n(&ch.h_cnt[ch.sweeps - 2], __ATOMIC_ACQUIRE) >= 0) continue;
            ALTRO_HIP_CHECK(hipGetLastError());
            err_ = "sweep counter was never published";
            return ALTRO_HIP_ERROR;
          }
          if (q != hipErrorNotReady) ALTRO_HIP_CHECK(q);
        }
      }
    }
    bool persistent_launched = false;
    size_t fused_ev = 0;
    if (tail_mode) {
      if constexpr (kMfmaBackward) {
        DevArrays<T> A = A_;
        ChainLists lists{};
        long long ninst_l = 0;
        bool whole_batch = false;
        A.chain_size = C > 1 ? chain_size_ : 0;
        for (int c = 0; c < C; ++c) {
          Chain& ch = chain[c];
          A.chain_base[c] = ch.sweeps;  // batched sweeps this chain ran (0 .. sweeps-1)
          if (!ch.tail) continue;
          if (ch.sweeps == 0) {  // (single chain, no batched sweep ran: every instance, in order)
            whole_batch = true;
            ninst_l += B_;
            continue;
          }
          lists.list[lists.n] = list_of(ch.sweeps % 2, ch);
          lists.count[lists.n] = ch.d_cnt + (ch.sweeps - 1);
          lists.n++;
          ninst_l += bound_of(ch);
          if (ch.st != stream_) {  // the engine's stream continues behind this chain's last sweep
            hipEventRecord(chain_ev_[c], ch.st);
            hipStreamWaitEvent(stream_, chain_ev_[c], 0);
          }
        }
        if (loop_launched) {
          ninst_l = loop_handover;  // (an upper bound: the list's length stays on the device)
          A.chain_size = 0;
        }
        const int ninst = (int)std::min<long long>(std::max<long long>(ninst_l, 1), (long long)B_ + (seg_on ? seg_total_ : 0));
        if (any_split()) {
          SegArrays(A);
          A.seg_lo = seg_col0_;  // (this launch: first shadow column / columns per chain, for the report of a handed-over column)
          A.seg_hi = seg_capc;
        }
        if (loop_launched) {
          A.act_list = d_loop_tail_;
          A.act_count = d_loop_ctl_ + kLwTail;
        } else if (whole_batch) {
          A.act_list = nullptr;
          A.act_count = nullptr;
          A.act_count_const = B_;
        } else if (lists.n == 1) {
          A.act_list = lists.list[0];
          A.act_count = lists.count[0];
        } else {
          int* const merged_count = d_counter_ + (size_t)C * cstride;
          hipLaunchKernelGGL((k_merge_lists<0>), dim3(1), dim3(256), 0, stream_, lists, d_merged_, merged_count);
          A.act_list = d_merged_;
          A.act_count = merged_count;
          timing_.launches += 1;
        }
        A.host_count = nullptr;
        A.next_list = nullptr;  // nobody comes after this launch
        A.next_count = nullptr;
        PoisonLds();
        if (prof) {
          fused_ev = nev;
          hipEventRecord(ProfEvent(nev++), stream_);
        }
        // (the variant without the circle layouts for problems that have no circle constraint: see forward2_body)
        bool circles = false;
        for (int r = 0; r < pd_.nruns; ++r)
          circles = circles || pd_.runs[r].fast == kFastC || pd_.runs[r].fast == kFastCB || pd_.runs[r].fast == kFastBC;
        // (with a fourth wave that runs the next iteration's backward pass beside the forward pass: see the kernel)
        int* const out = loop_launched ? d_loop_ctl_ + kLwWords : d_counter_ + max_sweeps + 2;
        const int spec_mode_ = this->spec_mode_ == kSpecAuto ? (circles ? (int)kSpecWave : (int)kSpecFree) : this->spec_mode_;
        // twin workgroups (TwinCtl): one behind every primary, dispatched after all of them (same launch, higher block
        // indices); not with a recorded history (its rows are appended in iteration order)
        TwinCtl tw{};
        if (twin_cap_ > 0 && !A.hist && !d.fast_forward_stalls) {
          if (!twin_box_clean_)  // (cleared by k_begin_solve otherwise: nothing touches the mailboxes before this launch)
            hipMemsetAsync(d_twin_box_, 0, (size_t)twin_cap_ * (kTwWords + 1) * sizeof(unsigned long long), stream_);
          tw = TwinCtl{d_twin_box_, d_twin_box_ + (size_t)twin_cap_ * kTwWords, ninst, twin_cap_, Bp_ - twin_cap_, kTwinLag, twin_debug_ ? 1 : 0};
        }
        const dim3 g(ninst + (tw.base > 0 ? std::min(ninst, twin_cap_) : 0)), b3(kFwdWaves * kBlock), b4((kFwdWaves + 1) * kBlock);
        timing_.twin_workgroups = tw.base > 0 ? (int)g.x - ninst : 0;  // twin workgroups of this launch
#define ALTRO_FUSED(CC, S, BLK) \
  hipLaunchKernelGGL((k_sweep_fused<T, M, CC, S>), g, BLK, fused_lds_bytes_, stream_, A, d_pd_, pd_, d, mode, 1, out, tw)
        if (any_split()) {
          // columns of split streaks may be in the lists: the variants that verify / retire / cancel them in the loop's
          // bookkeeping step (every other launch runs the kernels as they were before the segments existed)
          if (circles)
            hipLaunchKernelGGL((k_sweep_fused<T, M, true, kSpecWave, true>), g, b4, fused_lds_bytes_, stream_, A, d_pd_, pd_, d, mode, 1, out, tw);
          else
            hipLaunchKernelGGL((k_sweep_fused<T, M, false, kSpecFree, true>), g, b4, fused_lds_bytes_, stream_, A, d_pd_, pd_, d, mode, 1, out, tw);
        } else if (circles) {
          if (spec_mode_ == kSpecWave) ALTRO_FUSED(true, kSpecWave, b4);
          else if (spec_mode_ == kSpecFree) ALTRO_FUSED(true, kSpecFree, b4);
          else ALTRO_FUSED(true, kSpecOff, b3);
        } else {
          if (spec_mode_ == kSpecWave) ALTRO_FUSED(false, kSpecWave, b4);
          else if (spec_mode_ == kSpecFree) ALTRO_FUSED(false, kSpecFree, b4);
          else ALTRO_FUSED(false, kSpecOff, b3);
        }
#undef ALTRO_FUSED
        if (prof) hipEventRecord(ProfEvent(nev++), stream_);
        persistent_launched = true;
        timing_.launches += 1;
      }
    }
    if (nap_ok && !nap) nap = loop_launched || std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > kHostNapAfterUs;
    if (nap && tail_ev_) {
      // the persistent launch (or the device-side loop) runs for milliseconds: wait for it WITHOUT a core.  Round 4 waited in
      // hipEventSynchronize on a blocking-sync event; measured in round 6 (bench.py's device_loop key: one launch per solve, the
      // host does nothing else) that wait still costs a whole core -- the runtime spins on the signal.  So: query the event
      // and nap (~70 us with the timer slack: < 2 % of the solves that get here -- large batches only; the latency path
      // below kHostNapMinBatch instances keeps hipStreamSynchronize).
      ALTRO_HIP_CHECK(hipEventRecord(tail_ev_, stream_));
      for (;;) {
        const hipError_t q = hipEventQuery(tail_ev_);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) ALTRO_HIP_CHECK(q);
        std::this_thread::sleep_for(std::chrono::microseconds(20));
        timing_.host_naps += 1;
      }
    }
    for (int c = 1; c < C; ++c) ALTRO_HIP_CHECK(hipStreamSynchronize(chain[c].st));
    ALTRO_HIP_CHECK(hipStreamSynchronize(stream_));
    ALTRO_HIP_CHECK(hipGetLastError());
    if (any_split()) {
      // chains of segments: the last valid column of each over the instance's own (k_seg_fixup; a workgroup per instance,
      // all but the split ones leave at once)
      if constexpr (kMfmaBackward) {
        DevArrays<T> As = A_;
        SegArrays(As);
        hipLaunchKernelGGL((k_seg_fixup<T, M>), dim3(B_), dim3(kBlock), 0, stream_, As, d_pd_);
        timing_.launches += 1;
        int cur[kMaxChains] = {0};
        ALTRO_HIP_CHECK(CopySync(cur, d_seg_cursor_, sizeof(cur), hipMemcpyDeviceToHost));  // (synchronises the stream)
        for (int c = 0; c < C; ++c) timing_.segment_columns += std::min(cur[c], seg_capc);
      }
    }
    int sweeps = 0;  // longest chain of iterations, the look-ahead sweep of a chain that ran dry included
    for (int c = 0; c < C; ++c) sweeps = std::max(sweeps, chain[c].sweeps);
    if (persistent_launched) {
      int extra[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (loop_launched) {
        int words[kLwWords + 8];
        ALTRO_HIP_CHECK(CopySync(words, d_loop_ctl_, sizeof(words), hipMemcpyDeviceToHost));
        std::memcpy(extra, words + kLwWords, sizeof(extra));
        timing_.loop_workgroups = words[kLwGroups];
        timing_.loop_instance_iterations = words[kLwUnits];
        timing_.loop_handover = words[kLwTail];
        timing_.loop_iterations = words[kLwMaxLoops];
        sweeps = words[kLwMaxLoops];
        extra[2] += words[kLwMaxLoops];
        if (loop_log_) {
          const double per = words[kLwGroups] > 0 ? 0.01 / words[kLwGroups] : 0.0;  // 100 MHz ticks -> us per workgroup
          fprintf(stderr, "LOOPLOG %d workgroups (%d per CU), %d units, longest %d iterations, handed over %d | us per workgroup: slots %.1f E %.1f B %.1f F %.1f\n",
                  words[kLwGroups], loop_per_cu_, words[kLwUnits], words[kLwMaxLoops], words[kLwTail], per * words[kLwTicks],
                  per * words[kLwTicks + 1], per * words[kLwTicks + 2], per * words[kLwTicks + 3]);
        }
      } else {
        ALTRO_HIP_CHECK(CopySync(extra, d_counter_ + max_sweeps + 2, sizeof(extra), hipMemcpyDeviceToHost));
      }
      timing_.twin_handovers = extra[4];
      timing_.twin_claims = extra[5];
      timing_.fused_workgroup_iterations = extra[6];
      if (seg_on && twin_debug_) {
        int cur[kMaxChains] = {0};
        ALTRO_HIP_CHECK(CopySync(cur, d_seg_cursor_, sizeof(cur), hipMemcpyDeviceToHost));
        fprintf(stderr, "segments: shadow columns used per chain %d %d %d %d (of %d each), persistent launch over <= %d slots, longest own chain %d, longest workgroup %d\n",
                cur[0], cur[1], cur[2], cur[3], seg_capc, (int)timing_.twin_workgroups, extra[0], extra[6]);
      }
      if (twin_cap_ > 0 && twin_debug_) {  // the mailboxes after the launch, slot by slot
        std::vector<unsigned long long> box((size_t)twin_cap_ * kTwWords);
        ALTRO_HIP_CHECK(CopySync(box.data(), d_twin_box_, box.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        int hist[4] = {0, 0, 0, 0}, why[8] = {0};
        unsigned long long t0 = ~0ull;
        for (int s = 0; s < twin_cap_; ++s)
          if (box[(size_t)s * kTwWords + kTwStamp + kTsPStart]) t0 = std::min(t0, box[(size_t)s * kTwWords + kTwStamp + kTsPStart]);
        int shown = 0;
        for (int s = 0; s < twin_cap_; ++s) {
          const unsigned long long* w = &box[(size_t)s * kTwWords];
          hist[w[kTwHand] & 3]++;
          if ((w[kTwHand] & 3) == kTwOk && (shown++ % 8) == 0) {  // timelines of every eighth hand-over, us from the first workgroup's start
            const unsigned long long* ts = w + kTwStamp;
            auto us = [&](int i) { return ts[i] ? (double)(long long)(ts[i] - t0) * 0.01 : -1.0; };
            fprintf(stderr, "  slot %3d  P: start %.0f snap %.0f (loop %llu) claim %.0f hand %.0f (loops %llu) | T: start %.0f go %.0f cloned %.0f first %.0f done %.0f (loops %llu) "
                    "verdict %.0f commit %.0f | claim start it %d\n", s, us(kTsPStart), us(kTsPSnap), ts[kTsPLoopsAtSnap], us(kTsPClaim), us(kTsPHand),
                    w[kTwHandLoops], us(kTsTStart), us(kTsTGo), us(kTsTCloned), us(kTsTFirst), us(kTsTDone), ts[kTsTLoops], us(kTsTVerdict), us(kTsTCommit),
                    (int)(w[kTwClaim] >> 32));
          }
          if (w[kTwClaim] != 0 && (w[kTwHand] & 3) == kTwRefused) {
            why[w[kTwWhy] & 7]++;
            if (why[w[kTwWhy] & 7] <= 3)
              fprintf(stderr, "  twin slot %d refused: why %d at it_inner %d (claim start %d, snapshot total %d), snapshots %llu\n", s, (int)(w[kTwWhy] & 7),
                      (int)((w[kTwWhy] >> 8) & 0xffff), (int)((w[kTwWhy] >> 24) & 0xffff), (int)w[kTwClaimSnap], w[kTwSeq]);
          }
        }
        {
          double last_commit = 0, last_pend = 0, last_tstart = 0;
          int s_commit = -1, s_pend = -1;
          for (int s = 0; s < twin_cap_; ++s) {
            const unsigned long long* ts = &box[(size_t)s * kTwWords + kTwStamp];
            auto us = [&](int i) { return ts[i] ? (double)(long long)(ts[i] - t0) * 0.01 : -1.0; };
            if (us(kTsTCommit) > last_commit) { last_commit = us(kTsTCommit); s_commit = s; }
            if (us(kTsPEnd) > last_pend) { last_pend = us(kTsPEnd); s_pend = s; }
            last_tstart = std::max(last_tstart, us(kTsTStart));
          }
          fprintf(stderr, "twin timeline: last commit %.0f us (slot %d), last primary that finished by itself %.0f us (slot %d), last twin start %.0f us\n",
                  last_commit, s_commit, last_pend, s_pend, last_tstart);
        }
        fprintf(stderr, "twin mailboxes: launched %d claims %d handovers %d | hand word: open %d ok %d refused %d revoked %d | refusals of claims: "
                "streak broke %d, passed %d, counters %d, rho %d, drho %d, break since snapshot %d\n", timing_.twin_workgroups,
                timing_.twin_claims, timing_.twin_handovers, hist[0], hist[1], hist[2], hist[3], why[1], why[2], why[3], why[4], why[5], why[6]);
      }
      if (extra[3] != 0) {
        err_ = "k_sweep_fused: a forward wave gave up waiting for its sequence word (software synchronisation)";
        return ALTRO_HIP_ERROR;
      }
      timing_.fused_sweeps = extra[0];
      timing_.fused_instance_iterations = extra[1];
      sweeps = std::max(sweeps, extra[2]);
    }
    timing_.sweeps = sweeps;
    if (prof) {
      float ms = 0;
      hipEventElapsedTime(&ms, prof_ev_[0], prof_ev_[1]);
      timing_.init_ms = ms;
      for (int c = 0; c < C; ++c) {
        const std::vector<size_t>& ev = chain[c].ev;
        for (size_t e0 = 0; e0 + 3 < ev.size(); e0 += 3) {
          hipEventElapsedTime(&ms, prof_ev_[ev[e0]], prof_ev_[ev[e0 + 1]]);
          timing_.expansions_ms += ms;
          hipEventElapsedTime(&ms, prof_ev_[ev[e0 + 1]], prof_ev_[ev[e0 + 2]]);
          timing_.backward_pass_ms += ms;
          hipEventElapsedTime(&ms, prof_ev_[ev[e0 + 2]], prof_ev_[ev[e0 + 3]]);
          timing_.forward_pass_ms += ms;
        }
      }
      if (persistent_launched) {
        hipEventElapsedTime(&ms, prof_ev_[fused_ev], prof_ev_[fused_ev + 1]);
        timing_.fused_ms += ms;
      }
      if (loop_launched) {
        float lms = 0;
        hipEventElapsedTime(&lms, prof_ev_[loop_ev], prof_ev_[loop_ev + 1]);
        timing_.loop_ms = lms;
      }
      if (sweep_log_) {
        // timeline of the chains of sweeps (diagnostics): per sweep its start since the solve began, the length of the list it
        // worked on (what the sweep before it left) and the durations of its three kernels
        for (int c = 0; c < C; ++c) {
          const std::vector<size_t>& ev = chain[c].ev;
          int idx = 0;
          for (size_t e0 = 0; e0 + 3 < ev.size(); e0 += 3, ++idx) {
            float t = 0, e = 0, b = 0, f = 0;
            hipEventElapsedTime(&t, prof_ev_[0], prof_ev_[ev[e0]]);
            hipEventElapsedTime(&e, prof_ev_[ev[e0]], prof_ev_[ev[e0 + 1]]);
            hipEventElapsedTime(&b, prof_ev_[ev[e0 + 1]], prof_ev_[ev[e0 + 2]]);
            hipEventElapsedTime(&f, prof_ev_[ev[e0 + 2]], prof_ev_[ev[e0 + 3]]);
            if (idx % 4 == 0 || e0 + 6 >= ev.size() || std::string(sweep_log_) == "all")
              fprintf(stderr, "SWEEPLOG chain %d sweep %3d t %7.3f ms list %5d E %6.1f B %6.1f F %6.1f us\n", c, idx, t,
                      idx == 0 ? chain[c].hi - chain[c].lo : (int)chain[c].h_cnt[idx - 1], 1e3 * e, 1e3 * b, 1e3 * f);
          }
        }
        if (persistent_launched) {
          float t = 0;
          hipEventElapsedTime(&t, prof_ev_[0], prof_ev_[fused_ev]);
          fprintf(stderr, "SWEEPLOG persistent launch t %7.3f ms, %.3f ms, slots <= %d\n", t, ms, (int)timing_.twin_workgroups);
        }
      }
    }
    timing_.instance_iterations = -1;  // summed on demand (GetTiming): keeps a copy out of every solve
    timing_.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return ALTRO_OK;
  }

  static constexpr int kNumScalarT = 15, kNumScalarI = 7;
  // (time-varying dynamics, user discrete dynamics and per-knot model lists: every knot has its own (step, time, model))
  static constexpr bool kTimeVarying = model_knot_path<M>::value;
  static constexpr int kModelCount = model_list<M>::type::size;
  float *d_hk_ = nullptr, *d_tk_ = nullptr;  // per-knot steps / times (SetKnotTimes), owned through allocs_
  altro_desc desc_;
  int B_ = 0, Bp_ = 0, N_ = 0;
  bool uploaded_ = false;
  bool last_mode_ilqr_ = false;
  DevOpts last_opts_{};          // options of the last whole solve (ReplayCtg)
  bool ctg_fresh_ = false;       // the cost-to-go records are those of the last backward pass
  bool ctg_replayable_ = false;  // ... or can be recomputed from what the last whole solve left in memory
  ProblemDesc pd_{};
  ProblemDesc* d_pd_ = nullptr;
  DevArrays<T> A_{};
  double* d_tmp_ = nullptr;
  struct SegPtrs {
    int *end = nullptr, *next = nullptr, *flag = nullptr, *streak = nullptr, *tot0 = nullptr;
    double *rho0 = nullptr, *drho0 = nullptr;
  } seg_;
  void SegArrays(DevArrays<T>& A) const {  // the bookkeeping arrays of the segments into a launch's copy of A_
    A.seg_end = seg_.end; A.seg_next = seg_.next; A.seg_flag = seg_.flag; A.seg_streak = seg_.streak;
    A.seg_tot0 = seg_.tot0; A.seg_rho0 = seg_.rho0; A.seg_drho0 = seg_.drho0;
  }
  int seg_total_ = 0, seg_col0_ = 0, seg_parts_ = 4;  // shadow columns of the batched sweeps' segments (DevArrays::seg_*)
  int* d_seg_cursor_ = nullptr;                       // next free column of each chain's slice
  // diagnostics, read once (no getenv on the solve path): mailbox dump of the twins, timeline of the chains of sweeps, phase log of the loop
  const bool twin_debug_ = std::getenv("ALTRO_HIP_TWIN_DEBUG") != nullptr;
  const char* const sweep_log_ = std::getenv("ALTRO_HIP_SWEEP_LOG");
  const bool loop_log_ = std::getenv("ALTRO_HIP_LOOP_LOG") != nullptr;
  int seg_below_pct_ = 65;  // split only below this share of the batch
  int twin_cap_ = 0;                          // shadow columns behind the batch (twin workgroups of the persistent kernel)
  unsigned long long* d_twin_box_ = nullptr;  // their mailboxes, [twin_cap_][kTwWords]
  int* d_list_[2] = {nullptr, nullptr};
  bool mfma_offsets_ok_ = false;
  // ALTRO_HIP_BACKWARD = valu | coop selects a fallback backward kernel (tests)
  static bool BackwardEnvIs(const char* what) {
    const char* e = std::getenv("ALTRO_HIP_BACKWARD");
    return e && std::string(e) == what;
  }
  bool force_valu_backward_ = BackwardEnvIs("valu");
  bool force_coop_backward_ = BackwardEnvIs("coop");
  bool dense_expansions_ = std::getenv("ALTRO_HIP_NO_DENSE_EXPANSIONS") == nullptr;
  // speculative backward pass of the persistent kernel: on its fourth wave (default) or not at all (ALTRO_HIP_SPECULATION=off)
  // (default, kSpecAuto: the fourth wave -- free-running beside software-synchronised forward waves where the rollout
  //  wave paces the knot loop, in lock step on problems with circle constraints, whose knot loop is paced by the cost
  //  wave: there the sequence words only add their polls.  Measured per tail iteration: config 2 46.0 -> 42.0 us free,
  //  config 3 54.7 lock step against 57.0 us free.)
  static constexpr int kSpecAuto = -1;
  int spec_mode_ = [] {
    const char* e = std::getenv("ALTRO_HIP_SPECULATION");
    if (e && std::string(e) == "off") return (int)kSpecOff;
    if (e && std::string(e) == "free") return (int)kSpecFree;
    if (e && std::string(e) == "wave") return (int)kSpecWave;
    return kSpecAuto;
  }();
  bool kdg_ = false;  // forward pass reads the feedback gains from global memory (k_forward2<.., kSrcKdg>)
  bool rg_ = false;   // ... all of the rollout wave's per-knot inputs (k_forward2<.., kSrcGlb>)
  int fwd_per_wave_ = kBlock / kLineSearchLanes;
  size_t fwd_lds_bytes_ = 0, fwd_shared_bytes_ = 0, fwd_per_inst_bytes_ = 0;
  int num_cus_ = 256;
  bool fast_forward_ = std::getenv("ALTRO_HIP_FAST_FORWARD_STALLS") != nullptr;
  // the start of a solve as one launch (k_begin_solve); ALTRO_HIP_BEGIN_SOLVE=split restores the seven stream operations
  // (the bit-identity test of the two, tests/test_fused_gpu.py)
  const bool begin_merged_ = [] {
    const char* e = std::getenv("ALTRO_HIP_BEGIN_SOLVE");
    return !(e && std::strcmp(e, "split") == 0);
  }();
  bool twin_box_clean_ = false;  // this solve's k_begin_solve has cleared the twins' mailboxes
  double* d_stage_ = nullptr;
  size_t stage_cap_ = 0;
  static constexpr int kMaxChains = kMaxSweepChains;
  static constexpr int kDefaultChains = 4;
  bool counted_chained_ = false;
  long long chain_overlap_ticks_ = 0;  // result of the side-by-side check of the chain streams (100 MHz ticks)
  int chains_ = 1;      // chains of batched sweeps (see Chain); ALTRO_HIP_CHAINS overrides
  int chain_size_ = 0;  // instances per chain (a multiple of the wavefront size)
  hipStream_t chain_stream_[kMaxChains] = {};
  hipEvent_t chain_ev_[kMaxChains] = {};
  hipEvent_t start_ev_ = nullptr;
  hipEvent_t tail_ev_ = nullptr;  // blocking-sync event behind the last launch of a large batch's solve
  // host side of the sweep loop: spin briefly, then nap (see Solve)
  static constexpr int kHostNapMinBatch = 256;
  static constexpr unsigned kHostSpinsBeforeNap = 400;  // ~10 us of pause instructions
  static constexpr double kHostNapAfterUs = 500.0;      // a solve shorter than this never naps
  bool host_wait_backoff_ = true;
  // the device-side sweep loop (k_sweep_loop): persistent workgroups, their windows, the control words, the tail list
  int loop_groups_ = 0, loop_per_cu_ = 0;
  size_t loop_lds_bytes_ = 0;
  int *d_loop_win_ = nullptr, *d_loop_ctl_ = nullptr, *d_loop_tail_ = nullptr;
  static constexpr int lanes_per_wave() { return kBlock / kLineSearchLanes; }
  template <int SRC>
  static const void* LoopKernel() {
    if constexpr (kMfmaBackward && (SRC != kSrcGlb || kRgEligible))
      return reinterpret_cast<const void*>(&k_sweep_loop<T, M, SRC>);
    else
      return nullptr;
  }
  int* d_iota_ = nullptr;    // 0, 1, 2, ...: the active list of a chain's first sweep
  int* d_merged_ = nullptr;  // the lists of all chains, concatenated for the persistent kernel
  hipStream_t cur_ = nullptr;  // the stream the launch helpers enqueue on (a chain's, otherwise the engine's)
  bool poison_on_ = false;
  unsigned poison_pattern_ = 0;
  int poison_mix_ = 0;
  int fwd_single_at_ = 512;  // active instances (all chains) below which every instance gets a forward workgroup of its own
  int persist_at_ = 256;  // active instances at which the persistent tail kernel takes over
  size_t fused_lds_bytes_ = 0;
  bool no_fused_ = std::getenv("ALTRO_HIP_NO_FUSED_SWEEP") != nullptr;
  bool no_fused_first_ = false;  // (A/B builds: first sweep of a small batch as three launches)
  T *X_init_ = nullptr, *U_init_ = nullptr;
  double* d_scalarT_ = nullptr;
  int* d_scalarI_ = nullptr;
  double* d_phi_ = nullptr;
  std::vector<int> knot_class_, knot_rowbase_;
  std::vector<void*> allocs_;
  hipStream_t stream_ = nullptr;
  volatile int* h_counter_ = nullptr;  // pinned + mapped: one word per sweep
  int* h_counter_dev_ = nullptr;       // the same words as the device sees them
  int* d_counter_ = nullptr;
  int counter_cap_ = 0;
  std::vector<hipEvent_t> prof_ev_;
  altro_timing timing_{};
  std::string err_;
};

template <class T, class M>
EngineBase* MakeEngineImpl(const altro_desc& d, std::string* err) {
  if (d.n != M::n || d.m != M::m) {
    if (err) *err = "state/control dimensions do not match the model";
    return nullptr;
  }
  auto* e = new Engine<T, M>(d);
  if (e->Init() != ALTRO_OK) {
    if (err) *err = e->LastError();
    delete e;
    return nullptr;
  }
  return e;
}

}  // namespace altro_hip
