// altro_kernels.hpp — HIP kernels of the batched AL-iLQR solver (gfx950 / MI355X).
//
// One batched iLQR "sweep" = three launches, each advancing EVERY still-active instance by one
// inner iteration of altro::ilqr::iLQR<n,m>::Solve (altro/ilqr/ilqr.hpp:300-313):
//
//   k_expansions   grid (instance x knot): cost/AL expansion + RK4 Jacobian + knot cost.
//                  Embarrassingly parallel (ilqr.hpp:670-677).
//   k_backward     one lane per instance, serial in k: Riccati recursion with the reference's
//                  restart-on-Cholesky-failure schedule (ilqr.hpp:385-445).  Next knot's
//                  expansion is prefetched into registers while the current knot is computed.
//   k_forward      SPECULATIVE PARALLEL LINE SEARCH: the (up to) 20 backtracking trials of
//                  ilqr.hpp:525-545 are independent closed-loop rollouts, so each instance gets 20
//                  lanes (3 instances per wavefront) that evaluate alpha = 1, 1/2, ... 2^-19 side
//                  by side; a wave ballot picks the first trial the serial loop would have
//                  accepted, so the result is identical to the reference's sequential search.  The
//                  winner is then replayed by one lane, writing the new trajectory in place, and
//                  the same lane runs the per-instance state machine: convergence statistics,
//                  IsDone, dual/penalty update and the AL outer-loop transition
//                  (ilqr.hpp:568-619, al_solver.hpp:313-401).
//
// Instances are independent; a finished instance (phase == 0) simply masks its lanes.
#pragma once

#include "altro_device.hpp"

namespace altro_hip {

constexpr int kBlock = 64;  // one wavefront per workgroup: instances never share data

enum ForwardMode { kFwdStepOnly = 0, kFwdILQR = 1, kFwdAL = 2 };

template <class T>
ALTRO_DEV void hist_push(const DevArrays<T>& A, int b) {
  // SolverStats::NewIteration (solver_stats.cpp:54-66): snapshot the current row
  if (!A.hist) return;
  int len = A.hist_len[b];
  if (len < A.hist_cap) {
    const T vals[kHistFields] = {A.cost_cur[b], A.alpha[b], A.z[b],    A.grad[b],
                                 A.dJ[b],       A.reg_log[b], A.viol[b], A.penmax[b]};
#pragma unroll
    for (int f = 0; f < kHistFields; ++f)
      A.hist[((size_t)f * A.hist_cap + len) * A.Bp + b] = vals[f];
  }
  A.hist_len[b] = len + 1;
}

// -------------------------------------------------------------------------------------------------
// iLQR::UpdateExpansionsBlock (ilqr.hpp:670-677) over grid (instance, knot)
// -------------------------------------------------------------------------------------------------
template <class T, class M>
__global__ __launch_bounds__(kBlock) void k_expansions(DevArrays<T> A, const ProblemDesc* __restrict__ pd,
                                                       int all) {
  constexpr int n = M::n, m = M::m, nm = n + m;
  const int b = blockIdx.x * kBlock + threadIdx.x;
  const int k = blockIdx.y;
  if (b >= A.B) return;
  if (!all && A.phase[b] != 1) return;
  const int N = A.N;
  const size_t Bp = A.Bp;
  T x[n], u[m];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = A.X[((size_t)k * n + i) * Bp + b];
#pragma unroll
  for (int i = 0; i < m; ++i) u[i] = (k < N) ? A.U[((size_t)k * m + i) * Bp + b] : T(0);
  KnotCtx<T> C{A, pd, b};
  T gx[n], gu[m], hxx[n * n], hxu[n * m], huu[m * m];
  T J = knot_cost_expansion<T, n, m>(C, k, x, u, gx, gu, hxx, hxu, huu);
  A.costs[(size_t)k * Bp + b] = J;
#pragma unroll
  for (int e = 0; e < n * n; ++e) A.lxx[((size_t)k * n * n + e) * Bp + b] = hxx[e];
#pragma unroll
  for (int e = 0; e < n; ++e) A.lx[((size_t)k * n + e) * Bp + b] = gx[e];
  if (k < N) {
#pragma unroll
    for (int e = 0; e < n * m; ++e) A.lxu[((size_t)k * n * m + e) * Bp + b] = hxu[e];
#pragma unroll
    for (int e = 0; e < m * m; ++e) A.luu[((size_t)k * m * m + e) * Bp + b] = huu[e];
#pragma unroll
    for (int e = 0; e < m; ++e) A.lu[((size_t)k * m + e) * Bp + b] = gu[e];
    T Jc[n * nm];
    rk4_jacobian<T, M>(x, u, T(A.hstep[k]), Jc);
#pragma unroll
    for (int e = 0; e < n * nm; ++e) A.AB[((size_t)k * n * nm + e) * Bp + b] = Jc[e];
  }
}

// -------------------------------------------------------------------------------------------------
// iLQR::BackwardPass (ilqr.hpp:385-445), one lane per instance
// -------------------------------------------------------------------------------------------------
template <class T, int n, int m>
struct KnotExp {
  T AB[n * (n + m)], lxx[n * n], lxu[n * m], luu[m * m], lx[n], lu[m];
};
template <class T, int n, int m>
ALTRO_DEV void load_knot_exp(const DevArrays<T>& A, int k, int b, KnotExp<T, n, m>& E) {
  const size_t Bp = A.Bp;
#pragma unroll
  for (int e = 0; e < n * (n + m); ++e) E.AB[e] = A.AB[((size_t)k * n * (n + m) + e) * Bp + b];
#pragma unroll
  for (int e = 0; e < n * n; ++e) E.lxx[e] = A.lxx[((size_t)k * n * n + e) * Bp + b];
#pragma unroll
  for (int e = 0; e < n * m; ++e) E.lxu[e] = A.lxu[((size_t)k * n * m + e) * Bp + b];
#pragma unroll
  for (int e = 0; e < m * m; ++e) E.luu[e] = A.luu[((size_t)k * m * m + e) * Bp + b];
#pragma unroll
  for (int e = 0; e < n; ++e) E.lx[e] = A.lx[((size_t)k * n + e) * Bp + b];
#pragma unroll
  for (int e = 0; e < m; ++e) E.lu[e] = A.lu[((size_t)k * m + e) * Bp + b];
}

template <class T, class M>
__global__ __launch_bounds__(kBlock) void k_backward(DevArrays<T> A, DevOpts o, int all) {
  constexpr int n = M::n, m = M::m;
  constexpr bool kPrefetch = (n * (n + m) + n * n + n * m + m * m + n + m) <= 64;
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  if (!all && A.phase[b] != 1) return;
  const int N = A.N;
  const size_t Bp = A.Bp;
  // J0 = costs_.sum() of the expansion step (ilqr.hpp:516); it is also the inner solve's
  // initial_cost on its first iteration (ilqr.hpp:298: same trajectory, same duals/penalties).
  T J0 = T(0);
  for (int k = 0; k <= N; ++k) J0 += A.costs[(size_t)k * Bp + b];
  A.J0[b] = J0;
  if (A.need_init_cost[b]) {
    A.initial_cost[b] = J0;
    A.need_init_cost[b] = 0;
  }
  // CalcTerminalCostToGo (knot_point_function_type.hpp:135-138)
  T P[n * n], p[n];
  auto load_terminal = [&]() {
#pragma unroll
    for (int e = 0; e < n * n; ++e) P[e] = A.lxx[((size_t)N * n * n + e) * Bp + b];
#pragma unroll
    for (int e = 0; e < n; ++e) p[e] = A.lx[((size_t)N * n + e) * Bp + b];
  };
  load_terminal();
  if (A.record_ctg) {
#pragma unroll
    for (int e = 0; e < n * n; ++e) A.P[((size_t)N * n * n + e) * Bp + b] = P[e];
#pragma unroll
    for (int e = 0; e < n; ++e) A.p[((size_t)N * n + e) * Bp + b] = p[e];
  }
  T rho = A.rho_reg[b], drho = A.drho[b];
  T dV0 = T(0), dV1 = T(0);  // zeroed once, NOT per retry (quirk Q4)
  int max_reg_count = 0;
  int status = A.status[b];
  int k = N - 1;
  bool done = (N <= 0);
  KnotExp<T, n, m> E, En;
  if (!done) load_knot_exp<T, n, m>(A, k, b, E);
  while (!done) {
    const int kn = k > 0 ? k - 1 : 0;
    if (kPrefetch) load_knot_exp<T, n, m>(A, kn, b, En);
    T K[m * n], d[m];
    const bool ok = riccati_knot<T, n, m>(E.AB, E.lxx, E.lxu, E.luu, E.lx, E.lu, rho, P, p, K, d, &dV0, &dV1);
    if (!ok) {
      // ilqr.hpp:409-427: raise the regularisation, reset the cost-to-go, restart the sweep
      increase_reg(o, &rho, &drho);
      load_terminal();
      if (rho >= T(o.bp_reg_max)) max_reg_count++;
      if (max_reg_count >= o.bp_reg_fail_threshold) {
        status = ALTRO_BACKWARD_PASS_REGULARIZATION_FAILED;
        done = true;
      } else {
        k = N - 1;
        load_knot_exp<T, n, m>(A, k, b, E);
      }
    } else {
#pragma unroll
      for (int e = 0; e < m * n; ++e) A.K[((size_t)k * m * n + e) * Bp + b] = K[e];
#pragma unroll
      for (int e = 0; e < m; ++e) A.d[((size_t)k * m + e) * Bp + b] = d[e];
      if (A.record_ctg) {
#pragma unroll
        for (int e = 0; e < n * n; ++e) A.P[((size_t)k * n * n + e) * Bp + b] = P[e];
#pragma unroll
        for (int e = 0; e < n; ++e) A.p[((size_t)k * n + e) * Bp + b] = p[e];
      }
      if (k == 0) {
        done = true;
      } else {
        k = kn;
        if (kPrefetch)
          E = En;
        else
          load_knot_exp<T, n, m>(A, k, b, E);
      }
    }
  }
  A.reg_log[b] = rho;  // stats_.Log("reg", rho_)
  decrease_reg(o, &rho, &drho);
  A.rho_reg[b] = rho;
  A.drho[b] = drho;
  A.dV0[b] = dV0;
  A.dV1[b] = dV1;
  A.status[b] = status;
}

// -------------------------------------------------------------------------------------------------
// iLQR::Rollout (ilqr.hpp:453-459), one lane per instance
// -------------------------------------------------------------------------------------------------
template <class T, class M>
__global__ __launch_bounds__(kBlock) void k_rollout(DevArrays<T> A, int all) {
  constexpr int n = M::n, m = M::m;
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  if (!all && A.phase[b] != 1) return;
  const size_t Bp = A.Bp;
  T x[n], u[m], xn[n];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = A.x0[(size_t)i * Bp + b];
  for (int k = 0; k < A.N; ++k) {
#pragma unroll
    for (int i = 0; i < n; ++i) A.X[((size_t)k * n + i) * Bp + b] = x[i];
#pragma unroll
    for (int i = 0; i < m; ++i) u[i] = A.U[((size_t)k * m + i) * Bp + b];
    rk4_step<T, M>(x, u, T(A.hstep[k]), xn);
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = xn[i];
  }
#pragma unroll
  for (int i = 0; i < n; ++i) A.X[((size_t)A.N * n + i) * Bp + b] = x[i];
}

// -------------------------------------------------------------------------------------------------
// iLQR::Cost (ilqr.hpp:326-334, 758-763): per-knot costs over grid (instance, knot) + c_ stores,
// then a per-instance ordered sum.
// -------------------------------------------------------------------------------------------------
template <class T, class M>
__global__ __launch_bounds__(kBlock) void k_knot_costs(DevArrays<T> A, const ProblemDesc* __restrict__ pd) {
  constexpr int n = M::n, m = M::m;
  const int b = blockIdx.x * kBlock + threadIdx.x;
  const int k = blockIdx.y;
  if (b >= A.B) return;
  const size_t Bp = A.Bp;
  T x[n], u[m];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = A.X[((size_t)k * n + i) * Bp + b];
#pragma unroll
  for (int i = 0; i < m; ++i) u[i] = (k < A.N) ? A.U[((size_t)k * m + i) * Bp + b] : T(0);
  KnotCtx<T> C{A, pd, b};
  T v;
  A.costs[(size_t)k * Bp + b] = knot_cost<T, n, m, true>(C, k, x, u, &v);
}
template <class T>
__global__ __launch_bounds__(kBlock) void k_sum_costs(DevArrays<T> A, T* out) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  T J = T(0);
  for (int k = 0; k <= A.N; ++k) J += A.costs[(size_t)k * A.Bp + b];
  out[b] = J;
}

// -------------------------------------------------------------------------------------------------
// Row sweeps shared by the AL transitions
// -------------------------------------------------------------------------------------------------
// max over rows of the stored violation and of the penalty (al_solver.hpp:417-434)
template <class T>
ALTRO_DEV void rows_viol_pen(const DevArrays<T>& A, const ProblemDesc* pd, int b, T* viol, T* pen) {
  T vmax = T(0), pmax = T(0);
  for (int k = 0; k <= A.N; ++k) {
    const KnotClass& kc = pd->cls[A.knot_class[k]];
    const int rb = A.knot_rowbase[k];
    for (int ci = 0; ci < kc.ncon; ++ci) {
      const ConDesc& cd = kc.con[ci];
      for (int i = 0; i < cd.p; ++i) {
        const size_t idx = (size_t)(rb + cd.row_off + i) * A.Bp + b;
        vmax = max_(vmax, violation(cd.type, A.cval[idx]));
        pmax = max_(pmax, A.pen[idx]);
      }
    }
  }
  *viol = vmax;
  *pen = pmax;
}
// ConstraintValues::UpdateDuals (constraint_values.hpp:192-194): per-row penalty, STORED c_ (Q6);
// returns max violation / max penalty of the same rows (al_solver.hpp:357-366).
template <class T>
ALTRO_DEV void rows_update_duals(const DevArrays<T>& A, const ProblemDesc* pd, int b, T* viol, T* pen) {
  T vmax = T(0), pmax = T(0);
  for (int k = 0; k <= A.N; ++k) {
    const KnotClass& kc = pd->cls[A.knot_class[k]];
    const int rb = A.knot_rowbase[k];
    for (int ci = 0; ci < kc.ncon; ++ci) {
      const ConDesc& cd = kc.con[ci];
      for (int i = 0; i < cd.p; ++i) {
        const size_t idx = (size_t)(rb + cd.row_off + i) * A.Bp + b;
        const T c = A.cval[idx], rho = A.pen[idx];
        A.lam[idx] = dual_proj(cd.type, A.lam[idx] - rho * c);
        vmax = max_(vmax, violation(cd.type, c));
        pmax = max_(pmax, rho);
      }
    }
  }
  *viol = vmax;
  *pen = pmax;
}
// ConstraintValues::UpdatePenalties (constraint_values.hpp:202-207)
template <class T>
ALTRO_DEV void rows_update_penalties(const DevArrays<T>& A, const ProblemDesc* pd, int b) {
  for (int k = 0; k <= A.N; ++k) {
    const int cls = A.knot_class[k];
    const KnotClass& kc = pd->cls[cls];
    const int rb = A.knot_rowbase[k];
    for (int ci = 0; ci < kc.ncon; ++ci) {
      const ConDesc& cd = kc.con[ci];
      const T phi = T(A.phi[cls * kMaxConPerKnot + ci]);
      for (int i = 0; i < cd.p; ++i) A.pen[(size_t)(rb + cd.row_off + i) * A.Bp + b] *= phi;
    }
  }
}
template <class T>
ALTRO_DEV void rows_set(const DevArrays<T>& A, const ProblemDesc* pd, int b, bool zero_lam, bool set_pen, T rho) {
  for (int r = 0; r < pd->total_rows; ++r) {
    if (zero_lam) A.lam[(size_t)r * A.Bp + b] = T(0);
    if (set_pen) A.pen[(size_t)r * A.Bp + b] = rho;
  }
}

// iLQR::SolveSetup / ResetInternalVariables (ilqr.hpp:629-645, 680-690)
template <class T>
ALTRO_DEV void begin_inner_solve(const DevArrays<T>& A, const DevOpts& o, int b) {
  A.it_inner[b] = 0;
  A.status[b] = ALTRO_UNSOLVED;
  A.rho_reg[b] = T(o.bp_reg_initial);
  A.drho[b] = T(0);
  A.dV0[b] = T(0);
  A.dV1[b] = T(0);
  A.need_init_cost[b] = 1;  // stats_.initial_cost = Cost() is taken from the next expansion step
}

// AugmentedLagrangianiLQR::Init (al_solver.hpp:287-302) without the (unobservable) initial
// MaxViolation log, + activation of every instance.
template <class T>
__global__ __launch_bounds__(kBlock) void k_al_init(DevArrays<T> A, const ProblemDesc* __restrict__ pd, DevOpts o) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  rows_set(A, pd, b, o.reset_duals != 0, o.initial_penalty > 0, T(o.initial_penalty));  // quirk Q8
  // stats.Reset()
  A.initial_cost[b] = T(0);
  A.it_inner[b] = A.it_outer[b] = A.it_total[b] = 0;
  A.cost_cur[b] = A.cost_prev[b] = A.dJ[b] = A.grad[b] = A.viol[b] = A.penmax[b] = T(0);
  A.alpha[b] = A.z[b] = A.reg_log[b] = T(0);
  if (A.hist) A.hist_len[b] = 0;
  A.status_al[b] = ALTRO_UNSOLVED;
}
// finishing touch of AL Init for the step-level API: log viol (after a cost evaluation) and pen
template <class T>
__global__ __launch_bounds__(kBlock) void k_log_viol_pen(DevArrays<T> A, const ProblemDesc* __restrict__ pd) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  T v, p;
  rows_viol_pen(A, pd, b, &v, &p);
  A.viol[b] = v;
  A.penmax[b] = p;
}
template <class T>
__global__ __launch_bounds__(kBlock) void k_solve_setup(DevArrays<T> A, DevOpts o, int activate) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  begin_inner_solve(A, o, b);
  if (activate) A.phase[b] = 1;
}
template <class T>
__global__ __launch_bounds__(kBlock) void k_set_rows(DevArrays<T> A, const ProblemDesc* __restrict__ pd,
                                                     int zero_lam, int set_pen, T rho) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  rows_set(A, pd, b, zero_lam != 0, set_pen != 0, rho);
}
template <class T>
__global__ __launch_bounds__(kBlock) void k_update_duals(DevArrays<T> A, const ProblemDesc* __restrict__ pd) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  T v, p;
  rows_update_duals(A, pd, b, &v, &p);
}
template <class T>
__global__ __launch_bounds__(kBlock) void k_update_penalties(DevArrays<T> A, const ProblemDesc* __restrict__ pd) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  rows_update_penalties(A, pd, b);
}
template <class T>
__global__ __launch_bounds__(kBlock) void k_max_viol_pen(DevArrays<T> A, const ProblemDesc* __restrict__ pd,
                                                         T* viol, T* pen) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  T v, p;
  rows_viol_pen(A, pd, b, &v, &p);
  if (viol) viol[b] = v;
  if (pen) pen[b] = p;
}

// iLQR::UpdateConvergenceStatistics + IsDone (ilqr.hpp:568-619) for one instance.
// gsum = sum_k max_i |d_k,i| / (|u_k,i| + 1) with the POST-forward-pass controls (quirk Q12).
// Returns true when the inner solve is finished.
template <class T>
ALTRO_DEV bool conv_stats_and_done(const DevArrays<T>& A, const DevOpts& o, int b, T gsum, T viol) {
  const T grad = A.N > 0 ? gsum / T(A.N) : T(0);
  const int it = A.it_inner[b];
  const T dJ = (it == 0) ? A.initial_cost[b] - A.cost_cur[b] : A.cost_prev[b] - A.cost_cur[b];
  A.it_inner[b] = it + 1;
  const int itot = A.it_total[b] + 1;
  A.it_total[b] = itot;
  A.dJ[b] = dJ;
  A.viol[b] = viol;
  A.grad[b] = grad;
  hist_push(A, b);
  A.cost_prev[b] = A.cost_cur[b];  // NewIteration copies the row (solver_stats.cpp:54-66)
  int status = A.status[b];
  bool done = false;
  if (dJ < T(o.cost_tolerance) && grad < T(o.gradient_tolerance)) {
    status = ALTRO_SOLVED;
    done = true;
  } else if (it + 1 >= o.max_iterations_inner) {
    status = ALTRO_MAX_INNER_ITERATIONS;
    done = true;
  } else if (itot >= o.max_iterations_total) {
    status = ALTRO_MAX_ITERATIONS;
    done = true;
  } else if (status != ALTRO_UNSOLVED) {
    done = true;
  }
  A.status[b] = status;
  return done;
}

// AL outer-loop step after an inner solve finished: UpdateDuals, UpdateConvergenceStatistics,
// IsDone, UpdatePenalties (al_solver.hpp:313-401).  Returns true if the instance keeps iterating.
template <class T>
ALTRO_DEV bool al_outer_step(const DevArrays<T>& A, const ProblemDesc* pd, const DevOpts& o, int b) {
  T viol, pen;
  rows_update_duals(A, pd, b, &viol, &pen);
  const int outer = A.it_outer[b] + 1;
  A.it_outer[b] = outer;
  A.viol[b] = viol;
  A.penmax[b] = pen;
  const int st = A.status[b];
  int sal = -1;
  if (st != ALTRO_SOLVED)
    sal = st;
  else if (viol < T(o.constraint_tolerance))
    sal = ALTRO_SOLVED;
  else if (pen > T(o.maximum_penalty))
    sal = ALTRO_MAX_PENALTY;
  else if (outer >= o.max_iterations_outer)
    sal = ALTRO_MAX_OUTER_ITERATIONS;
  else if (A.it_total[b] >= o.max_iterations_total)
    sal = ALTRO_MAX_ITERATIONS;
  if (sal >= 0) {
    A.status_al[b] = sal;
    return false;
  }
  rows_update_penalties(A, pd, b);
  begin_inner_solve(A, o, b);
  return true;
}

// step-level iLQR::UpdateConvergenceStatistics
template <class T, class M>
__global__ __launch_bounds__(kBlock) void k_conv_stats(DevArrays<T> A, const ProblemDesc* __restrict__ pd, DevOpts o) {
  constexpr int m = M::m;
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  T gsum = T(0);
  for (int k = 0; k < A.N; ++k) {
    T mx = T(0);
#pragma unroll
    for (int i = 0; i < m; ++i) {
      const size_t idx = ((size_t)k * m + i) * A.Bp + b;
      mx = max_(mx, abs_(A.d[idx]) / (abs_(A.U[idx]) + T(1)));
    }
    gsum += mx;
  }
  T v, p;
  rows_viol_pen(A, pd, b, &v, &p);
  const int st = A.status[b];
  conv_stats_and_done(A, o, b, gsum, v);
  A.status[b] = st;  // the status change belongs to IsDone, which the step-level API does not call
}

// -------------------------------------------------------------------------------------------------
// iLQR::ForwardPass (ilqr.hpp:512-558) with speculative parallel line search, + state machine
// -------------------------------------------------------------------------------------------------
template <class T, class M>
__global__ __launch_bounds__(kBlock) void k_forward(DevArrays<T> A, const ProblemDesc* __restrict__ pd,
                                                    DevOpts o, int mode, int all, int* active_counter) {
  constexpr int n = M::n, m = M::m;
  constexpr int LS = kLineSearchLanes;
  constexpr int kPerWave = kBlock / LS;  // instances per wavefront (3)
  const int lane = threadIdx.x;
  const int grp = lane / LS;
  const int t = lane - grp * LS;
  const int b = blockIdx.x * kPerWave + grp;
  const size_t Bp = A.Bp;
  const int N = A.N;
  const bool valid = (grp < kPerWave) && (b < A.B) && (all || A.phase[b] == 1);
  const int bb = valid ? b : 0;  // idle lanes shadow instance 0's loads but never store
  KnotCtx<T> C{A, pd, bb};

  const T J0 = A.J0[bb];
  const T dV0 = A.dV0[bb], dV1 = A.dV1[bb];
  T x0[n];
#pragma unroll
  for (int i = 0; i < n; ++i) x0[i] = A.x0[(size_t)i * Bp + bb];

  const int ls_max = o.line_search_max_iterations;
  bool accepted = false;
  T alpha_sel = T(0), J_sel = J0, z_sel = T(-1), g_sel = T(0), g_old = T(0);
  T alpha_replay = T(0);
  bool have_replay = false;
  int last_status = ALTRO_UNSOLVED;

  T alpha_base = T(1);
  for (int base = 0; base < ls_max && !accepted; base += LS) {
    // this lane's step length: alpha /= decrease_factor, t times (ilqr.hpp:544)
    T alpha = alpha_base;
    for (int i = 0; i < t; ++i) alpha /= T(o.line_search_decrease_factor);
    const bool live = valid && (base + t < ls_max);
    // ---- pass 1: closed-loop rollout + cost for this lane's alpha (ilqr.hpp:468-499, 527) -------
    bool ok = true;
    int st = ALTRO_UNSOLVED;
    T J = T(0), gs = T(0), go = T(0);
    T xb[n];
#pragma unroll
    for (int i = 0; i < n; ++i) xb[i] = x0[i];
    for (int k = 0; k < N; ++k) {
      T xk[n], uk[m], K[m * n], d[m], ub[m], xn[n];
#pragma unroll
      for (int i = 0; i < n; ++i) xk[i] = A.X[((size_t)k * n + i) * Bp + bb];
#pragma unroll
      for (int i = 0; i < m; ++i) uk[i] = A.U[((size_t)k * m + i) * Bp + bb];
#pragma unroll
      for (int e = 0; e < m * n; ++e) K[e] = A.K[((size_t)k * m * n + e) * Bp + bb];
#pragma unroll
      for (int i = 0; i < m; ++i) d[i] = A.d[((size_t)k * m + i) * Bp + bb];
      if (ok) {
        T gm = T(0), gmo = T(0);
#pragma unroll
        for (int i = 0; i < m; ++i) {
          T s = T(0);
#pragma unroll
          for (int l = 0; l < n; ++l) s += K[i + l * m] * (xb[l] - xk[l]);
          ub[i] = uk[i] + s + d[i] * alpha;
          gm = max_(gm, abs_(d[i]) / (abs_(ub[i]) + T(1)));
          gmo = max_(gmo, abs_(d[i]) / (abs_(uk[i]) + T(1)));
        }
        gs += gm;
        go += gmo;
        J += knot_cost<T, n, m, false>(C, k, xb, ub, nullptr);
        rk4_step<T, M>(xb, ub, T(A.hstep[k]), xn);
        if (o.check_forwardpass_bounds) {
          T sx = T(0), su = T(0);
#pragma unroll
          for (int i = 0; i < n; ++i) sx += xn[i] * xn[i];
#pragma unroll
          for (int i = 0; i < m; ++i) su += ub[i] * ub[i];
          if (sqrt_(sx) > T(o.state_max)) {
            ok = false;
            st = ALTRO_STATE_LIMIT;
          } else if (sqrt_(su) > T(o.control_max)) {
            ok = false;
            st = ALTRO_CONTROL_LIMIT;
          }
        }
#pragma unroll
        for (int i = 0; i < n; ++i) xb[i] = xn[i];
      }
    }
    if (ok) {
      T uz[m];
#pragma unroll
      for (int i = 0; i < m; ++i) uz[i] = T(0);
      J += knot_cost<T, n, m, false>(C, N, xb, uz, nullptr);
    }
    // ---- acceptance test (ilqr.hpp:528-542) --------------------------------------------------
    const T expected = -alpha * (dV0 + alpha * dV1);
    const T z = (expected > T(0)) ? (J0 - J) / expected : T(-1);
    const bool acc = live && ok && T(o.line_search_lower_bound) <= z &&
                     z <= T(o.line_search_upper_bound) && J < J0;
    // ---- pick the first accepted trial of this instance, exactly as the serial loop would ------
    const unsigned long long accm = __ballot(acc);
    const unsigned long long okm = __ballot(live && ok);
    const unsigned gmask = (1u << LS) - 1u;
    const unsigned acc_g = (unsigned)(accm >> (grp * LS)) & gmask;
    const unsigned ok_g = (unsigned)(okm >> (grp * LS)) & gmask;
    int nlive = ls_max - base;
    if (nlive > LS) nlive = LS;
    if (acc_g) {
      const int tsel = __ffs(acc_g) - 1;
      const int src = grp * LS + tsel;
      alpha_sel = __shfl(alpha, src);
      J_sel = __shfl(J, src);
      z_sel = __shfl(z, src);
      g_sel = __shfl(gs, src);
      accepted = true;
      last_status = ALTRO_UNSOLVED;  // the accepted rollout was the last one run (ilqr.hpp:497)
      alpha_replay = alpha_sel;
      have_replay = true;
    } else {
      // no acceptance in this round: the serial loop ran all `nlive` trials; c_ now holds the
      // constraint values of the last trial whose rollout succeeded (quirk Q6), and status_ is the
      // outcome of the very last rollout.
      const int src_last = grp * LS + (nlive - 1);
      last_status = __shfl(st, src_last);
      if (ok_g) {
        const int tl = 31 - __clz(ok_g);
        alpha_replay = __shfl(alpha, grp * LS + tl);
        have_replay = true;
      }
    }
    g_old = __shfl(go, grp * LS);  // trial 0 always runs the whole horizon unless it blew up
    if (!(ok_g & 1u)) g_old = T(-1);
    alpha_base = __shfl(alpha, grp * LS + (LS - 1)) / T(o.line_search_decrease_factor);
  }

  if (!valid || t != 0) return;

  // ---- pass 2 (one lane per instance): replay the selected step.  Accepted: write the new
  //      trajectory in place ((*Z_) = (*Zbar_), ilqr.hpp:548) and the c_ it leaves behind.
  //      Rejected: only reproduce the stale c_ of the last evaluated candidate (quirk Q6). --------
  T viol = T(0);
  if (have_replay) {
    T xb[n];
#pragma unroll
    for (int i = 0; i < n; ++i) xb[i] = x0[i];
    for (int k = 0; k < N; ++k) {
      T xk[n], uk[m], K[m * n], d[m], ub[m], xn[n];
#pragma unroll
      for (int i = 0; i < n; ++i) xk[i] = A.X[((size_t)k * n + i) * Bp + b];
#pragma unroll
      for (int i = 0; i < m; ++i) uk[i] = A.U[((size_t)k * m + i) * Bp + b];
#pragma unroll
      for (int e = 0; e < m * n; ++e) K[e] = A.K[((size_t)k * m * n + e) * Bp + b];
#pragma unroll
      for (int i = 0; i < m; ++i) d[i] = A.d[((size_t)k * m + i) * Bp + b];
#pragma unroll
      for (int i = 0; i < m; ++i) {
        T s = T(0);
#pragma unroll
        for (int l = 0; l < n; ++l) s += K[i + l * m] * (xb[l] - xk[l]);
        ub[i] = uk[i] + s + d[i] * alpha_replay;
      }
      T v;
      knot_cost<T, n, m, true>(C, k, xb, ub, &v);
      viol = max_(viol, v);
      if (accepted) {
#pragma unroll
        for (int i = 0; i < n; ++i) A.X[((size_t)k * n + i) * Bp + b] = xb[i];
#pragma unroll
        for (int i = 0; i < m; ++i) A.U[((size_t)k * m + i) * Bp + b] = ub[i];
      }
      rk4_step<T, M>(xb, ub, T(A.hstep[k]), xn);
#pragma unroll
      for (int i = 0; i < n; ++i) xb[i] = xn[i];
    }
    T uz[m], v;
#pragma unroll
    for (int i = 0; i < m; ++i) uz[i] = T(0);
    knot_cost<T, n, m, true>(C, N, xb, uz, &v);
    viol = max_(viol, v);
    if (accepted) {
#pragma unroll
      for (int i = 0; i < n; ++i) A.X[((size_t)N * n + i) * Bp + b] = xb[i];
    }
  } else {
    T pmax;
    rows_viol_pen(A, pd, b, &viol, &pmax);  // c_ untouched since the expansion step
  }

  if (accepted) {
    A.cost_cur[b] = J_sel;  // stats_.Log("cost"/"alpha"/"z")
    A.alpha[b] = alpha_sel;
    A.z[b] = z_sel;
  } else {
    T rho = A.rho_reg[b], drho = A.drho[b];
    increase_reg(o, &rho, &drho);  // ilqr.hpp:550
    A.rho_reg[b] = rho;
    A.drho[b] = drho;
  }
  A.status[b] = last_status;
  if (mode == kFwdStepOnly) {
    A.viol[b] = viol;
    return;
  }

  T gsum = accepted ? g_sel : g_old;
  if (!accepted && g_old < T(0)) {  // trial 0 aborted early: recompute with the unchanged controls
    gsum = T(0);
    for (int k = 0; k < N; ++k) {
      T mx = T(0);
#pragma unroll
      for (int i = 0; i < m; ++i) {
        const size_t idx = ((size_t)k * m + i) * Bp + b;
        mx = max_(mx, abs_(A.d[idx]) / (abs_(A.U[idx]) + T(1)));
      }
      gsum += mx;
    }
  }
  bool active = true;
  if (conv_stats_and_done(A, o, b, gsum, viol)) {
    if (mode == kFwdAL) {
      active = al_outer_step(A, pd, o, b);
    } else {
      active = false;
    }
  }
  if (!active) {
    A.phase[b] = 0;
  } else if (active_counter) {
    atomicAdd(active_counter, 1);
  }
}

// gather {cost, violation, iterations_total, status} as 4 fp64 per instance (RCCL payload)
template <class T>
__global__ __launch_bounds__(kBlock) void k_pack_results(DevArrays<T> A, double* dst, int ilqr_mode) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  dst[4 * (size_t)b + 0] = (double)A.cost_cur[b];
  dst[4 * (size_t)b + 1] = (double)A.viol[b];
  dst[4 * (size_t)b + 2] = (double)A.it_total[b];
  dst[4 * (size_t)b + 3] = (double)(ilqr_mode ? A.status[b] : A.status_al[b]);
}

}  // namespace altro_hip
