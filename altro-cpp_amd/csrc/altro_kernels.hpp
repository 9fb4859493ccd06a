// altro_kernels.hpp — HIP kernels of the batched AL-iLQR solver (gfx950 / MI355X).
//
// One batched iLQR "sweep" = three launches, each advancing EVERY still-active instance by one
// inner iteration of altro::ilqr::iLQR<n,m>::Solve (altro/ilqr/ilqr.hpp:300-313):
//
//   k_expansions     grid (instance x knot): cost/AL expansion + RK4 Jacobian + knot cost.
//                    Embarrassingly parallel (ilqr.hpp:670-677).
//   k_backward_mfma  backward Riccati recursion on the fp64 4x4x4 matrix cores, 16 lanes per instance,
//                    4 instances per wavefront (n = 3, m = 2; fp64 arithmetic, fp64 or fp32 storage);
//                    k_backward_coop: one instance per wavefront, matrices in LDS (n >= 6);
//                    k_backward: one lane per instance on the VALU (fallback).  All keep the reference's
//                    restart-on-Cholesky-failure schedule (ilqr.hpp:385-445) with a wave-uniform k.
//   k_forward2       SPECULATIVE PARALLEL LINE SEARCH: the (up to) 20 backtracking trials of
//                    ilqr.hpp:525-545 are independent closed-loop rollouts, so each instance gets 20
//                    lanes that evaluate alpha = 1, 1/2, ... 2^-19 side by side; a wave ballot picks
//                    the first trial the serial loop would have accepted.  Inputs are staged in LDS;
//                    the step is pipelined over a rollout wave, a cost wave and an auxiliary wave
//                    (bound checks, gradient measure, candidate stores); every trial stores its
//                    candidate trajectory so the winner is copied, not re-integrated; the same kernel
//                    runs the per-instance state machine: convergence statistics, IsDone, dual/penalty
//                    update and the AL outer-loop transition (ilqr.hpp:568-619, al_solver.hpp:313-401).
//                    k_forward is the single-wave, HBM-reading fallback.
//   k_sweep_fused    the tail of a batched solve: one workgroup per straggler instance runs whole
//                    iterations (expansions, MFMA backward pass, three-wave forward pass) in a loop
//                    until its instance is finished -- one persistent launch, no host in the loop.
//
// Instances are independent; the ones still iterating are kept in a dense list rebuilt every sweep.
#pragma once

#include <type_traits>

#include "altro_device.hpp"

namespace altro_hip {

constexpr int kBlock = 64;  // one wavefront per workgroup: instances never share data
constexpr int kFwdWaves = 3;  // k_forward2 / k_sweep_fused: rollout wave, cost wave, auxiliary wave
// knots per synchronisation of the persistent kernel's knot loop (the batched sweeps: 2), see producer_syncs_after.
// Round 2 (hardware barriers only): 4 measured 5.24 -> 5.30 ms on config 2, 5.03 -> 4.74 ms on config 3, the headline kept
// 2.  Round 3: with the forward waves of config 2 synchronised through sequence words (kSpecFree) a meeting costs the
// consumers an LDS round trip, and 4 wins on both (tail iteration 43.8 -> 42.2 us on config 2, 57.1 -> 55.6 us on
// config 3): 4.
#ifndef ALTRO_SYNC_FUSED
#define ALTRO_SYNC_FUSED 4
#endif
constexpr int kSyncFused = ALTRO_SYNC_FUSED;
constexpr int kEAheadToErrWord = -7;      // kSyErr - kSyEAhead0 (FwdSyncWord, asserted there)
constexpr int kFwdSpinLimit = 1 << 22;   // polls of an LDS sequence word (~0.1 us each) before a wave gives up
// Debugging aid (ALTRO_HIP_DEBUG_POISON): fills the LDS of the CU it lands on with a pattern, so that a kernel that reads
// LDS it has not written computes with the pattern instead of with whatever the previous kernel happened to leave there.
// ALTRO_HIP_DEBUG_POISON: the shadow columns [col0, col0 + ncols) of one per-instance array ([rows][Bp][words] 32-bit words)
// filled with the poison pattern before a solve -- a clone that forgot to copy something computes with NaN words instead
// of with what an earlier solve left in the column.
template <int kDummy>
__global__ __launch_bounds__(256) void k_poison_columns(unsigned* arr, unsigned rows, unsigned Bp, unsigned col0, unsigned ncols,
                                                         unsigned words, unsigned pattern, int mix) {
  const size_t per_row = (size_t)ncols * words, total = (size_t)rows * per_row;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / per_row, w = i - r * per_row;
    arr[(r * Bp + col0) * words + w] = mix ? (pattern ^ ((unsigned)i * 2654435761u)) : pattern;
  }
}
template <int kDummy>
__global__ __launch_bounds__(256) void k_poison_lds(unsigned pattern, int words, int mix, int* sink) {
  extern __shared__ unsigned poison_smem[];
  for (int i = threadIdx.x; i < words; i += 256) poison_smem[i] = mix ? (pattern ^ ((unsigned)i * 2654435761u)) : pattern;
  __syncthreads();
  if (sink && poison_smem[(threadIdx.x * 97 + blockIdx.x) % words] == 0x13572468u && mix == 7) sink[0] = 1;
}
// workgroup barrier that only waits for this wave's LDS traffic (not for its global loads / stores)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Debug build only (-DALTRO_STAMPS, scripts/gpu_stamps.sh): phase stamps of the persistent kernel, per wave, summed over the
// iterations of workgroup 0 in LDS and printed by the kernel at its end.  s_memtime runs at the shader clock.
#ifdef ALTRO_STAMPS
__shared__ long long g_stamp_acc[32];
__device__ __forceinline__ void stamp_add(long long* acc, int slot, long long t0) {
  if ((threadIdx.x & 63) == 0) acc[slot] += (long long)__builtin_amdgcn_s_memtime() - t0;
}
#define ALTRO_STAMP_T0() ((long long)__builtin_amdgcn_s_memtime())
#define ALTRO_STAMP_ADD(slot, t0) stamp_add(g_stamp_acc, (slot), (t0))
#else
#define ALTRO_STAMP_T0() 0ll
#define ALTRO_STAMP_ADD(slot, t0) (void)(t0)
#endif

// constraint rows and per-instance scalars: arr[row*Bp + b]
#define SOA(arr, row) (arr)[(unsigned)(row) * (unsigned)Bp + (unsigned)b]
// record base of knot k for this lane's instance: arr + (k*Bp + b)*EP
#define RECP(arr, k, EP) ((arr) + ((size_t)(unsigned)(k) * (unsigned)Bp + (unsigned)b) * (unsigned)(EP))

// Instance handled by slot `idx` of this launch (-1: none).  `all` = 1: launches that cover every instance; 2: every
// instance that is still iterating (phase == 1), addressed by its own index (k_expansions' dense mode).
template <class T>
ALTRO_DEV int instance_of_slot(const DevArrays<T>& A, int idx, int all) {
  if (all == 2) {
    const int hi = A.chain_hi ? A.chain_hi : A.B;
    int b = idx + A.chain_lo;
    if (b >= hi) {  // behind the chain's own instances: its slice of shadow columns (segments of rejection streaks)
      if (!A.seg_end) return -1;
      b = A.seg_lo + (b - hi);
      if (b >= A.seg_hi) return -1;
    }
    return A.phase[b] == 1 ? b : -1;
  }
  if (all) return idx < A.B ? idx : -1;
  const int cnt = A.act_count ? *A.act_count : A.act_count_const;
  if (idx >= cnt) return -1;
  return A.act_list ? A.act_list[idx] : idx;
}

// XCD-AWARE WORKGROUP ORDER (round 4).  The hardware hands workgroup i of a launch to XCD i % 8, each XCD with an L2 of
// its own.  Neighbouring instances share cache lines -- the constraint rows are [row][b] with 8-byte elements (16
// instances per 128-byte line), X / U / gain records hold 4 / 8 / 2 instances per line -- so with the natural order the
// five or six workgroups that cover one line sit on as many XCDs and every one of them pulls the line through its own L2
// (and writes its own partial copy back).  The remap gives XCD x the x-th contiguous eighth of the slots: neighbours meet
// in one L2.  A bijection on the first 8 * (nblocks / 8) blocks; the ragged tail keeps its index.
ALTRO_DEV int xcd_block(int bid, int nblocks, int on) {
  constexpr int kXcd = 8;
  const int per = nblocks / kXcd;
  if (!on || bid >= per * kXcd) return bid;
  return (bid % kXcd) * per + bid / kXcd;
}

template <class T>
ALTRO_DEV const KnotClass& class_of_knot(const DevArrays<T>& A, const ProblemDesc* pd, int k, int* rowbase) {
  *rowbase = A.knot_rowbase[k];
  return pd->cls[A.knot_class[k]];
}

enum ForwardMode { kFwdStepOnly = 0, kFwdILQR = 1, kFwdAL = 2 };

template <class T>
ALTRO_DEV void hist_push(const DevArrays<T>& A, int b) {
  // SolverStats::NewIteration (solver_stats.cpp:54-66): snapshot the current row
  if (!A.hist) return;
  int len = A.hist_len[b];
  if (len < A.hist_cap) {
    const double vals[kHistFields] = {A.cost_cur[b], A.alpha[b], A.z[b],    A.grad[b],
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
// One (instance, knot) of iLQR::UpdateExpansions: returns the knot cost, writes the record.
template <class T, class M>
ALTRO_DEV T expansion_body(const DevArrays<T>& A, const ProblemDesc* __restrict__ pd, int b, int k) {
  constexpr int n = M::n, m = M::m;
  using R = Rec<T, n, m>;
  using RS = rec_scalar_t<T, M>;
  using RR = Rec<RS, n, m>;
  const int N = A.N;
  const unsigned Bp = A.Bp;
  T xr[R::nP], ur[R::mP];
  load_rec<T, R::nP>(RECP(A.X, k, R::nP), xr);
#pragma unroll
  for (int i = 0; i < R::mP; ++i) ur[i] = T(0);
  if (k < N) load_rec<T, R::mP>(RECP(A.U, k, R::mP), ur);
  CtxG<T> C(A, b);
  T E[R::EP];
#pragma unroll
  for (int e = 0; e < R::EP; ++e) E[e] = T(0);
  int rb;
  const KnotClass& kc = class_of_knot(A, pd, k, &rb);
  const T J = knot_cost_expansion<T, n, m>(C, pd, kc, rb, xr, ur, E + R::oLx, E + R::oLu, E + R::oLxx, E + R::oLxu,
                                           E + R::oLuu);
  A.costs[(unsigned)k * Bp + (unsigned)b] = J;
  if (k < N) discrete_jacobian<T, M>(xr, ur, step_of(A, pd, k), E + R::oAB, time_of(A, k), model_of(A, k));
  store_rec_as<T, RS, R::EP, RR::EP, R::eE>(RECP((RS*)A.EXP, k, RR::EP), E);
  return J;
}
// tell the host how many instances this sweep works on (it is polling the mapped word)
template <class T>
ALTRO_DEV void publish_count(const DevArrays<T>& A) {
  if (A.host_count)
    __hip_atomic_store(A.host_count, A.act_count ? *A.act_count : A.act_count_const, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
// all = 2 (dense mode, while a good part of the batch is still iterating): lane = instance index instead of a slot of
// the active list -- the list is appended to by atomics in arbitrary order, so a wavefront's instances are scattered
// over the batch and every 8-byte row / 32-byte record access of a lane pulls its own cache line (this kernel is
// HBM-bound).  The blocks of knot 0 also REBUILD the list for the backward and forward kernels of the sweep: runs of
// up to 64 instances in index order (the runs themselves land in arbitrary order), so that the instances of a
// workgroup are neighbours there too.
template <class T, class M>
__global__ __launch_bounds__(kBlock) void k_expansions(DevArrays<T> A, const ProblemDesc* __restrict__ pd,
                                                       int all, int* order_list = nullptr, int* order_count = nullptr) {
  const int b = instance_of_slot(A, blockIdx.x * kBlock + threadIdx.x, all);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) publish_count(A);
  if (all == 2 && blockIdx.y == 0) {
    const unsigned long long mask = __ballot(b >= 0);
    if (mask != 0ull) {
      int base = 0;
      if (threadIdx.x == 0) base = atomicAdd(order_count, (int)__popcll(mask));
      base = __shfl(base, 0);
      if (b >= 0) order_list[base + (int)__popcll(mask & ((1ull << threadIdx.x) - 1ull))] = b;
    }
  }
  if (b < 0) return;
  expansion_body<T, M>(A, pd, b, blockIdx.y);
}

// -------------------------------------------------------------------------------------------------
// iLQR::BackwardPass (ilqr.hpp:385-445), one lane per instance, k uniform across the wavefront.
//
// The reference restarts the whole sweep when a Cholesky factorisation fails (after raising the
// regularisation).  Here a lane whose factorisation failed simply sits out the rest of the current
// sweep and the wave runs another sweep for the lanes that need one: per instance the sequence of
// operations (and of dV accumulations, quirk Q4) is exactly the reference's, and because k is a
// scalar every record address is `scalar base + lane offset + immediate`.
// -------------------------------------------------------------------------------------------------
template <class T, class M>
__global__ __launch_bounds__(kBlock) void k_backward(DevArrays<T> A, DevOpts o, int all) {
  constexpr int n = M::n, m = M::m;
  using R = Rec<T, n, m>;
  using RS = rec_scalar_t<T, M>;
  using RR = Rec<RS, n, m>;
  const RS* const EXPp = (const RS*)A.EXP;
  const int b0 = instance_of_slot(A, blockIdx.x * kBlock + threadIdx.x, all);
  const bool lane_on = b0 >= 0;
  if (__ballot(lane_on) == 0ull) return;
  const int b = lane_on ? b0 : 0;
  const int N = A.N;
  const unsigned Bp = A.Bp;
  auto load_exp = [&](int k, T* E) __attribute__((always_inline)) {
    load_rec_as<T, RS, R::EP, RR::EP, R::eE>(RECP(EXPp, k, RR::EP), E);
  };
  // J0 = costs_.sum() of the expansion step (ilqr.hpp:516); it is also the inner solve's
  // initial_cost on its first iteration (ilqr.hpp:298: same trajectory, same duals/penalties).
  double J0 = 0.0;
#pragma unroll 8
  for (int k = 0; k <= N; ++k) J0 += (double)A.costs[(unsigned)k * Bp + (unsigned)b];
  double rho = A.rho_reg[b], drho = A.drho[b];
  double dV0 = 0.0, dV1 = 0.0;  // zeroed once, NOT per retry (quirk Q4)
  int max_reg_count = 0;
  int status = A.status[b];
  bool need = lane_on && N > 0;  // this lane still has to complete a sweep
  T P[n * n], p[n];
  T E[R::EP];
  while (__ballot(need) != 0ull) {
    // CalcTerminalCostToGo (knot_point_function_type.hpp:135-138)
    load_exp(N, E);
#pragma unroll
    for (int e = 0; e < n * n; ++e) P[e] = E[R::oLxx + e];
#pragma unroll
    for (int e = 0; e < n; ++e) p[e] = E[R::oLx + e];
    if (A.record_ctg && need) {
      T c[R::CP];
#pragma unroll
      for (int e = 0; e < R::CP; ++e) c[e] = T(0);
#pragma unroll
      for (int e = 0; e < n * n; ++e) c[R::oP + e] = P[e];
#pragma unroll
      for (int e = 0; e < n; ++e) c[R::op + e] = p[e];
      store_rec<T, R::CP>(RECP(A.CTG, N, R::CP), c);
    }
    bool running = need;
    load_exp(N - 1, E);
    for (int k = N - 1; k >= 0; --k) {
      // Q-function assembly consumes the expansion registers ...
      QExp<T, n, m> Q;
      riccati_q<T, n, m>(E + R::oAB, E + R::oLxx, E + R::oLxu, E + R::oLuu, E + R::oLx, E + R::oLu, P, p, Q);
      // ... which are immediately refilled with the next knot's record: the loads fly while the
      // Cholesky / gains / cost-to-go half of this knot executes (no second register buffer)
      if (k > 0) load_exp(k - 1, E);
      if (running) {
        T KD[R::KP];
#pragma unroll
        for (int e = 0; e < R::KP; ++e) KD[e] = T(0);
        const bool ok = riccati_gains<T, n, m>(Q, T(rho), P, p, KD + R::oK, KD + R::oD, &dV0, &dV1);
        if (!ok) {
          // ilqr.hpp:409-427: raise the regularisation and restart the sweep (next round)
          increase_reg(o, &rho, &drho);
          if (rho >= o.bp_reg_max) max_reg_count++;
          if (max_reg_count >= o.bp_reg_fail_threshold) {
            status = ALTRO_BACKWARD_PASS_REGULARIZATION_FAILED;
            need = false;
          }
          running = false;
        } else {
          store_rec_as<T, RS, R::KP, RR::KP, m * n + m>(RECP((RS*)A.KD, k, RR::KP), KD);
          if (A.record_ctg) {
            T c[R::CP];
#pragma unroll
            for (int e = 0; e < R::CP; ++e) c[e] = T(0);
#pragma unroll
            for (int e = 0; e < n * n; ++e) c[R::oP + e] = P[e];
#pragma unroll
            for (int e = 0; e < n; ++e) c[R::op + e] = p[e];
            store_rec<T, R::CP>(RECP(A.CTG, k, R::CP), c);
          }
          if (k == 0) need = false;  // sweep completed
        }
      }
    }
  }
  if (!lane_on) return;
  A.J0[b] = J0;
  if (A.need_init_cost[b]) {
    A.initial_cost[b] = J0;
    A.need_init_cost[b] = 0;
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
// iLQR::BackwardPass with ONE INSTANCE PER WAVEFRONT, for the larger models (n = 6, n = 12): the
// one-lane-per-instance kernel above keeps P, the record and the Q-function of an instance in one lane's
// registers, which for n = 12 is > 1000 values -- it spills to scratch and leaves 1024 instances on 16
// wavefronts.  Here the 64 lanes of a wave share the matrices of one instance in LDS (~9 KB) and each
// lane computes every 64th output element of every product; the Cholesky factor of the m x m block is
// computed redundantly by all lanes, the n + 1 triangular solves go one per lane.  The arithmetic --
// operation by operation, summation order and arithmetic type included -- is riccati_q / riccati_gains, so
// the engine returns the same bits as with k_backward; the
// restart-on-failure schedule is the reference's, trivially: one instance, wave-uniform control flow.
// -------------------------------------------------------------------------------------------------
template <class T, class M>
__global__ __launch_bounds__(kBlock) void k_backward_coop(DevArrays<T> A, DevOpts o, int all) {
  constexpr int n = M::n, m = M::m;
  using R = Rec<T, n, m>;
  using RS = rec_scalar_t<T, M>;  // storage type of the expansion / gain records
  using RR = Rec<RS, n, m>;
  using S = T;  // arithmetic type of the recursion (riccati_q / riccati_gains do the same)
  const RS* const EXPp = (const RS*)A.EXP;
  const int lane = threadIdx.x;
  const int b = instance_of_slot(A, blockIdx.x, all);
  if (b < 0) return;  // uniform
  const int N = A.N;
  const unsigned Bp = A.Bp;
  __shared__ S sE[R::EP];                      // expansion record of the current knot
  __shared__ S sP[n * n], sp[n];               // cost-to-go of knot k + 1, then of knot k
  __shared__ S sAtP[n * n], sBtP[m * n];       // A^T P, B^T P
  __shared__ S sQxx[n * n], sQxu[n * m], sQuu[m * m], sQx[n], sQu[m];
  __shared__ S sK[m * n], sd[m], sKtQuu[n * m];
  const S* sAm = sE + R::oAB;
  const S* sBm = sE + R::oAB + n * n;
  constexpr int kPer = (R::EP + kBlock - 1) / kBlock;  // record elements per lane
  auto wsync = [&]() __attribute__((always_inline)) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); };

  // running cost in knot order (ilqr.hpp:326-334)
  double J0 = 0.0;
  for (int base = 0; base <= N; base += kBlock) {
    const int kk = base + lane;
    const double v = (double)A.costs[(unsigned)(kk <= N ? kk : N) * Bp + (unsigned)b];
    for (int j = 0; j < kBlock && base + j <= N; ++j) J0 += __shfl(v, j);
  }
  double rho = A.rho_reg[b], drho = A.drho[b];
  double dV0 = 0.0, dV1 = 0.0;  // zeroed once, NOT per retry (quirk Q4)
  int max_reg_count = 0;
  int status = A.status[b];
  bool need = N > 0;
  auto fetch = [&](int k, S* r) __attribute__((always_inline)) {
    const RS* rec = RECP(EXPp, k, RR::EP);
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int e = lane + j * kBlock;
      r[j] = (S)rec[e < R::eE ? e : R::eE - 1];
    }
  };
  auto put = [&](const S* r) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int e = lane + j * kBlock;
      if (e < R::EP) sE[e] = r[j];
    }
  };
  auto store_ctg = [&](int k) __attribute__((always_inline)) {
    T* c = RECP(A.CTG, k, R::CP);
    for (int e = lane; e < n * n; e += kBlock) c[R::oP + e] = (T)sP[e];
    if (lane < n) c[R::op + lane] = (T)sp[lane];
  };
  while (need) {
    // CalcTerminalCostToGo (knot_point_function_type.hpp:135-138)
    {
      const RS* rec = RECP(EXPp, N, RR::EP);
      for (int e = lane; e < n * n; e += kBlock) sP[e] = (S)rec[R::oLxx + e];
      if (lane < n) sp[lane] = (S)rec[R::oLx + lane];
    }
    wsync();
    if (A.record_ctg) store_ctg(N);
    bool failed = false;
    S nxt[kPer];
    fetch(N - 1, nxt);
    for (int k = N - 1; k >= 0; --k) {
      put(nxt);
      if (k > 0) fetch(k - 1, nxt);  // in flight while this knot is processed
      wsync();
      // ---- riccati_q: A^T P, B^T P ----
      for (int e = lane; e < n * n + m * n; e += kBlock) {
        if (e < n * n) {
          const int i = e % n, j = e / n;
          S s = S(0);
#pragma unroll
          for (int l = 0; l < n; ++l) s += sAm[l + i * n] * sP[l + j * n];
          sAtP[i + j * n] = s;
        } else {
          const int f = e - n * n, i = f % m, j = f / m;
          S s = S(0);
#pragma unroll
          for (int l = 0; l < n; ++l) s += sBm[l + i * n] * sP[l + j * n];
          sBtP[i + j * m] = s;
        }
      }
      wsync();
      // ---- Qxx, Qxu, Quu, Qx, Qu ----
      constexpr int c1 = n * n, c2 = c1 + n * m, c3 = c2 + m * m, c4 = c3 + n, c5 = c4 + m;
      for (int e = lane; e < c5; e += kBlock) {
        if (e < c1) {
          const int i = e % n, j = e / n;
          S s = S(0);
#pragma unroll
          for (int l = 0; l < n; ++l) s += sAtP[i + l * n] * sAm[l + j * n];
          sQxx[e] = sE[R::oLxx + e] + s;
        } else if (e < c2) {
          const int f = e - c1, i = f % n, j = f / n;
          S s = S(0);
#pragma unroll
          for (int l = 0; l < n; ++l) s += sAtP[i + l * n] * sBm[l + j * n];
          sQxu[f] = sE[R::oLxu + f] + s;
        } else if (e < c3) {
          const int f = e - c2, i = f % m, j = f / m;
          S s = S(0);
#pragma unroll
          for (int l = 0; l < n; ++l) s += sBtP[i + l * m] * sBm[l + j * n];
          sQuu[f] = sE[R::oLuu + f] + s;
        } else if (e < c4) {
          const int i = e - c3;
          S s = S(0);
#pragma unroll
          for (int l = 0; l < n; ++l) s += sAm[l + i * n] * sp[l];
          sQx[i] = sE[R::oLx + i] + s;
        } else {
          const int i = e - c4;
          S s = S(0);
#pragma unroll
          for (int l = 0; l < n; ++l) s += sBm[l + i * n] * sp[l];
          sQu[i] = sE[R::oLu + i] + s;
        }
      }
      wsync();
      // ---- riccati_gains: Eigen::LLT of Quu + rho I (every lane, redundantly) ----
      S L[m * m], Linv[m];
#pragma unroll
      for (int e = 0; e < m * m; ++e) L[e] = sQuu[e];
#pragma unroll
      for (int i = 0; i < m; ++i) L[i + i * m] += S(rho);
      bool ok = true;
#pragma unroll
      for (int j = 0; j < m; ++j) {
        S xjj = L[j + j * m];
#pragma unroll
        for (int l = 0; l < j; ++l) xjj -= L[j + l * m] * L[j + l * m];
        if (xjj <= S(0)) ok = false;
        const S ljj = sqrt_(xjj);
        L[j + j * m] = ljj;
        Linv[j] = S(1) / ljj;
#pragma unroll
        for (int i = j + 1; i < m; ++i) {
          S s = L[i + j * m];
#pragma unroll
          for (int l = 0; l < j; ++l) s -= L[i + l * m] * L[j + l * m];
          L[i + j * m] = s * Linv[j];
        }
      }
      if (!ok) {  // uniform: ilqr.hpp:409-427, raise the regularisation and restart the sweep
        increase_reg(o, &rho, &drho);
        if (rho >= o.bp_reg_max) max_reg_count++;
        if (max_reg_count >= o.bp_reg_fail_threshold) {
          status = ALTRO_BACKWARD_PASS_REGULARIZATION_FAILED;
          need = false;
        }
        failed = true;
        break;
      }
      // ---- K = -(L L^T)^-1 Qxu^T, d = -(L L^T)^-1 Qu: one right-hand side per lane ----
      if (lane <= n) {
        const int j = lane;
        S col[m];
#pragma unroll
        for (int i = 0; i < m; ++i) col[i] = (j < n) ? sQxu[(j < n ? j : 0) + i * n] : sQu[i];
#pragma unroll
        for (int i = 0; i < m; ++i) {
          S s = col[i];
#pragma unroll
          for (int l = 0; l < i; ++l) s -= L[i + l * m] * col[l];
          col[i] = s * Linv[i];
        }
#pragma unroll
        for (int i = m - 1; i >= 0; --i) {
          S s = col[i];
#pragma unroll
          for (int l = i + 1; l < m; ++l) s -= L[l + i * m] * col[l];
          col[i] = s * Linv[i];
        }
#pragma unroll
        for (int i = 0; i < m; ++i) {
          if (j < n)
            sK[i + j * m] = -col[i];
          else
            sd[i] = -col[i];
        }
      }
      wsync();
      // ---- K^T Quu ----
      for (int e = lane; e < n * m; e += kBlock) {
        const int i = e % n, j = e / n;
        S s = S(0);
#pragma unroll
        for (int l = 0; l < m; ++l) s += sK[l + i * m] * sQuu[l + j * m];
        sKtQuu[e] = s;
      }
      wsync();
      // ---- cost-to-go with the un-regularised Q (knot_point_function_type.hpp:220-230) ----
      for (int e = lane; e < n * n + n; e += kBlock) {
        if (e < n * n) {
          const int i = e % n, j = e / n;
          S a = S(0), bq = S(0), c = S(0);
#pragma unroll
          for (int l = 0; l < m; ++l) {
            a += sKtQuu[i + l * n] * sK[l + j * m];
            bq += sK[l + i * m] * sQxu[j + l * n];
            c += sQxu[i + l * n] * sK[l + j * m];
          }
          sP[e] = sQxx[e] + a + bq + c;
        } else {
          const int i = e - n * n;
          S a = S(0), bq = S(0), c = S(0);
#pragma unroll
          for (int l = 0; l < m; ++l) {
            a += sKtQuu[i + l * n] * sd[l];
            bq += sK[l + i * m] * sQu[l];
            c += sQxu[i + l * n] * sd[l];
          }
          sp[i] = sQx[i] + a + bq + c;
        }
      }
      // ---- expected decrease (every lane), gains out ----
      {
        S v0 = S(0), v1 = S(0);
#pragma unroll
        for (int i = 0; i < m; ++i) {
          v0 += sd[i] * sQu[i];
          S s = S(0);
#pragma unroll
          for (int l = 0; l < m; ++l) s += sQuu[i + l * m] * sd[l];
          v1 += sd[i] * s;
        }
        dV0 += (double)v0;
        dV1 += (double)(S(0.5) * v1);
      }
      {
        RS* kd = RECP((RS*)A.KD, k, RR::KP);
        for (int e = lane; e < RR::KP; e += kBlock) {
          S v = S(0);
          if (e >= R::oK && e < R::oK + m * n) v = sK[e - R::oK];
          if (e >= R::oD && e < R::oD + m) v = sd[e - R::oD];
          kd[e] = (RS)v;
        }
      }
      wsync();
      if (A.record_ctg) store_ctg(k);
    }
    if (!failed) need = false;  // sweep completed
  }
  if (lane != 0) return;
  A.J0[b] = J0;
  if (A.need_init_cost[b]) {
    A.initial_cost[b] = J0;
    A.need_init_cost[b] = 0;
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
// iLQR::BackwardPass on the fp64 matrix cores (n = 3, m = 2: the unicycle of the headline config).
//
// v_mfma_f64_4x4x4_4b_f64 multiplies FOUR independent 4x4x4 blocks per instruction.  Lane layout
// (probed on gfx950, scripts/probes/mfma_f64_probe.hip): with r = lane/16, blk = (lane/4)%4,
// c = lane%4,
//     A[blk][i][k] sits in lane (r=k, c=i),  B[blk][k][j] in lane (r=k, c=j),  D[blk][i][j] in (r=i, c=j).
// So a tile kept in "D form" (lane (r,c) holds X[r][c]) is directly the RIGHT operand of the next
// product, and the same register used as the LEFT operand means X^T.  The Riccati step only ever
// needs A^T., B^T., K^T., Qux^T. on the left (and the symmetric P, Quu, Quu^-1), so the whole
// recursion chains through the matrix cores with no lane shuffles:
//     W_A = P A, W_B = P B                                   (P in D form = P^T = P)
//     [Qxx|Qx] = [lxx|lx] + A^T [W_A|p]     [Qux|Qu] = [lux|lu] + B^T [W_A|p]     Quu = luu + B^T W_B
//     [K|d] = -(Quu + rho I)^-1 [Qux|Qu]                     (2x2 Cholesky in VALU, explicit inverse)
//     G = Quu [K|d]
//     [P|p] = [Qxx|Qx] + K^T G + K^T [Qux|Qu] + Qux^T [K|d]  (3 accumulating MFMAs, reference order)
// 10 MFMAs + ~150 VALU per knot instead of ~330 fp64 VALU, and each wavefront carries 4 instances
// (16 lanes each), so a 4096-instance batch fills all 1024 SIMDs.  Vectors ride along as the 4th
// column of the 4x4 tiles.  Rounding differs from the VALU kernel only in association
// (A^T(PA) vs (A^T P)A, explicit 2x2 inverse); semantics (restart on Cholesky failure, quirks Q3/Q4)
// are identical.
// -------------------------------------------------------------------------------------------------
ALTRO_DEV double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  // y <- y + y * (0.5 - 0.5 x y^2): two steps take the ~2^-26 hardware estimate to full precision
  double h = 0.5 * x * y;
  double e = fma(-h, y, 0.5);
  y = fma(y, e, y);
  h = 0.5 * x * y;
  e = fma(-h, y, 0.5);
  y = fma(y, e, y);
  return y;
}
ALTRO_DEV double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
  // two Newton steps on the ~2^-26 hardware estimate: y <- y + y (1 - x y)
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  return y;
}
ALTRO_DEV double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// Cross-lane moves of the MFMA layout without the LDS crossbar.  quad_bcast<J>: every lane reads the
// value of lane c = J of its own quad (one instance's row r) -- a DPP quad_perm.  rows01: the values
// that rows r = 0 and r = 1 (lanes 0-15 / 16-31) hold, delivered to both rows (v_permlane16_swap
// exchanges row 1 of its first operand with row 0 of its second).  Rows 2 and 3 receive their own
// pair (2, 3), which the callers never use.
template <int J>
ALTRO_DEV double quad_bcast(double x) {
  constexpr int ctrl = J | (J << 2) | (J << 4) | (J << 6);
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
ALTRO_DEV void rows01(double x, double& from_row0, double& from_row1) {
  const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto bb = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  from_row0 = __hiloint2double((int)bb[0], (int)a[0]);
  from_row1 = __hiloint2double((int)bb[1], (int)a[1]);
}

#ifndef ALTRO_BWD_AHEAD
#define ALTRO_BWD_AHEAD 6
#endif
constexpr int kBwdAhead = ALTRO_BWD_AHEAD;  // knots per prefetch block of the MFMA backward pass (A/B builds: profiles/r05_experiments.txt)
constexpr int kBwdFrontPad = 2 * kBwdAhead;  // records in front of knot 0 that the prefetch may touch
constexpr int kBwdChunk = 126;  // knots of gains buffered in LDS between two bulk stores (4 instances: 32 KiB)

// Body of the MFMA backward pass for one wavefront (lane = 0..63; the instance of block blk is slot
// slot_base + blk of the launch).  sKD: LDS buffer of kBwdChunk * 4 * KP + 64 doubles.
// FUSED (k_sweep_fused): one instance per wavefront (block 0), the gains go straight into the forward
// pass's LDS block sKDf[k * KP + e] (+ a junk slot at sKDf[fused_junk + lane]) and are also written to
// A.KD at the end; the running cost J0 is summed by the other wave; dV0 / dV1 are handed over in fh[1..2].
// T is the STORAGE type of the engine (records, gains, costs); the recursion itself always runs in fp64 on
// the fp64 matrix cores -- an fp32 engine reads float tiles, converts exactly, and rounds the gains once.
// SPEC (the fourth wave of k_sweep_fused): the backward pass of the NEXT iteration, run beside the forward pass of
// this one on the assumption that the line search rejects every trial -- then the trajectory and the multipliers do
// not change, the expansions of the next iteration are the ones in memory, and the only input that differs is the
// regularisation (spec_rho, spec_drho = what phase 3 will set).  Nothing is written to global memory: the gains go to
// a second LDS block (sKDf), the hand-over values to fh[1], [2], [4], [5], the regularisation that was used to fh[6]
// and fh[7] = 1 if the pass went through without a Cholesky failure (a failure simply invalidates the speculation).
// The wave takes part in the workgroup barriers of the forward pass: one after every knot with even index (the
// forward waves sync once per pair of knots), counted in *nbar for the caller to top up.  Plain s_barrier: nothing
// this wave writes is read before the kernel's own __syncthreads.  (Schedules that decouple the two paces -- 70 % or
// 88 % of the recursion's knots spread over the loop's barriers, the rest behind barriers A / S / V -- measured
// 4 - 9 % slower than this lock step: profiles/r02_experiments_not_kept.txt.)
template <class T, class M, bool CTG, bool FUSED, bool SPEC = false, int AHEAD = kBwdAhead>
ALTRO_DEV void backward_mfma_body(const DevArrays<T>& A, const DevOpts& o, int all, int lane, int slot_base,
                                  double* sKD, T* sKDf, int fused_junk, double* fh, double spec_rho = 0.0,
                                  double spec_drho = 0.0, int* nbar = nullptr, int b_fixed = -1) {
  static_assert(!SPEC || FUSED, "the speculative pass is a variant of the fused one");
  // 4 x 4 tiles with the vectors riding as column n: n <= 3 states, m <= 2 controls (round 4: was n = 3, m = 2 only -- the
  // tile offsets below are generic; what m = 1 changes is the 2 x 2 inverse, see qc)
  static_assert(M::n >= 1 && M::n <= 3 && M::m >= 1 && M::m <= 2, "4x4 MFMA backward pass: n <= 3 (one tile column for the vectors), m <= 2");
  constexpr int n = M::n, m = M::m;
  using R = Rec<T, n, m>;
  using RS = rec_scalar_t<T, M>;  // storage type of the expansion / gain records in HBM
  using RR = Rec<RS, n, m>;
  static_assert(RR::KP == R::KP, "the gain chunk in LDS is laid out like the stored record");
  const int r = lane >> 4, c = lane & 3, blk = (lane >> 2) & 3;
  const int b0 = (FUSED && blk != 0) ? -1 : ((FUSED && b_fixed >= 0) ? b_fixed : instance_of_slot(A, slot_base + blk, all));
  const bool inst_on = b0 >= 0;
  if (__ballot(inst_on) == 0ull) return;
  const int b = inst_on ? b0 : 0;
  const int N = A.N;
  const unsigned Bp = A.Bp;
  // element of each tile this lane owns (offset inside the expansion record; -1: structural zero)
  const int offA = (r < n && c < n) ? R::oAB + r + c * n : -1;
  const int offB = (r < n && c < m) ? R::oAB + n * n + r + c * n : -1;
  const int off1 = (r < n) ? (c < n ? R::oLxx + r + c * n : R::oLx + r) : -1;          // [lxx | lx]
  const int off2 = (r < m) ? (c < n ? R::oLxu + c + r * n : R::oLu + r) : -1;          // [lux | lu]
  const int off3 = (r < m && c < m) ? R::oLuu + r + c * m : -1;                        // luu
  const int offKD = (r < m) ? (c < n ? R::oK + r + c * m : R::oD + r) : -1;            // [K | d]
  const int offCT = (r < n) ? (c < n ? R::oP + r + c * n : R::op + r) : -1;            // [P | p]
  // Every tile load is unconditional: lanes that own a structural zero read the zeroed pad record behind
  // the last knot with stride 0, and the prefetch cursor may run below knot 0 into the front pad
  // (kBwdFrontPad records, see the allocation).  Offsets are 32-bit BYTE offsets from the start of the
  // front pad, so a load is one instruction (scalar base + vector offset) and a cursor step one
  // subtraction; the engine only selects this kernel when the array is smaller than 4 GiB.
  const unsigned strideB = Bp * (unsigned)RR::EP * (unsigned)sizeof(RS);
  const unsigned frontB = (unsigned)kBwdFrontPad * strideB;
  const unsigned zoff = frontB + (unsigned)(N + 1) * strideB;
  const unsigned rec0 = frontB + (unsigned)b * (unsigned)RR::EP * (unsigned)sizeof(RS);
  const unsigned sA = offA >= 0 ? strideB : 0u, sB = offB >= 0 ? strideB : 0u, s1 = off1 >= 0 ? strideB : 0u,
                 s2 = off2 >= 0 ? strideB : 0u, s3 = off3 >= 0 ? strideB : 0u;
  constexpr unsigned kES = (unsigned)sizeof(RS);
  const unsigned bA = offA >= 0 ? rec0 + kES * offA : zoff, bB = offB >= 0 ? rec0 + kES * offB : zoff,
                 b1 = off1 >= 0 ? rec0 + kES * off1 : zoff, b2 = off2 >= 0 ? rec0 + kES * off2 : zoff,
                 b3 = off3 >= 0 ? rec0 + kES * off3 : zoff;
  const char* __restrict__ Eb = reinterpret_cast<const char*>(A.EXP) - (size_t)frontB;
  auto ldE = [&](unsigned off) __attribute__((always_inline)) -> double {
    return (double)*reinterpret_cast<const RS*>(Eb + off);
  };

  struct Tiles {
    double tA, tB, t1, t2, t3;
  };
  double rho = SPEC ? spec_rho : A.rho_reg[b], drho = SPEC ? spec_drho : A.drho[b];
  bool spec_bad = false;
  double dV0 = 0.0, dV1 = 0.0;  // zeroed once, NOT per retry (quirk Q4)
  int max_reg_count = 0;
  int status = A.status[b];
  bool need = inst_on && N > 0;
  const unsigned verdict_bit = 1u << (blk * 4);
  // Tile loads run a block of kBwdAhead knots ahead of the recursion: HBM / Infinity-Cache latency
  // (~1 us) is longer than one knot of the dependent chain (~0.3 us).  Two register blocks ping-pong;
  // the loads of the next block are issued right after the first knot of the current one, so every
  // wait -- including the conservative one the compiler places at the loop header -- only covers
  // loads that are at least kBwdAhead - 1 knots old.
  constexpr int H = AHEAD;  // (<= kBwdAhead: the front pad is sized for that)
  static_assert(AHEAD >= 1 && AHEAD <= kBwdAhead, "prefetch depth");
  unsigned iA, iB, i1, i2, i3;
  Tiles Sa[H], Sb[H];
  double Pp;
  auto issue = [&](Tiles& S) __attribute__((always_inline)) {
    // unconditional (past knot 0 the cursor walks into the front pad): a load count that does not
    // depend on control flow lets the compiler wait for exactly the set it needs
    S.tA = ldE(iA);
    S.tB = ldE(iB);
    S.t1 = ldE(i1);
    S.t2 = ldE(i2);
    S.t3 = ldE(i3);
    iA -= sA;
    iB -= sB;
    i1 -= s1;
    i2 -= s2;
    i3 -= s3;
  };
  auto prime = [&]() __attribute__((always_inline)) {
    Pp = ldE(b1 + (unsigned)N * s1);  // CalcTerminalCostToGo: [P|p] = [lxx|lx] of knot N
    const int kl = N - 1;
    iA = bA + (unsigned)kl * sA;
    iB = bB + (unsigned)kl * sB;
    i1 = b1 + (unsigned)kl * s1;
    i2 = b2 + (unsigned)kl * s2;
    i3 = b3 + (unsigned)kl * s3;
#pragma unroll
    for (int j = 0; j < H; ++j) issue(Sa[j]);
  };
  prime();  // in flight while the running cost is summed
  bool primed = true;

  // Running cost of the current trajectory, summed in knot order (ilqr.hpp:326-334): the 16 lanes of
  // the instance fetch the per-knot costs side by side, then hand them over one by one.
  double J0 = 0.0;
  if (!FUSED) {
    const int q = r * 4 + c;  // 0..15 inside the instance
    for (int base = 0; base <= N; base += 128) {
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = base + j * 16 + q;
        v[j] = (double)A.costs[(unsigned)(k <= N ? k : N) * Bp + (unsigned)b];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (base + j * 16 > N) break;
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) {
          const double t = __shfl(v[j], (qq >> 2) * 16 + blk * 4 + (qq & 3));
          if (base + j * 16 + qq <= N) J0 += t;
        }
      }
    }
  }
  // The gains are collected in LDS and written out in bulk: a store in the loop would share the memory
  // counter with the prefetched tiles (loads and stores retire out of order with respect to each
  // other), and every wait on a tile would have to drain the whole queue.
  // CTG build only: junk sink of the lanes that own no element -- 64 elements behind the last cost-to-go record (round 6: it
  // used to be the first 64 elements of the line-search candidates, i.e. instance 0's; with several chains of sweeps another
  // chain's backward pass then overwrote candidates that chain 0's forward pass was about to read -- instance 0 of a batch
  // of >= 2048 solved with altro_set_record_ctg(1) took 119 iterations instead of 11)
  T* const sink = A.CTG + (size_t)(unsigned)(N + 1) * Bp * R::CP + lane;
  while (__ballot(need) != 0ull) {
    if (!primed) prime();  // restart after a failed factorisation
    primed = false;
    if (CTG) *((need && offCT >= 0) ? RECP(A.CTG, N, R::CP) + offCT : sink) = (T)Pp;
    bool running = need;
    const bool in_sweep = need;  // (this block's instance takes part in this sweep)
    // write the buffered knots (k_top, k_top - 1, ... in slots 0 .. slot-1) of the four instances out
    // (only of the instances that take part in THIS sweep: in a restarted sweep the slots of a block whose instance finished in
    //  an earlier one hold whatever that sweep's last chunk left there -- the same knots' values only as long as a pass is ONE
    //  chunk; with N > kBwdChunk they are other knots' gains, and writing them out corrupted the finished instance's record.
    //  Found in round 6 by an experimental build with 32-knot chunks on the obstacle batch, whose Cholesky restarts are real.)
    int slot = 0, k_top = N - 1;
    auto flush = [&]() __attribute__((always_inline)) {
      for (int i = lane; i < slot * 4 * R::KP; i += kBlock) {
        const int e = i % R::KP, ib = (i / R::KP) % 4, sl = i / (4 * R::KP);
        const int bi = __shfl(b, ib * 4);  // instance of block ib (its lane r = 0, c = 0)
        const int on = __shfl(in_sweep ? 1 : 0, ib * 4);
        if (on) ((RS*)A.KD)[((size_t)(unsigned)(k_top - sl) * Bp + (unsigned)bi) * RR::KP + e] = (RS)sKD[i];
      }
      k_top -= slot;
      slot = 0;
    };
    auto step = [&](int k, Tiles& S) __attribute__((always_inline)) {
      // The vector column / the gain rows ride along unmasked: used as a LEFT operand they only add
      // a fourth output row, which meets the structurally-zero fourth row of A and B in every later
      // product (and is never stored).
      const double WA = mfma4(Pp, S.tA, 0.0);
      const double WB = mfma4(Pp, S.tB, 0.0);
      const double Waug = (c < n) ? WA : Pp;      // [P A | p]
      const double Q1 = mfma4(S.tA, Waug, S.t1);  // [Qxx | Qx]
      const double Q2 = mfma4(S.tB, Waug, S.t2);  // [Qux | Qu]
      const double Q3 = mfma4(S.tB, WB, S.t3);    // Quu
      // Quu entries to rows 0 and 1 of the instance
      double q00, q10, q01, q11;
      rows01(quad_bcast<0>(Q3), q00, q10);
      rows01(quad_bcast<1>(Q3), q01, q11);
      (void)q01;
      // (Quu + rho I)^-1.  Eigen::LLT fails on a pivot <= 0 (knot_point_function_type.hpp:197-211);
      // its pivots are a and det / a, so the verdict is a <= 0 || det <= 0, and the inverse of the 2x2
      // SPD matrix is the adjugate over det: one reciprocal on the dependent chain instead of two
      // reciprocal square roots in sequence (same O(cond) * eps accuracy as forming L^-T L^-1).
      // (m = 1: Quu is the scalar q00; a unit second diagonal entry makes det = qa, the verdict "qa <= 0" and the
      //  adjugate entry -qc / det = -1 / qa -- Eigen's 1 x 1 LLT up to the rounding of one reciprocal)
      const double qa = q00 + rho, qc = (m >= 2) ? q11 + rho : 1.0;
      const double det = fma(qa, qc, -(q10 * q10));
      double rd = rcp_nr(det);
      pin(rd);  // computed by every lane, not inside a region of the lanes that hold the 2 x 2 block
      // rows 2 and 3 did not receive Quu: every lane reads the verdict of lane (r = 0, c = 0) of its block
      const bool fail = ((unsigned)__ballot((qa <= 0.0) || (det <= 0.0)) & verdict_bit) != 0u;
      // -(Quu + rho I)^-1, this lane's entry (zero outside the 2x2 block)
      // (rows 2 and 3 hold det = 0 / rd = inf: the zero must be selected after the product)
      const double MinvNeg = (r < m && c < m) ? (r == c ? (r == 0 ? -qc : -qa) : q10) * rd : 0.0;
      const double KD = mfma4(MinvNeg, Q2, 0.0);  // [K | d], gains from the REGULARISED Quu (quirk Q3)
      const double G = mfma4(Q3, KD, 0.0);        // Quu [K | d] with the UN-regularised Quu
      // [P|p] = [Qxx|Qx] + K^T [Qux|Qu] + Qux^T [K|d] + K^T Quu [K|d]; the term that needs G goes last
      double Pn = mfma4(Q2, KD, Q1);
      Pn = mfma4(KD, Q2, Pn);
      Pn = mfma4(KD, G, Pn);
      // Branch-free state update: a divergent branch costs ~50 cycles on this chain.  The only real
      // branch is the (wave-uniform, rare) regularisation increase.
      const bool failed = running && fail;
      const bool commit = running && !fail;
      running = commit;
      bool gave_up = false;
      if (__ballot(failed) != 0ull) {
        spec_bad = true;
        // ilqr.hpp:409-427: raise the regularisation and restart the sweep (next round)
        double rho2 = rho, drho2 = drho;
        increase_reg(o, &rho2, &drho2);
        const int cnt2 = max_reg_count + (rho2 >= o.bp_reg_max ? 1 : 0);
        const bool give = cnt2 >= o.bp_reg_fail_threshold;
        rho = failed ? rho2 : rho;
        drho = failed ? drho2 : drho;
        max_reg_count = failed ? cnt2 : max_reg_count;
        status = (failed && give) ? (int)ALTRO_BACKWARD_PASS_REGULARIZATION_FAILED : status;
        gave_up = failed && give;
      }
      Pp = commit ? Pn : Pp;
      pin(Pp);
      // expected cost decrease d^T Qu, d^T Quu d: every lane accumulates its own product (rows 0 and 1
      // of the vector column are the meaningful ones); the two rows are added once, after the sweep
      double KDc = commit ? KD : 0.0;
      pin(KDc);  // a select, then two unconditional FMAs -- not an if / else around them
      dV0 = fma(KDc, Q2, dV0);
      dV1 = fma(KDc, G, dV1);
      // gains into the LDS block (lanes with nothing to store hit a junk slot)
      if (FUSED) {
        int idx = (commit && offKD >= 0) ? k * R::KP + offKD : fused_junk + lane;
        pin(idx);
        sKDf[idx] = (T)(RS)KD;  // as stored
      } else {
        int idx = (commit && offKD >= 0) ? (slot * 4 + blk) * R::KP + offKD : kBwdChunk * 4 * R::KP + lane;
        pin(idx);
        sKD[idx] = KD;
      }
      if (CTG) *((commit && offCT >= 0) ? RECP(A.CTG, k, R::CP) + offCT : sink) = (T)Pn;
      need = need && !gave_up;
      slot++;
      if (SPEC && nbar && (k & (kSyncFused - 1)) == 0) {  // the forward waves' barrier of this stretch of knots
        __builtin_amdgcn_s_barrier();
        ++*nbar;
      }
    };
    int k = N - 1;
    // one block of H knots from `Cur`; false once knot 0 has been processed
    auto block = [&](Tiles* Cur, Tiles* Nxt) __attribute__((always_inline)) -> bool {
#pragma unroll
      for (int j = 0; j < H; ++j) {
        step(k - j, Cur[j]);
        if (j == 0) {
#pragma unroll
          for (int jj = 0; jj < H; ++jj) issue(Nxt[jj]);
        }
        if (k - j == 0) return false;
      }
      k -= H;
      if (!FUSED && slot + H > kBwdChunk) flush();
      return true;
    };
    for (;;) {
      if (!block(Sa, Sb)) break;
      if (!block(Sb, Sa)) break;
    }
    need = need && !running;  // instances that reached knot 0 are done
    if (!FUSED) flush();
    if (SPEC) need = false;  // one sweep only: a failed factorisation invalidates the speculation
  }
  {
    double a0, a1;
    rows01(dV0, a0, a1);
    dV0 = a0 + a1;
    rows01(dV1, a0, a1);
    dV1 = a0 + a1;
  }
  if (!inst_on || r != 0) return;
  if (c == 3) {  // the lane that holds row 0 of the vector column
    if (!SPEC) {
      A.dV0[b] = dV0;
      A.dV1[b] = 0.5 * dV1;
    }
    if (FUSED) {
      fh[1] = dV0;
      fh[2] = 0.5 * dV1;
    }
  }
  if (c != 0) return;
  if (SPEC) {
    fh[6] = rho;  // stats_.Log("reg", rho_) of the speculated iteration
    decrease_reg(o, &rho, &drho);
    fh[4] = rho;
    fh[5] = drho;
    fh[7] = spec_bad ? 0.0 : 1.0;
    return;
  }
  if (!FUSED) {
    A.J0[b] = J0;
    if (A.need_init_cost[b]) {
      A.initial_cost[b] = J0;
      A.need_init_cost[b] = 0;
    }
  }
  const double rho_used = rho;
  A.reg_log[b] = rho_used;  // stats_.Log("reg", rho_)
  decrease_reg(o, &rho, &drho);
  A.rho_reg[b] = rho;
  A.drho[b] = drho;
  A.status[b] = status;
  if (FUSED) {  // the forward pass of the same kernel must not depend on L1 seeing these stores
    fh[4] = rho;
    fh[5] = drho;
  }
}

template <class T, class M, bool CTG>
__global__ __launch_bounds__(kBlock) void k_backward_mfma(DevArrays<T> A, DevOpts o, int all) {
  using R = Rec<T, M::n, M::m>;
  __shared__ double sKD[kBwdChunk * 4 * R::KP + kBlock];  // + one junk slot per lane
  backward_mfma_body<T, M, CTG, false>(A, o, all, threadIdx.x, xcd_block((int)blockIdx.x, (int)gridDim.x, A.xcd_remap) * 4, sKD, nullptr, 0,
                                       nullptr);
}

// -------------------------------------------------------------------------------------------------
// iLQR::BackwardPass on the 16x16x4 fp64 matrix cores for the larger models (4 < n <= 12, m <= 4: the
// 2-dof triple integrator and the 12-state model of BASELINE configs[4]): ONE INSTANCE PER WAVEFRONT, every
// matrix of the Riccati step a 16 x 16 tile spread over the 64 lanes.
//
// v_mfma_f64_16x16x4_f64 (lane layout confirmed on gfx950 by scripts/probes/mfma_f64_16x16_probe.hip): with
// r = lane / 16, c = lane % 16, the A operand holds A[i = c][k = r], the B operand B[k = r][j = c], and the
// result D[i][j] sits in lane (r = i % 4, c = j), register i / 4.  A tile kept in that "D form" (register v of
// lane (r, c) = X[r + 4 v][c]) is therefore chunk v of the RIGHT operand of a later product as it stands, and
// the same register passed as the LEFT operand means chunk v of X^T.  With
//     prod(X, Z) = X^T Z = sum over the row chunks v of mfma(X[v], Z[v], .)
// the recursion of knot_point_function_type.hpp:149-235 chains through the matrix cores with no lane shuffle,
// vectors riding along as column n of the tiles, exactly like the 4 x 4 kernel above:
//     W_A = prod([P|p], A) = P A (+ a row p^T A that only ever meets zero rows), W_B = prod([P|p], B)
//     [Qxx|Qx] = [lxx|lx] + prod(A, [W_A|p])    [Qux|Qu] = [lux|lu] + prod(B, [W_A|p])    Quu = luu + prod(B, W_B)
//     [K|d] = prod(-(Quu + rho I)^-1, [Qux|Qu])          (gains from the REGULARISED Quu, quirk Q3)
//     G = prod(Quu, [K|d])                               (un-regularised)
//     [P|p] = [Qxx|Qx] + prod([Qux|Qu], [K|d]) + prod([K|d], [Qux|Qu]) + prod([K|d], G)
// Products contract over the rows of both operands: ceil(n / 4) instructions when those are state rows, ONE when
// they are control rows (m <= 4): 5 * 3 + 5 = 20 matrix instructions per knot for n = 12 (64 cycles each) instead
// of ~36 kflop on the vector ALUs through LDS (k_backward_coop: ~4 us per knot).  The m x m Cholesky
// factorisation (the reference's failure verdict: a pivot <= 0) and the inverse are computed by every lane from an
// LDS broadcast of Quu, each lane solving for its own column.  Tile loads run a block of knots ahead of the
// recursion; the gains are buffered in LDS and written out every kM16Chunk knots.
// -------------------------------------------------------------------------------------------------
typedef double mfma16_acc __attribute__((ext_vector_type(4)));
constexpr int kM16Ahead = 2;   // knots per prefetch block
constexpr int kM16Chunk = 32;  // knots of gains buffered in LDS between two bulk stores

template <class T, class M, bool CTG>
__global__ __launch_bounds__(kBlock) void k_backward_mfma16(DevArrays<T> A, DevOpts o, int all) {
  constexpr int n = M::n, m = M::m;
  static_assert(n >= 4 && n <= 12 && m >= 1 && m <= 4, "16x16 MFMA backward pass: 4 <= n <= 12 (one spare column for the vectors), m <= 4");
  using R = Rec<T, n, m>;
  using RS = rec_scalar_t<T, M>;
  using RR = Rec<RS, n, m>;
  constexpr int RC = (n + 3) / 4;  // row chunks of a state-sized tile
  const int lane = threadIdx.x;
  const int r = lane >> 4, c = lane & 15;
  const int b = instance_of_slot(A, blockIdx.x, all);
  if (b < 0) return;  // uniform
  const int N = A.N;
  const unsigned Bp = A.Bp;
  __shared__ double sKD[kM16Chunk * RR::KP];
  __shared__ double sQ[16];

  // ---- tile element owned by this lane in register v (byte offset inside the expansion record; the lanes that
  //      own a structural zero read the zeroed pad record behind knot N with stride 0) ----
  // 32-bit BYTE offsets from the start of the front pad (kBwdFrontPad records in front of knot 0, which the
  // last, unused prefetch block may touch): a load is scalar base + vector offset, a cursor step one subtraction
  constexpr unsigned kES = (unsigned)sizeof(RS);
  const unsigned strideB = Bp * (unsigned)RR::EP * kES;
  const unsigned frontB = (unsigned)kBwdFrontPad * strideB;
  const unsigned zoff = frontB + (unsigned)(N + 1) * strideB;
  const unsigned rec0 = frontB + (unsigned)b * (unsigned)RR::EP * kES;
  const char* __restrict__ Eb = reinterpret_cast<const char*>(A.EXP) - (size_t)frontB;
  auto ldE = [&](unsigned off) __attribute__((always_inline)) -> double {
    return (double)*reinterpret_cast<const RS*>(Eb + off);
  };
  static_assert(2 * kM16Ahead <= kBwdFrontPad, "the prefetch may run two blocks below knot 0");
  for (int i = lane; i < kM16Chunk * RR::KP; i += kBlock) sKD[i] = 0.0;  // record padding is stored too
  constexpr int NT = 3 * RC + 2;  // tile registers per knot: A, B, [lxx|lx] (RC each), [lux|lu], luu
  unsigned base[NT], step[NT];
#pragma unroll
  for (int v = 0; v < RC; ++v) {
    const int row = r + 4 * v;
    const int eA = (row < n && c < n) ? R::oAB + row + c * n : -1;
    const int eB = (row < n && c < m) ? R::oAB + n * n + row + c * n : -1;
    const int e1 = (row < n) ? (c < n ? R::oLxx + row + c * n : (c == n ? R::oLx + row : -1)) : -1;
    base[v] = eA >= 0 ? rec0 + kES * eA : zoff;
    step[v] = eA >= 0 ? strideB : 0u;
    base[RC + v] = eB >= 0 ? rec0 + kES * eB : zoff;
    step[RC + v] = eB >= 0 ? strideB : 0u;
    base[2 * RC + v] = e1 >= 0 ? rec0 + kES * e1 : zoff;
    step[2 * RC + v] = e1 >= 0 ? strideB : 0u;
  }
  {
    const int e2 = (r < m) ? (c < n ? R::oLxu + c + r * n : (c == n ? R::oLu + r : -1)) : -1;  // [lux | lu]
    const int e3 = (r < m && c < m) ? R::oLuu + r + c * m : -1;
    base[3 * RC] = e2 >= 0 ? rec0 + kES * e2 : zoff;
    step[3 * RC] = e2 >= 0 ? strideB : 0u;
    base[3 * RC + 1] = e3 >= 0 ? rec0 + kES * e3 : zoff;
    step[3 * RC + 1] = e3 >= 0 ? strideB : 0u;
  }
  const int offKD = (r < m) ? (c < n ? R::oK + r + c * m : (c == n ? R::oD + r : -1)) : -1;  // [K | d], register 0
  struct Tiles {
    double t[NT];
  };
  unsigned cur[NT];
  auto issue = [&](Tiles& S) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      S.t[i] = ldE(cur[i]);
      cur[i] -= step[i];  // past knot 0 the cursor walks into the front pad (those tiles are never used)
    }
  };

  // running cost in knot order (ilqr.hpp:326-334)
  double J0 = 0.0;
  for (int kb = 0; kb <= N; kb += kBlock) {
    const int kk = kb + lane;
    const double v = (double)A.costs[(unsigned)(kk <= N ? kk : N) * Bp + (unsigned)b];
    for (int j = 0; j < kBlock && kb + j <= N; ++j) J0 += __shfl(v, j);
  }
  double rho = A.rho_reg[b], drho = A.drho[b];
  double dV0 = 0.0, dV1 = 0.0;  // zeroed once, NOT per retry (quirk Q4)
  int max_reg_count = 0;
  int status = A.status[b];
  bool need = N > 0;

  auto mfma = [](double a, double bb, mfma16_acc acc) __attribute__((always_inline)) -> mfma16_acc {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc, 0, 0, 0);
  };
  const mfma16_acc zero4 = {0.0, 0.0, 0.0, 0.0};

  while (need) {
    // CalcTerminalCostToGo (knot_point_function_type.hpp:135-138): [P|p] = [lxx|lx] of knot N
    double Pp[RC];
#pragma unroll
    for (int v = 0; v < RC; ++v) Pp[v] = ldE(base[2 * RC + v] + (unsigned)N * step[2 * RC + v]);
    if (CTG) {
#pragma unroll
      for (int v = 0; v < RC; ++v) {
        const int row = r + 4 * v;
        if (row < n && c <= n) RECP(A.CTG, N, R::CP)[c < n ? R::oP + row + c * n : R::op + row] = (T)Pp[v];
      }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) cur[i] = base[i] + (unsigned)(N - 1) * step[i];
    Tiles Sa[kM16Ahead], Sb[kM16Ahead];
#pragma unroll
    for (int j = 0; j < kM16Ahead; ++j) issue(Sa[j]);
    bool failed = false;
    int slot = 0, k_top = N - 1;
    auto flush = [&]() __attribute__((always_inline)) {
      RS* const KDp = (RS*)A.KD;
      for (int i = lane; i < slot * RR::KP; i += kBlock) {
        const int sl = i / RR::KP, e = i - sl * RR::KP;
        KDp[((size_t)(unsigned)(k_top - sl) * Bp + (unsigned)b) * RR::KP + e] = (RS)sKD[i];
      }
      k_top -= slot;
      slot = 0;
    };
    // one knot; returns false on a failed factorisation
    auto knot = [&](int k, const Tiles& S) __attribute__((always_inline)) -> bool {
      const double* tA = S.t;
      const double* tB = S.t + RC;
      const double* t1 = S.t + 2 * RC;
      const double t2 = S.t[3 * RC], t3 = S.t[3 * RC + 1];
      // W_A = P A, W_B = P B (contraction over the state rows of [P|p] and of A / B)
      mfma16_acc WA = zero4, WB = zero4;
#pragma unroll
      for (int v = 0; v < RC; ++v) {
        WA = mfma(Pp[v], tA[v], WA);
        WB = mfma(Pp[v], tB[v], WB);
      }
      // [W_A | p]: column n takes the vector
      double Waug[RC];
#pragma unroll
      for (int v = 0; v < RC; ++v) Waug[v] = (c < n) ? WA[v] : Pp[v];
      mfma16_acc Q2 = {t2, 0.0, 0.0, 0.0}, Q3 = {t3, 0.0, 0.0, 0.0}, Q1;
#pragma unroll
      for (int v = 0; v < 4; ++v) Q1[v] = v < RC ? t1[v < RC ? v : 0] : 0.0;
#pragma unroll
      for (int v = 0; v < RC; ++v) {
        Q3 = mfma(tB[v], WB[v], Q3);    // Quu
        Q2 = mfma(tB[v], Waug[v], Q2);  // [Qux | Qu]
      }
#pragma unroll
      for (int v = 0; v < RC; ++v) Q1 = mfma(tA[v], Waug[v], Q1);  // [Qxx | Qx] (not needed before the cost-to-go)
      // ---- (Quu + rho I)^-1: broadcast the m x m block through LDS, factorise redundantly ----
      if (r < m && c < m) sQ[r + c * 4] = Q3[0];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      double L[m * m], Linv[m];
#pragma unroll
      for (int j = 0; j < m; ++j)
#pragma unroll
        for (int i = 0; i < m; ++i) L[i + j * m] = sQ[i + j * 4];
#pragma unroll
      for (int i = 0; i < m; ++i) L[i + i * m] += rho;
      bool ok = true;
#pragma unroll
      for (int j = 0; j < m; ++j) {  // Eigen::LLT (lower): a pivot <= 0 is a failure
        double xjj = L[j + j * m];
#pragma unroll
        for (int l = 0; l < j; ++l) xjj = fma(-L[j + l * m], L[j + l * m], xjj);
        if (xjj <= 0.0) ok = false;
        const double li = rsqrt_nr(ok ? xjj : 1.0);
        Linv[j] = li;
        L[j + j * m] = xjj * li;
#pragma unroll
        for (int i = j + 1; i < m; ++i) {
          double sacc = L[i + j * m];
#pragma unroll
          for (int l = 0; l < j; ++l) sacc = fma(-L[i + l * m], L[j + l * m], sacc);
          L[i + j * m] = sacc * li;
        }
      }
      if (!ok) return false;  // uniform: every lane factorised the same matrix
      // column c of the inverse (each lane its own right-hand side e_c), then this lane's row
      double x[m];
#pragma unroll
      for (int i = 0; i < m; ++i) {
        double sacc = (c == i) ? 1.0 : 0.0;
#pragma unroll
        for (int l = 0; l < i; ++l) sacc = fma(-L[i + l * m], x[l], sacc);
        x[i] = sacc * Linv[i];
      }
#pragma unroll
      for (int i = m - 1; i >= 0; --i) {
        double sacc = x[i];
#pragma unroll
        for (int l = i + 1; l < m; ++l) sacc = fma(-L[l + i * m], x[l], sacc);
        x[i] = sacc * Linv[i];
      }
      double mine = 0.0;
#pragma unroll
      for (int i = 0; i < m; ++i) mine = (r == i) ? x[i] : mine;
      const double MinvNeg = (r < m && c < m) ? -mine : 0.0;
      // [K | d] and Quu [K | d]: contraction over the control rows, one instruction each
      const mfma16_acc KDt = mfma(MinvNeg, Q2[0], zero4);
      const mfma16_acc G = mfma(Q3[0], KDt[0], zero4);
      // [P|p] = [Qxx|Qx] + Qux^T [K|d] + K^T [Qux|Qu] + K^T Quu [K|d]
      mfma16_acc Pn = mfma(Q2[0], KDt[0], Q1);
      Pn = mfma(KDt[0], Q2[0], Pn);
      Pn = mfma(KDt[0], G[0], Pn);
#pragma unroll
      for (int v = 0; v < RC; ++v) Pp[v] = Pn[v];
      // expected cost decrease: column n, rows < m hold d; every lane accumulates, rows are added after the sweep
      dV0 = fma(KDt[0], Q2[0], dV0);
      dV1 = fma(KDt[0], G[0], dV1);
      if (offKD >= 0) sKD[slot * RR::KP + offKD] = KDt[0];
      if (CTG) {
#pragma unroll
        for (int v = 0; v < RC; ++v) {
          const int row = r + 4 * v;
          if (row < n && c <= n) RECP(A.CTG, k, R::CP)[c < n ? R::oP + row + c * n : R::op + row] = (T)Pn[v];
        }
      }
      ++slot;
      return true;
    };
    int k = N - 1;
    auto block = [&](Tiles* Cur, Tiles* Nxt) __attribute__((always_inline)) -> int {  // 1 go on, 0 reached knot 0, -1 failed
#pragma unroll
      for (int j = 0; j < kM16Ahead; ++j) {
        if (!knot(k - j, Cur[j])) return -1;
        if (j == 0) {
#pragma unroll
          for (int jj = 0; jj < kM16Ahead; ++jj) issue(Nxt[jj]);
        }
        if (k - j == 0) return 0;
      }
      k -= kM16Ahead;
      if (slot + kM16Ahead > kM16Chunk) flush();
      return 1;
    };
    int res;
    for (;;) {
      res = block(Sa, Sb);
      if (res != 1) break;
      res = block(Sb, Sa);
      if (res != 1) break;
    }
    if (res < 0) {
      // ilqr.hpp:409-427: raise the regularisation and restart the sweep.  What the expected decrease has
      // accumulated so far stays (quirk Q4) -- including this knot's share: the reference adds it only after a
      // successful factorisation, and knot() returns before touching dV0 / dV1 on failure.
      increase_reg(o, &rho, &drho);
      if (rho >= o.bp_reg_max) max_reg_count++;
      if (max_reg_count >= o.bp_reg_fail_threshold) {
        status = ALTRO_BACKWARD_PASS_REGULARIZATION_FAILED;
        need = false;
      }
      failed = true;
    }
    flush();
    if (!failed) need = false;  // sweep completed
  }
  // d^T Qu, d^T Quu d: rows 0 .. m-1 of column n
  {
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int i = 0; i < m; ++i) {
      a0 += __shfl(dV0, 16 * i + n);
      a1 += __shfl(dV1, 16 * i + n);
    }
    dV0 = a0;
    dV1 = 0.5 * a1;
  }
  if (lane != 0) return;
  A.J0[b] = J0;
  if (A.need_init_cost[b]) {
    A.initial_cost[b] = J0;
    A.need_init_cost[b] = 0;
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
__global__ __launch_bounds__(kBlock) void k_rollout(DevArrays<T> A, const ProblemDesc* __restrict__ pd, int all) {
  constexpr int n = M::n, m = M::m;
  using R = Rec<T, n, m>;
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  if (!all && A.phase[b] != 1) return;
  const unsigned Bp = A.Bp;
  T x[R::nP], u[R::mP], xn[n];
  load_rec<T, R::nP>(A.x0 + (size_t)b * R::nP, x);
  for (int k = 0; k < A.N; ++k) {
    store_rec<T, R::nP>(RECP(A.X, k, R::nP), x);
    load_rec<T, R::mP>(RECP(A.U, k, R::mP), u);
    discrete_step<T, M>(x, u, step_of(A, pd, k), xn, time_of(A, k), model_of(A, k));
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = xn[i];
  }
  store_rec<T, R::nP>(RECP(A.X, A.N, R::nP), x);
}

// -------------------------------------------------------------------------------------------------
// iLQR::Cost (ilqr.hpp:326-334, 758-763): per-knot costs over grid (instance, knot) + c_ stores,
// then a per-instance ordered sum.
// -------------------------------------------------------------------------------------------------
template <class T, class M>
__global__ __launch_bounds__(kBlock) void k_knot_costs(DevArrays<T> A, const ProblemDesc* __restrict__ pd) {
  constexpr int n = M::n, m = M::m;
  using R = Rec<T, n, m>;
  const int b = blockIdx.x * kBlock + threadIdx.x;
  const int k = blockIdx.y;
  if (b >= A.B) return;
  const unsigned Bp = A.Bp;
  T x[R::nP], u[R::mP];
  load_rec<T, R::nP>(RECP(A.X, k, R::nP), x);
#pragma unroll
  for (int i = 0; i < R::mP; ++i) u[i] = T(0);
  if (k < A.N) load_rec<T, R::mP>(RECP(A.U, k, R::mP), u);
  CtxG<T> C(A, b);
  T v;
  int rb;
  const KnotClass& kc = class_of_knot(A, pd, k, &rb);
  A.costs[(unsigned)k * Bp + (unsigned)b] = knot_cost<T, n, m, true>(C, pd, kc, rb, x, u, &v);
}
template <class T>
__global__ __launch_bounds__(kBlock) void k_sum_costs(DevArrays<T> A, double* out) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  double J = 0.0;
  for (int k = 0; k <= A.N; ++k) J += (double)A.costs[(unsigned)k * (unsigned)A.Bp + (unsigned)b];
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
  A.rho_reg[b] = o.bp_reg_initial;
  A.drho[b] = 0.0;
  A.dV0[b] = 0.0;
  A.dV1[b] = 0.0;
  A.need_init_cost[b] = 1;  // stats_.initial_cost = Cost() is taken from the next expansion step
}

// SolverStats::Reset (solver_stats.cpp:31-45)
template <class T>
ALTRO_DEV void reset_stats(const DevArrays<T>& A, int b) {
  A.initial_cost[b] = 0.0;
  A.it_inner[b] = A.it_outer[b] = A.it_total[b] = 0;
  A.cost_cur[b] = A.cost_prev[b] = A.dJ[b] = A.grad[b] = A.viol[b] = 0.0;
  A.alpha[b] = A.z[b] = A.reg_log[b] = 0.0;
  if (A.hist) A.hist_len[b] = 0;
}
template <class T>
__global__ __launch_bounds__(kBlock) void k_reset_stats(DevArrays<T> A) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  reset_stats(A, b);
  A.penmax[b] = 0.0;
}

// AugmentedLagrangianiLQR::Init (al_solver.hpp:287-302) without the (unobservable) initial
// MaxViolation log, + activation of every instance.
// Grid (instances / 64, row chunks): the multipliers and penalties of an instance are a few hundred rows, and one lane
// writing them one after the other was 18 us of every solve (a batch of one as much as a batch of 4096); blockIdx.y
// takes kAlInitRows rows each, the blocks of chunk 0 do the per-instance part.
constexpr int kAlInitRows = 32;
template <class T>
__global__ __launch_bounds__(kBlock) void k_al_init(DevArrays<T> A, const ProblemDesc* __restrict__ pd, DevOpts o) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  {
    const bool zero_lam = o.reset_duals != 0, set_pen = o.initial_penalty > 0;  // quirk Q8
    const int r0 = (int)blockIdx.y * kAlInitRows, r1 = min(r0 + kAlInitRows, pd->total_rows);
    for (int r = r0; r < r1; ++r) {
      if (zero_lam) A.lam[(size_t)r * A.Bp + b] = T(0);
      if (set_pen) A.pen[(size_t)r * A.Bp + b] = T(o.initial_penalty);
    }
  }
  if (blockIdx.y != 0) return;
  reset_stats(A, b);  // stats.Reset()
  // stats.Log("pen", GetMaxPenalty()) of Init (al_solver.hpp:301)
  if (o.initial_penalty > 0) {
    A.penmax[b] = o.initial_penalty;
  } else {
    T v, pm;
    rows_viol_pen(A, pd, b, &v, &pm);
    A.penmax[b] = (double)pm;
  }
  A.status_al[b] = ALTRO_UNSOLVED;
}
// finishing touch of AL Init for the step-level API: log viol (after a cost evaluation) and pen
template <class T>
__global__ __launch_bounds__(kBlock) void k_log_viol_pen(DevArrays<T> A, const ProblemDesc* __restrict__ pd) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  T v, p;
  rows_viol_pen(A, pd, b, &v, &p);
  A.viol[b] = (double)v;
  A.penmax[b] = (double)p;
}
template <class T>
__global__ __launch_bounds__(kBlock) void k_solve_setup(DevArrays<T> A, DevOpts o, int activate) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  begin_inner_solve(A, o, b);
  if (activate) A.phase[b] = 1;
  if (A.seg_end) {  // no segment bookkeeping survives a solve
    A.seg_end[b] = kSegNoEnd;
    A.seg_next[b] = -1;
    A.seg_flag[b] = 0;
    A.seg_streak[b] = 0;
  }
}
// -------------------------------------------------------------------------------------------------
// The start of a whole solve in ONE launch (round 6): what Solve() used to enqueue as k_al_init, a memset of the shadow
// columns' flags, k_solve_setup, k_rollout and the memsets of the sweep counters and of the twins' mailboxes -- seven
// stream operations, 73 us in front of the first iteration of a batch of one (scripts/gpu_latency_trace.sh: 41 us of them
// the open-loop rollout, a single lane that met the latency of its control loads at every knot).  None of the parts reads
// what another writes, so they run side by side over blockIdx.y:
//   y = 0                 per instance: AL Init's statistics (al_solver.hpp:287-302), SolveSetup (ilqr.hpp:629-645),
//                         activation, then iLQR::Rollout (ilqr.hpp:453-459) with the NEXT knot's controls, step, time and
//                         model requested before this knot's RK4 chain starts -- same discrete_step on the same inputs as
//                         k_rollout, same bits
//   y = 1 .. rows_y       AL Init's multiplier / penalty rows, kAlInitRows each (as k_al_init)
//   y > rows_y            zero jobs: word ranges cleared by all threads of those blocks (grid stride)
// The step-level API keeps the separate kernels.
// -------------------------------------------------------------------------------------------------
constexpr int kZeroJobs = 3;
struct ZeroJobs {
  unsigned* p[kZeroJobs];
  unsigned n[kZeroJobs];  // 32-bit words
};
template <class T, class M>
__global__ __launch_bounds__(kBlock) void k_begin_solve(DevArrays<T> A, const ProblemDesc* __restrict__ pd, DevOpts o, int al,
                                                        int rows_y, ZeroJobs z) {
  constexpr int n = M::n;
  using R = Rec<T, M::n, M::m>;
  const int by = (int)blockIdx.y;
  if (by > rows_y) {
    const unsigned nthreads = (gridDim.y - 1u - (unsigned)rows_y) * gridDim.x * kBlock;
    const unsigned t = ((unsigned)(by - rows_y - 1) * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
#pragma unroll
    for (int j = 0; j < kZeroJobs; ++j)
      for (unsigned i = t; i < z.n[j]; i += nthreads) z.p[j][i] = 0u;
    return;
  }
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  if (by > 0) {
    const bool zero_lam = o.reset_duals != 0, set_pen = o.initial_penalty > 0;  // quirk Q8
    const int r0 = (by - 1) * kAlInitRows, r1 = min(r0 + kAlInitRows, pd->total_rows);
    for (int r = r0; r < r1; ++r) {
      if (zero_lam) A.lam[(size_t)r * A.Bp + b] = T(0);
      if (set_pen) A.pen[(size_t)r * A.Bp + b] = T(o.initial_penalty);
    }
    return;
  }
  if (al) {
    reset_stats(A, b);  // stats.Reset()
    if (o.initial_penalty > 0) {  // stats.Log("pen", GetMaxPenalty()) of Init (al_solver.hpp:301)
      A.penmax[b] = o.initial_penalty;
    } else {
      T v, pm;
      rows_viol_pen(A, pd, b, &v, &pm);
      A.penmax[b] = (double)pm;
    }
    A.status_al[b] = ALTRO_UNSOLVED;
  }
  begin_inner_solve(A, o, b);
  A.phase[b] = 1;
  if (A.seg_end) {  // no segment bookkeeping survives a solve
    A.seg_end[b] = kSegNoEnd;
    A.seg_next[b] = -1;
    A.seg_flag[b] = 0;
    A.seg_streak[b] = 0;
  }
  const unsigned Bp = A.Bp;
  const int N = A.N;
  T x[R::nP], u[R::mP], un[R::mP], xn[n];
  load_rec<T, R::nP>(A.x0 + (size_t)b * R::nP, x);
  if (N > 0) load_rec<T, R::mP>(RECP(A.U, 0, R::mP), u);
  T h = step_of(A, pd, 0);
  float t = time_of(A, 0);
  int md = model_of(A, 0);
  for (int k = 0; k < N; ++k) {
    store_rec<T, R::nP>(RECP(A.X, k, R::nP), x);
    const int kn = k + 1 < N ? k + 1 : k;
    load_rec<T, R::mP>(RECP(A.U, kn, R::mP), un);
    const T hn = step_of(A, pd, kn);
    const float tn = time_of(A, kn);
    const int mdn = model_of(A, kn);
    discrete_step<T, M>(x, u, h, xn, t, md);
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = xn[i];
#pragma unroll
    for (int i = 0; i < R::mP; ++i) u[i] = un[i];
    h = hn;
    t = tn;
    md = mdn;
  }
  store_rec<T, R::nP>(RECP(A.X, N, R::nP), x);
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
                                                         double* viol, double* pen) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  T v, p;
  rows_viol_pen(A, pd, b, &v, &p);
  if (viol) viol[b] = (double)v;
  if (pen) pen[b] = (double)p;
}

// iLQR::UpdateConvergenceStatistics + IsDone (ilqr.hpp:568-619) for one instance.
// gsum = sum_k max_i |d_k,i| / (|u_k,i| + 1) with the POST-forward-pass controls (quirk Q12).
// Returns true when the inner solve is finished.
template <class T>
ALTRO_DEV bool conv_stats_and_done(const DevArrays<T>& A, const DevOpts& o, int b, double gsum, double viol) {
  const double grad = A.N > 0 ? gsum / (double)A.N : 0.0;
  const int it = A.it_inner[b];
  const double dJ = (it == 0) ? A.initial_cost[b] - A.cost_cur[b] : A.cost_prev[b] - A.cost_cur[b];
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
  if (dJ < o.cost_tolerance && grad < o.gradient_tolerance) {
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

// AL outer-loop decision after an inner solve finished and the duals were updated:
// UpdateConvergenceStatistics + IsDone (al_solver.hpp:357-401).  viol / pen are the max violation
// of the stored c_ and the max penalty.  Returns true if the instance keeps iterating (the caller
// then applies UpdatePenalties and starts the next inner solve).
template <class T>
ALTRO_DEV bool al_outer_decide(const DevArrays<T>& A, const DevOpts& o, int b, double viol, double pen) {
  const int outer = A.it_outer[b] + 1;
  A.it_outer[b] = outer;
  A.viol[b] = viol;
  A.penmax[b] = pen;
  const int st = A.status[b];
  int sal = -1;
  if (st != ALTRO_SOLVED)
    sal = st;
  else if (viol < o.constraint_tolerance)
    sal = ALTRO_SOLVED;
  else if (pen > o.maximum_penalty)
    sal = ALTRO_MAX_PENALTY;
  else if (outer >= o.max_iterations_outer)
    sal = ALTRO_MAX_OUTER_ITERATIONS;
  else if (A.it_total[b] >= o.max_iterations_total)
    sal = ALTRO_MAX_ITERATIONS;
  if (sal >= 0) {
    A.status_al[b] = sal;
    return false;
  }
  return true;
}

// step-level iLQR::UpdateConvergenceStatistics
template <class T, class M>
__global__ __launch_bounds__(kBlock) void k_conv_stats(DevArrays<T> A, const ProblemDesc* __restrict__ pd, DevOpts o) {
  constexpr int n = M::n, m = M::m;
  using R = Rec<T, n, m>;
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= A.B) return;
  const unsigned Bp = A.Bp;
  double gsum = 0.0;
  for (int k = 0; k < A.N; ++k) {
    T u[R::mP], kd[R::KP];
    load_rec<T, R::mP>(RECP(A.U, k, R::mP), u);
    using RS = rec_scalar_t<T, M>;
    using RR = Rec<RS, n, m>;
    load_rec_as<T, RS, R::KP, RR::KP, m * n + m>(RECP((const RS*)A.KD, k, RR::KP), kd);
    T mx = T(0);
#pragma unroll
    for (int i = 0; i < m; ++i) mx = max_(mx, abs_(kd[R::oD + i]) / (abs_(u[i]) + T(1)));
    gsum += (double)mx;
  }
  T v, p;
  rows_viol_pen(A, pd, b, &v, &p);
  const int st = A.status[b];
  conv_stats_and_done(A, o, b, gsum, (double)v);
  A.status[b] = st;  // the status change belongs to IsDone, which the step-level API does not call
}

// Per-instance scalars that the state machine of the forward pass reads: fetched at kernel start so the
// ~1 us of memory latency hides behind the rollout instead of sitting at the end of the kernel.
struct InstPre {
  int it_inner, it_total;
  double initial_cost, cost_cur, cost_prev, rho_reg, drho;
};
template <class T>
ALTRO_DEV InstPre load_inst_pre(const DevArrays<T>& A, int b) {
  InstPre p;
  p.it_inner = A.it_inner[b];
  p.it_total = A.it_total[b];
  p.initial_cost = A.initial_cost[b];
  p.cost_cur = A.cost_cur[b];
  p.cost_prev = A.cost_prev[b];
  p.rho_reg = A.rho_reg[b];
  p.drho = A.drho[b];
  return p;
}
// conv_stats_and_done on pre-fetched scalars; cost_cur / status are the values this forward pass set
template <class T>
ALTRO_DEV bool conv_stats_and_done_pre(const DevArrays<T>& A, const DevOpts& o, int b, double gsum, double viol,
                                       const InstPre& pre, double cost_cur, int status) {
  const double grad = A.N > 0 ? gsum / (double)A.N : 0.0;
  const int it = pre.it_inner;
  const double dJ = (it == 0) ? pre.initial_cost - cost_cur : pre.cost_prev - cost_cur;
  A.it_inner[b] = it + 1;
  const int itot = pre.it_total + 1;
  A.it_total[b] = itot;
  A.dJ[b] = dJ;
  A.viol[b] = viol;
  A.grad[b] = grad;
  if (A.hist) {
    A.status[b] = status;  // hist_push snapshots the stored row
    hist_push(A, b);
  }
  A.cost_prev[b] = cost_cur;  // NewIteration copies the row (solver_stats.cpp:54-66)
  bool done = false;
  if (dJ < o.cost_tolerance && grad < o.gradient_tolerance) {
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

// Phase 3 of the forward pass: per-instance state machine (shared by both forward kernels).
// Lane 0 of the instance takes the decisions; the row sweeps of the AL transition (dual and penalty
// updates) are spread over its 20 lanes.  sKD / sU: LDS copies of the gains / controls (or nullptr).
// SPLIT: this caller may split rejection streaks into segments (the batched forward kernel; the persistent kernel only
// verifies / retires / cancels the segments it is handed -- with the split compiled in, its register allocation falls back
// to 560 B of scratch per lane and every iteration of it gets ~10 % slower)
// SEGCODE: the bookkeeping of the segments is compiled in at all (the persistent kernel has a variant without: the one every
// solve that never split runs, at the register allocation it had before the segments existed)
template <class T, class M, bool SPLIT = false, bool SEGCODE = true>
ALTRO_DEV void forward_phase3(const DevArrays<T>& A, const ProblemDesc* pd, const DevOpts& o, int mode, int b, int grp,
                              int t, bool accepted, double alpha_sel, double J_sel, double z_sel, double g_sel,
                              int last_status, double viol, const T* sKD, const T* sU, const InstPre& pre,
                              int* active_out = nullptr, T* sLamW = nullptr, T* sPenW = nullptr,
                              double* ff = nullptr, int kd_stride = Rec<T, M::n, M::m>::KP,
                              int kd_off = Rec<T, M::n, M::m>::oD, const int* eahead_words = nullptr, int eahead_waves = 0,
                              int eahead_tag = 0) {
  constexpr int m = M::m;
  constexpr int LS = kLineSearchLanes;
  using R = Rec<T, M::n, M::m>;
  const unsigned Bp = A.Bp;
  const int N = A.N;
  // ---- phase 3: per-instance state machine.  Lane 0 of the instance takes the decisions; the
  //      row sweeps of the AL transition (dual and penalty updates) are spread over its 20 lanes.
  // rejected step: the controls are unchanged and quirk Q12 evaluates the gradient measure with the
  // current Z_ (ilqr.hpp:574-583).  Each lane sums a contiguous block of knots in order, the blocks
  // are then added in lane order: the same sum up to the association of the partial sums.
  double gsum_rej = 0.0;
  if (!accepted && mode != kFwdStepOnly) {
    const int per = (N + LS - 1) / LS;
    double part = 0.0;
    for (int j = 0; j < per; ++j) {
      const int k = t * per + j;
      if (k < N) {
        T mx = T(0);
#pragma unroll
        for (int i = 0; i < m; ++i) {
          using RS = rec_scalar_t<T, M>;
          const T dv = sKD ? sKD[k * kd_stride + kd_off + i]
                           : (T)RECP((const RS*)A.KD, k, (Rec<RS, M::n, M::m>::KP))[R::oD + i];
          const T uv = sU ? sU[k * R::mP + i] : RECP(A.U, k, R::mP)[i];
          mx = max_(mx, abs_(dv) / (abs_(uv) + T(1)));
        }
        part += (double)mx;
      }
    }
    if (active_out) {  // the persistent kernel: the instance sits in lanes 0 .. 19 of its wave, the lane index is a constant
#pragma unroll
      for (int j = 0; j < LS; ++j)
        gsum_rej += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(part), j), __builtin_amdgcn_readlane(__double2loint(part), j));
    } else {
      for (int j = 0; j < LS; ++j) gsum_rej += __shfl(part, grp * LS + j);
    }
  }
  int inner_done = 0;
  double rho_next = 0.0, drho_next = 0.0;  // (lane 0) the regularisation entering the next iteration
  if (t == 0) {
    double cost_cur = pre.cost_cur;
    {
      // (unconditional stores: sibling branches that each store one double get merged into a store
      // through a pointer table in scratch memory)
      double rho = pre.rho_reg, drho = pre.drho;
      if (!accepted) increase_reg(o, &rho, &drho);  // ilqr.hpp:550
      A.rho_reg[b] = rho;
      A.drho[b] = drho;
      rho_next = rho;
      drho_next = drho;
      if (ff) {  // what the stall detector of the persistent kernel compares between iterations
        ff[0] = accepted ? 0.0 : 1.0;
        ff[1] = rho;
        ff[2] = drho;
      }
    }
    if (accepted) {
      A.cost_cur[b] = J_sel;  // stats_.Log("cost"/"alpha"/"z")
      A.alpha[b] = alpha_sel;
      A.z[b] = z_sel;
      cost_cur = J_sel;
    }
    if (mode == kFwdStepOnly) {
      A.status[b] = last_status;
      A.viol[b] = viol;
    } else {
      const double gsum = accepted ? g_sel : gsum_rej;
      inner_done = conv_stats_and_done_pre(A, o, b, gsum, viol, pre, cost_cur, last_status) ? 1 : 0;
      if (ff) {  // (persistent kernel: the counters this iteration leaves -- what a twin workgroup's hand-over compares)
        ff[6] = (double)(pre.it_inner + 1);
        ff[7] = (double)(pre.it_total + 1);
      }
    }
  }
  if (mode == kFwdStepOnly) return;
  inner_done = __shfl(inner_done, grp * LS);
  // ---- segments of a rejection streak (DevArrays::seg_*): verify / retire, cancel, split ----
  if (SEGCODE && A.seg_end) {
    int drop = 0, nclones = 0, first = 0, seg_len = 0;
    if (t == 0) {
      const bool rc = !accepted && !inner_done;  // every trial rejected, and the inner solve goes on
      const int streak = rc ? A.seg_streak[b] + 1 : 0;
      A.seg_streak[b] = streak;
      const int it_next = pre.it_inner + 1, tot_next = pre.it_total + 1;
      const int flag = A.seg_flag[b];
      const int nxt = A.seg_next[b];
      if (flag & kSegCancelled) {
        drop = 1;  // a predecessor did not arrive where this column assumed it would: its work is void
      } else if (nxt >= 0) {
        bool cancel = !rc;
        if (rc && it_next == A.seg_end[b]) {
          // the end of this column's segment: does it hold, bit for bit, what the next column assumed when it started?
          const bool same = A.seg_tot0[nxt] == tot_next && __double_as_longlong(A.seg_rho0[nxt]) == __double_as_longlong(rho_next) &&
                            __double_as_longlong(A.seg_drho0[nxt]) == __double_as_longlong(drho_next) &&
                            (__hip_atomic_load(A.seg_flag + nxt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kSegCancelled) == 0;
          if (same) {
            A.seg_flag[b] = flag | kSegRetired;  // (the next column owns the instance from here: k_seg_fixup follows the chain)
            drop = 1;
          } else {
            cancel = true;
          }
        }
        if (cancel) {  // an accepted step, an end of the inner solve, a regularisation off the rule: the successors are void
          int sgd = nxt;
          for (int guard = 0; sgd >= 0 && guard < 256; ++guard) {
            atomicOr(A.seg_flag + sgd, kSegCancelled);
            sgd = A.seg_next[sgd];
          }
          A.seg_next[b] = -1;
          A.seg_end[b] = kSegNoEnd;
        }
      } else if (SPLIT && rc && streak >= 2 && !active_out && A.next_count && A.seg_parts > 1) {
        // ---- split: what is left of this inner solve (ilqr.hpp:600-611 caps it) in seg_parts segments ----
        const int r1 = o.max_iterations_inner - it_next, r2 = o.max_iterations_total - tot_next;
        const int Rl = r1 < r2 ? r1 : r2;
        if (Rl >= kSegMinRemaining) {
          const int want = A.seg_parts - 1;
          const int base = atomicAdd(A.seg_cursor, want);
          if (base + want <= A.seg_hi - A.seg_lo) {
            nclones = want;
            first = A.seg_lo + base;
            seg_len = (Rl + want) / (want + 1);
            A.seg_end[b] = it_next + seg_len;
            A.seg_next[b] = first;
            double r = rho_next, d = drho_next;
            for (int j = 1; j <= want; ++j) {
              // the regularisation entering the segment: every iteration before it runs its backward pass
              // (DecreaseRegularization, ilqr.hpp:440) and rejects its line search (IncreaseRegularization, :550)
              for (int q = 0; q < seg_len; ++q) {
                decrease_reg(o, &r, &d);
                increase_reg(o, &r, &d);
              }
              const int col = first + j - 1;
              A.seg_rho0[col] = r;
              A.seg_drho0[col] = d;
              A.seg_tot0[col] = tot_next + j * seg_len;
              A.seg_next[col] = j < want ? col + 1 : -1;
              A.seg_end[col] = j < want ? it_next + (j + 1) * seg_len : kSegNoEnd;
              A.seg_flag[col] = 0;
              A.seg_streak[col] = streak;
              // per-instance solver state of the clone: this column's, with counters and regularisation advanced; the
              // previous cost is the (unchanged) current one, as every iteration of a streak leaves it
              A.rho_reg[col] = r;
              A.drho[col] = d;
              A.it_inner[col] = it_next + j * seg_len;
              A.it_total[col] = tot_next + j * seg_len;
              const double cc = accepted ? J_sel : pre.cost_cur;
              A.cost_cur[col] = cc;
              A.cost_prev[col] = cc;
              A.initial_cost[col] = pre.initial_cost;
              A.dV0[col] = 0.0; A.dV1[col] = 0.0; A.J0[col] = 0.0;
              A.dJ[col] = A.dJ[b]; A.grad[col] = A.grad[b]; A.viol[col] = A.viol[b]; A.penmax[col] = A.penmax[b];
              A.alpha[col] = A.alpha[b]; A.z[col] = A.z[b]; A.reg_log[col] = A.reg_log[b];
              A.status[col] = ALTRO_UNSOLVED;
              A.status_al[col] = A.status_al[b];
              A.it_outer[col] = A.it_outer[b];
              A.need_init_cost[col] = 0;
              A.phase[col] = 1;
            }
            const int at = atomicAdd(A.next_count, want);  // the clones iterate from the next sweep on
            for (int j = 0; j < want; ++j) A.next_list[at + j] = first + j;
          } else {
            atomicSub(A.seg_cursor, want);  // (no columns left in this chain's slice)
          }
        }
      }
    }
    drop = __shfl(drop, grp * LS);
    nclones = __shfl(nclones, grp * LS);
    first = __shfl(first, grp * LS);
    if (SPLIT && nclones > 0) {
      // the clones' share of the instance: trajectory, multipliers, penalties, parameters -- nothing a rejected iteration
      // changes (the stored constraint values, knot costs, records and gains are recomputed before they are read)
      using R_ = Rec<T, M::n, M::m>;
      for (int j = 0; j < nclones; ++j) {
        const unsigned col = (unsigned)(first + j);
        for (int i = t; i < (N + 1) * R_::nP; i += LS) {
          const int k = i / R_::nP, e = i - k * R_::nP;
          A.X[((size_t)(unsigned)k * Bp + col) * R_::nP + e] = A.X[((size_t)(unsigned)k * Bp + (unsigned)b) * R_::nP + e];
        }
        for (int i = t; i < N * R_::mP; i += LS) {
          const int k = i / R_::mP, e = i - k * R_::mP;
          A.U[((size_t)(unsigned)k * Bp + col) * R_::mP + e] = A.U[((size_t)(unsigned)k * Bp + (unsigned)b) * R_::mP + e];
        }
        if (t < R_::nP) A.x0[(size_t)col * R_::nP + t] = A.x0[(size_t)(unsigned)b * R_::nP + t];
        for (int r = t; r < pd->total_rows; r += LS) {
          A.lam[(unsigned)r * Bp + col] = A.lam[(unsigned)r * Bp + (unsigned)b];
          A.pen[(unsigned)r * Bp + col] = A.pen[(unsigned)r * Bp + (unsigned)b];
          A.cval[(unsigned)r * Bp + col] = A.cval[(unsigned)r * Bp + (unsigned)b];
        }
        T* const ip = const_cast<T*>(A.ipool);
        for (int r = t; r < pd->nslots; r += LS) ip[(unsigned)r * Bp + col] = ip[(unsigned)r * Bp + (unsigned)b];
      }
    }
    if (drop) {  // retired or cancelled: this column leaves the solve here
      if (t == 0) {
        if (ff) ff[3] = 1.0;
        if (active_out) *active_out = 0;
        A.phase[b] = 0;
      }
      return;
    }
  }
  bool active = true;
  if (inner_done) {
    if (mode == kFwdAL) {
      // (persistent kernel: the other waves may be computing the next iteration's expansions AHEAD from the LDS copies
      //  of the multipliers that the sweeps below rewrite -- k_sweep_fused, "E AHEAD".  Their result is dropped when
      //  the inner solve ends, but the records they leave in memory must be this iteration's, not a mixture: wait
      //  until they are through.  Bounded like every poll of the kernel.)
      if (eahead_words && !accepted) {
        bool seen = true;
        for (int w = 0; w < eahead_waves; ++w) {
          bool ok = false;
          for (int tries = 0; tries < kFwdSpinLimit && !ok; ++tries)
            ok = __builtin_amdgcn_readfirstlane(__hip_atomic_load(eahead_words + w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >= eahead_tag;
          seen = seen && ok;
        }
        // (gave up: the sweeps below would rewrite multipliers that the expansion waves still read -- the launch reports
        //  it like every other poll that times out, FwdSync::wait_for)
        if (!seen) __hip_atomic_store(const_cast<int*>(eahead_words) + kEAheadToErrWord, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      // AugmentedLagrangianiLQR: UpdateDuals, UpdateConvergenceStatistics, IsDone, UpdatePenalties
      // (al_solver.hpp:313-401); each lane sweeps the rows of knots t, t+20, ...
      T vpart = T(0), ppart = T(0);
      for (int k = t; k <= N; k += LS) {
        int rb;
        const KnotClass& kc = class_of_knot(A, pd, k, &rb);
        for (int ci = 0; ci < kc.ncon; ++ci) {
          const ConDesc& cd = kc.con[ci];
          for (int i = 0; i < cd.p; ++i) {
            const unsigned idx = (unsigned)(rb + cd.row_off + i) * Bp + (unsigned)b;
            const T c = A.cval[idx], rho = A.pen[idx];
            const T lnew = dual_proj(cd.type, A.lam[idx] - rho * c);  // constraint_values.hpp:192-194
            A.lam[idx] = lnew;
            if (sLamW) sLamW[rb + cd.row_off + i] = lnew;  // LDS-resident copy (persistent kernel)
            vpart = max_(vpart, violation(cd.type, c));
            ppart = max_(ppart, rho);
          }
        }
      }
      T vm = vpart, pm = ppart;
      for (int j = 0; j < LS; ++j) {
        vm = max_(vm, __shfl(vpart, grp * LS + j));
        pm = max_(pm, __shfl(ppart, grp * LS + j));
      }
      int cont = 0;
      if (t == 0) cont = al_outer_decide(A, o, b, (double)vm, (double)pm) ? 1 : 0;
      cont = __shfl(cont, grp * LS);
      if (cont) {
        for (int k = t; k <= N; k += LS) {  // constraint_values.hpp:202-207
          const int cls = A.knot_class[k];
          const KnotClass& kc = pd->cls[cls];
          const int rb = A.knot_rowbase[k];
          for (int ci = 0; ci < kc.ncon; ++ci) {
            const T phi = T(A.phi[cls * kMaxConPerKnot + ci]);
            for (int i = 0; i < kc.con[ci].p; ++i) {
              const unsigned idx = (unsigned)(rb + kc.con[ci].row_off + i) * Bp + (unsigned)b;
              const T pnew = A.pen[idx] * phi;
              A.pen[idx] = pnew;
              if (sPenW) sPenW[rb + kc.con[ci].row_off + i] = pnew;
            }
          }
        }
        if (t == 0) {
          begin_inner_solve(A, o, b);
          if (ff) {
            ff[5] = 1.0;  // (persistent kernel: LDS mirror of need_init_cost)
            ff[6] = 0.0;  // ... and of it_inner
          }
        }
      }
      active = cont != 0;
    } else {
      active = false;
    }
  }
  if (t != 0) return;
  if (ff) ff[3] = inner_done ? 1.0 : 0.0;
  if (active_out) *active_out = active ? 1 : 0;  // persistent sweep kernel: keep iterating?
  if (!active) {
    A.phase[b] = 0;
  } else if (A.next_count) {
    const int slot = atomicAdd(A.next_count, 1);  // order is arbitrary; instances are independent
    A.next_list[slot] = b;
  }
}

// One run of knots of the closed-loop rollout + cost of ONE line-search trial (one lane):
// iLQR::RolloutClosedLoop + ALCost::Evaluate per knot (ilqr.hpp:468-499, al_cost.hpp:264-274).
// FK picks a compile-time constraint layout (FastKind); kFastGeneric is the table-driven fallback.
// Values and operation order are identical across the variants.
template <class T, class M, class Ctx, bool LDS, int FK>
ALTRO_DEV void rollout_run(const Ctx& C, const ProblemDesc* pd, const DevArrays<T>& A, const KnotRun& run, int kend,
                           const T* sX, const T* sU, const T* sKD, T alpha, T hh, bool valid,
                           unsigned tb, unsigned Bp, int b, bool check_bounds, T state_max2, T control_max2,
                           T* xb, double& J, double& gs, bool& ok, int& st) {
  constexpr int n = M::n, m = M::m, nm = n + m;
  constexpr int LS = kLineSearchLanes;
  const KnotClass& kc = pd->cls[run.cls];
  RunConsts<T, n, m> RC;
  load_run_consts<T, n, m>(C, pd, kc, RC);
  // layout of the specialised variants: circle rows / bound rows inside the knot
  constexpr bool kHasB = (FK == kFastB || FK == kFastCB || FK == kFastBC);
  constexpr bool kHasC = (FK == kFastC || FK == kFastCB || FK == kFastBC);
  constexpr int kCi = (FK == kFastBC) ? 1 : 0;  // constraint index of the circle
  constexpr int kBi = (FK == kFastCB) ? 1 : 0;  // constraint index of the bound
  int c_p = 0, c_pi = 0, c_off = 0, c_row = 0, b_row = 0;
  if (kHasC) {
    c_p = kc.con[kCi].p;
    c_pi = kc.con[kCi].per_instance;
    c_off = kc.con[kCi].param_off;
    c_row = kc.con[kCi].row_off;
  }
  if (kHasB) b_row = kc.con[kBi].row_off;
  const int nrows = kc.nrows;
  for (int k = run.k_begin; k < kend; ++k) {
    using R = Rec<T, n, m>;
    T xk[R::nP], uk[R::mP], kd[R::KP], ub[m], xn[n];
    if (LDS) {
      load_rec<T, R::nP>(sX + k * R::nP, xk);
      load_rec<T, R::mP>(sU + k * R::mP, uk);
      load_rec<T, R::KP>(sKD + k * R::KP, kd);
    } else {
      using RS = rec_scalar_t<T, M>;
      using RR = Rec<RS, n, m>;
      load_rec<T, R::nP>(RECP(A.X, k, R::nP), xk);
      load_rec<T, R::mP>(RECP(A.U, k, R::mP), uk);
      load_rec_as<T, RS, R::KP, RR::KP, m * n + m>(RECP((const RS*)A.KD, k, RR::KP), kd);
    }
    const T* K = kd + R::oK;
    const T* d = kd + R::oD;
    const int rb = run.rowbase + (k - run.k_begin) * nrows;
    // duals / penalty of the bound rows: loaded up front so that one wait covers the whole knot
    T blam[2 * m], brho = T(1);
    if (kHasB) {
      brho = C.pen(rb + b_row);
#pragma unroll
      for (int j = 0; j < 2 * m; ++j) blam[j] = C.lam(rb + b_row + j);
    }
    if (ok) {
      // grad term max_i |d_i| / (|u_i| + 1): pick the maximiser by cross-multiplication, divide once
      T gnum = T(0), gden = T(1);
#pragma unroll
      for (int i = 0; i < m; ++i) {
        T s = T(0);
#pragma unroll
        for (int l = 0; l < n; ++l) s += K[i + l * m] * (xb[l] - xk[l]);
        ub[i] = uk[i] + s + d[i] * alpha;
        const T num = abs_(d[i]), den = abs_(ub[i]) + T(1);
        if (num * gden > gnum * den) {
          gnum = num;
          gden = den;
        }
      }
      gs += (double)(gnum / gden);
      if (FK == kFastGeneric) {
        J += (double)knot_cost_fast<T, n, m>(C, pd, kc, RC, rb, xb, ub);
      } else {
        // quadratic cost (diagonal Q, R guaranteed by the host for the fast kinds)
        T Jk = T(0);
      {
      T xQx = T(0), uRu = T(0), qx = T(0), ru = T(0);
#pragma unroll
        for (int i = 0; i < n; ++i) {
          xQx += xb[i] * (RC.Qd[i] * xb[i]);
          qx += RC.q[i] * xb[i];
        }
#pragma unroll
        for (int i = 0; i < m; ++i) {
          uRu += ub[i] * (RC.Rd[i] * ub[i]);
          ru += RC.r[i] * ub[i];
        }
        Jk = T(0.5) * xQx + T(0.5) * uRu + qx + ru + RC.c;
        auto circle_term = [&]() {
          const T rho = C.pen(rb + c_row);
          T a = T(0), bsum = T(0);
          for (int i = 0; i < c_p; ++i) {
            T dx = xb[0] - C.par(c_pi, c_off, 3 * i);
            T dy = xb[1] - C.par(c_pi, c_off, 3 * i + 1);
            T rr = C.par(c_pi, c_off, 3 * i + 2);
            T c = circle_value(dx, dy, rr);
            T lam = C.lam(rb + c_row + i);
            T lp = dual_proj(1, lam - rho * c);
            a += lp * lp;
            bsum += lam * lam;
          }
          T Jc = a - bsum;
          Jk += Jc / (2 * rho);
        };
        auto bound_term = [&]() {
          T a = T(0), bsum = T(0);
#pragma unroll
          for (int j = 0; j < m; ++j) {
            T c = RC.bnd[j] - ub[j];
            T lp = dual_proj(1, blam[j] - brho * c);
            a += lp * lp;
            bsum += blam[j] * blam[j];
          }
#pragma unroll
          for (int j = 0; j < m; ++j) {
            T c = ub[j] - RC.bnd[m + j];
            T lp = dual_proj(1, blam[m + j] - brho * c);
            a += lp * lp;
            bsum += blam[m + j] * blam[m + j];
          }
          T Jc = a - bsum;
          Jk += Jc / (2 * brho);
        };
        if (FK == kFastB) bound_term();
        if (FK == kFastC) circle_term();
        if (FK == kFastCB) {
          circle_term();
          bound_term();
        }
        if (FK == kFastBC) {
          bound_term();
          circle_term();
        }
        }
      J += (double)Jk;
      }
      if (valid) {  // idle lanes must not touch instance 0's candidates
        T* cand = A.trial + (tb + (unsigned)k * (unsigned)(LS * nm));
#pragma unroll
        for (int i = 0; i < n; ++i) cand[i] = xb[i];
#pragma unroll
        for (int i = 0; i < m; ++i) cand[n + i] = ub[i];
      }
      discrete_step<T, M>(xb, ub, A.hk ? T(A.hk[k]) : hh, xn, time_of(A, k), model_of(A, k));  // (per-knot steps / times: this kernel only)
      if (check_bounds) {
        // ||x||_2 > state_max  <=>  ||x||^2 > state_max^2 (ilqr.hpp:484-495), no sqrt needed
        T sx = T(0), su = T(0);
#pragma unroll
        for (int i = 0; i < n; ++i) sx += xn[i] * xn[i];
#pragma unroll
        for (int i = 0; i < m; ++i) su += ub[i] * ub[i];
        if (sx > state_max2) {
          ok = false;
          st = ALTRO_STATE_LIMIT;
        } else if (su > control_max2) {
          ok = false;
          st = ALTRO_CONTROL_LIMIT;
        }
      }
#pragma unroll
      for (int i = 0; i < n; ++i) xb[i] = xn[i];
    }
  }
}

// DISTANCE BETWEEN THE STAGED BLOCKS OF A WORKGROUP'S INSTANCES (round 6).  The lanes of a wave read the
// staged block 20 per address, three addresses per wave (one per instance), the blocks `total()` elements apart.  The LDS has
// 64 banks of 4 bytes (256 B), and what SQ_LDS_BANK_CONFLICT counts for that pattern depends on the distance modulo 256 B
// (scripts/probes/lds_conflict_probe.hip, profiles/r06_lds_conflict_probe.txt): 32 B -- the unicycle's 17 696 B block with
// fp64 records, config 2 -- 1.86 / 2.78 conflict cycles per b64 / b128 read, 0 B 2.78 / 4.64, 128 B 0.93, but 64, 96 and
// 160 B none; config 3 stages nothing but the multipliers (kSrcGlb) and counts none.  The block is padded to
// ALTRO_FWD_BLOCK_MOD = 160 B modulo 256 (the engine sizes the allocation with the same function; -1: no padding, rounds 1 - 5).
// What it buys is LDS issue slots, not time -- a lone wave reads at the same pace with and without the counted conflicts, and
// the A/B on the real kernel is neutral (6.09 - 6.19 against 6.15 - 6.20 ms per step; 96 B, equally conflict-free in the
// probe, is 5 % SLOWER: profiles/r06_experiments.txt #5).
#ifndef ALTRO_FWD_BLOCK_MOD
#define ALTRO_FWD_BLOCK_MOD 160
#endif
__host__ __device__ constexpr int fwd_block_pad_bytes(long long raw_bytes) {
  return ALTRO_FWD_BLOCK_MOD < 0 ? 0 : (int)(((ALTRO_FWD_BLOCK_MOD - raw_bytes % 256) + 256) % 256);
}
template <class T>
struct FwdLds {  // element counts of one instance's staged block (16-byte aligned sub-blocks)
  int nX, nU, nKD, nR, nS, V;
  ALTRO_DEV int padv(int e) const { return (e + V - 1) / V * V; }
  ALTRO_DEV int rowsP() const { return padv(nR); }
  ALTRO_DEV int raw() const { return nX + nU + nKD + 2 * padv(nR) + padv(nS); }
  ALTRO_DEV int total() const { return raw() + fwd_block_pad_bytes((long long)raw() * (long long)sizeof(T)) / (int)sizeof(T); }
};

// Single-wave forward pass that reads everything from global memory: the fallback when the staged block of
// one instance exceeds the LDS, or line_search_max_iterations > 20 (rounds of 20 trials).
template <class T, class M>
__global__ __launch_bounds__(kBlock) void k_forward(DevArrays<T> A, const ProblemDesc* __restrict__ pd, DevOpts o,
                                                    int mode, int all, int per_wave) {
  using Ctx = CtxG<T>;
  constexpr bool LDS = false;
  constexpr int n = M::n, m = M::m, nm = n + m;
  constexpr int LS = kLineSearchLanes;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x;
  const int grp = lane / LS;
  const int t = lane - grp * LS;
  const int b0 = (grp < per_wave) ? instance_of_slot(A, blockIdx.x * per_wave + grp, all) : -1;
  const unsigned Bp = A.Bp;
  const int N = A.N;
  const bool valid = b0 >= 0;
  if (__ballot(valid) == 0ull) return;  // nothing to do for this wave (finished instances)
  const int b = valid ? b0 : 0;         // idle lanes shadow instance 0's loads but never store

  using R = Rec<T, n, m>;
  const T *sX = nullptr, *sU = nullptr, *sKD = nullptr;  // this variant reads from global memory
  const CtxG<T> C(A, b);
  const T hh = T(pd->hstep);

  const double J0 = A.J0[b];
  const double dV0 = A.dV0[b], dV1 = A.dV1[b];
  T x0[R::nP];
  load_rec<T, R::nP>(A.x0 + (size_t)b * R::nP, x0);
  const T state_max2 = T(o.state_max) * T(o.state_max);
  const T control_max2 = T(o.control_max) * T(o.control_max);

  const int ls_max = o.line_search_max_iterations;
  bool accepted = false;
  T alpha_sel = T(0);
  double J_sel = J0, z_sel = -1.0, g_sel = 0.0;
  int t_replay = -1;  // trial whose candidate defines c_ (and Z_ when accepted)
  int last_status = ALTRO_UNSOLVED;

  T alpha_base = T(1);
  for (int base = 0; base < ls_max && !accepted; base += LS) {
    // this lane's step length: alpha /= decrease_factor, t times (ilqr.hpp:544)
    T alpha = alpha_base;
    for (int i = 0; i < t; ++i) alpha /= T(o.line_search_decrease_factor);
    const bool live = valid && (base + t < ls_max);
    // ---- phase 1: closed-loop rollout + cost for this lane's alpha (ilqr.hpp:468-499, 527) -------
    bool ok = true;
    int st = ALTRO_UNSOLVED;
    double J = 0.0, gs = 0.0;
    T xb[n];
#pragma unroll
    for (int i = 0; i < n; ++i) xb[i] = x0[i];
    // candidate scratch, instance-major [b][k][trial][x|u]: the 20 trials of an instance write one
    // contiguous 20*(n+m)-element block per knot
    const unsigned tb = ((unsigned)b * (unsigned)(N + 1) * (unsigned)LS + (unsigned)t) * (unsigned)nm;
    for (int r = 0; r < pd->nruns; ++r) {
      const KnotRun run = pd->runs[r];
      const int kend = run.k_end < N ? run.k_end : N;
#define ALTRO_RUN(FK)                                                                                     \
  rollout_run<T, M, Ctx, LDS, FK>(C, pd, A, run, kend, sX, sU, sKD, alpha, hh, valid, tb, Bp, b,          \
                                  o.check_forwardpass_bounds != 0, state_max2, control_max2, xb, J, gs, ok, st)
      switch (run.fast) {
        case kFastNone: ALTRO_RUN(kFastNone); break;
        case kFastB: ALTRO_RUN(kFastB); break;
        case kFastCB: ALTRO_RUN(kFastCB); break;
        case kFastBC: ALTRO_RUN(kFastBC); break;
        case kFastC: ALTRO_RUN(kFastC); break;
        default: ALTRO_RUN(kFastGeneric); break;
      }
#undef ALTRO_RUN
    }
    if (ok) {
      T uz[m];
#pragma unroll
      for (int i = 0; i < m; ++i) uz[i] = T(0);
      const KnotRun runN = pd->runs[pd->nruns - 1];  // the terminal knot closes the last run
      const KnotClass& kcN = pd->cls[runN.cls];
      J += (double)knot_cost<T, n, m, false>(C, pd, kcN, runN.rowbase + (N - runN.k_begin) * kcN.nrows, xb, uz, nullptr);
      if (valid) {
        T* cand = A.trial + (tb + (unsigned)N * (unsigned)(LS * nm));
#pragma unroll
        for (int i = 0; i < n; ++i) cand[i] = xb[i];
      }
    }
    // ---- acceptance test (ilqr.hpp:528-542) --------------------------------------------------
    const double expected = -(double)alpha * (dV0 + (double)alpha * dV1);
    const double z = (expected > 0.0) ? (J0 - J) / expected : -1.0;
    const bool acc = live && ok && o.line_search_lower_bound <= z && z <= o.line_search_upper_bound && J < J0;
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
      t_replay = tsel;
    } else {
      // no acceptance in this round: the serial loop ran all `nlive` trials; c_ now holds the
      // constraint values of the last trial whose rollout succeeded (quirk Q6), and status_ is the
      // outcome of the very last rollout.
      last_status = __shfl(st, grp * LS + (nlive - 1));
      if (ok_g) t_replay = 31 - __clz(ok_g);
    }
    alpha_base = __shfl(alpha, grp * LS + (LS - 1)) / T(o.line_search_decrease_factor);
    if (!accepted && base + LS < ls_max && t_replay >= 0) {
      // (only reachable with line_search_max_iterations > 20) the next round overwrites the
      // candidates: c_ then falls back to the values of the expansion step.
      t_replay = -1;
    }
  }
  if (!valid) return;
  __threadfence_block();  // candidates written by the other lanes of this wave are read below

  // ---- phase 2: all 20 lanes of the instance, knots strided over lanes -----------------------
  //      accepted: (*Z_) = (*Zbar_) (ilqr.hpp:548); always: c_ of the last evaluated candidate.
  T viol = T(0);
  if (t_replay >= 0) {
    const unsigned rbo = ((unsigned)b * (unsigned)(N + 1) * (unsigned)LS + (unsigned)t_replay) * (unsigned)nm;
    for (int k = t; k <= N; k += LS) {
      T xs[n], us[m];
      const T* cand = A.trial + (rbo + (unsigned)k * (unsigned)(LS * nm));
#pragma unroll
      for (int i = 0; i < n; ++i) xs[i] = cand[i];
#pragma unroll
      for (int i = 0; i < m; ++i) us[i] = (k < N) ? cand[n + i] : T(0);
      int rb;
      const KnotClass& kc = class_of_knot(A, pd, k, &rb);
      T v;
      knot_cost<T, n, m, true>(C, pd, kc, rb, xs, us, &v);
      viol = max_(viol, v);
      if (accepted) {
        T xr[R::nP], ur[R::mP];
#pragma unroll
        for (int i = 0; i < R::nP; ++i) xr[i] = i < n ? xs[i < n ? i : 0] : T(0);
#pragma unroll
        for (int i = 0; i < R::mP; ++i) ur[i] = i < m ? us[i < m ? i : 0] : T(0);
        store_rec<T, R::nP>(RECP(A.X, k, R::nP), xr);
        if (k < N) store_rec<T, R::mP>(RECP(A.U, k, R::mP), ur);
      }
    }
  }
  // max over the instance's lanes
  {
    T vm = viol;
    for (int j = 0; j < LS; ++j) vm = max_(vm, __shfl(viol, grp * LS + j));
    viol = vm;
  }
  if (t_replay < 0) {
    // c_ untouched since the expansion step: recompute the max violation of the stored values
    T vpart = T(0);
    for (int k = t; k <= N; k += LS) {
      int rb;
      const KnotClass& kc = class_of_knot(A, pd, k, &rb);
      for (int ci = 0; ci < kc.ncon; ++ci)
        for (int i = 0; i < kc.con[ci].p; ++i)
          vpart = max_(vpart, violation(kc.con[ci].type, SOA(A.cval, rb + kc.con[ci].row_off + i)));
    }
    T vm = vpart;
    for (int j = 0; j < LS; ++j) vm = max_(vm, __shfl(vpart, grp * LS + j));
    viol = vm;
  }

  {
    const InstPre pre = load_inst_pre(A, b);
    forward_phase3<T, M>(A, pd, o, mode, b, grp, t, accepted, (double)alpha_sel, J_sel, z_sel, g_sel, last_status,
                       (double)viol, (const T*)nullptr, (const T*)nullptr, pre);
  }
}

// -------------------------------------------------------------------------------------------------
// Forward pass, two-wave pipeline (the production variant whenever the staged block fits in LDS).
//
// The dependent chain of a rollout is  x_k -> u_k = u + K dx + alpha d -> RK4 -> x_{k+1}.  The AL
// cost, the gradient statistic and the candidate stores only CONSUME (x_k, u_k).  A single
// wavefront issues one instruction per ~4.4 cycles no matter how few lanes are active, so the step
// is split over the two wavefronts of a 128-thread workgroup (two SIMDs of one CU):
//   wave 0 "rollout": LDS reads of (x,u,K,d)_k, control law, RK4 (3 sincos), bound checks, and a
//                     5-value hand-off of (xbar_k, ubar_k) into a double-buffered LDS slot;
//   wave 1 "cost":    one step behind: AL knot cost (compile-time constraint layout), gradient
//                     statistic, candidate stores; then acceptance ballot, phases 2 and 3.
// One workgroup barrier per knot; slot k&1 is rewritten by wave 0 only after barrier k+1, which wave
// 1 reaches only after it has consumed slot k&1.  Same lane layout in both waves (3 instances x 20
// trials).  Values and operation order are identical to the single-wave kernel.
// -------------------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, which would
// make the cost wave wait for its (scattered, never re-read in the loop) candidate stores at every
// knot -- and the rollout wave with it.
// CANDIDATE SCRATCH OF THE BATCHED FORWARD KERNEL (round 4).  Storing the (xbar, ubar) of all 20 speculative trials of every
// knot -- 81 KB per instance and launch, 331 MB per sweep of 4096 instances -- made k_forward2 the one kernel whose HBM
// traffic was 11 - 23 x its algorithmic bytes, written at 1.5 - 2.5 TB/s while the knot loops run: the
// write bandwidth, not the CUs, bounded the throughput phase (more resident workgroups did not help).  What phase 2 reads
// back is ONE trial: the accepted one -- trial 0 .. 7 in 99.3 % (kTurn90) / 95 % (obstacles) of the accepted steps -- or,
// when every trial was rejected, the LAST live one (quirk Q6: c_ of the last evaluated candidate).  So only the first
// `front` trials and the last live trial get a slot (front + 1 slots per knot, 7 for the default front = 6: 28 KB per
// instance); a deeper winner is REPLAYED by the rollout wave -- the same code on the same inputs, bit-identical -- into the
// shared last slot before phase 2 (a workgroup-uniform, rare branch).  front >= 19: one slot per trial, never a replay.
struct CandLayout {
  int front, last, slots;  // trials [0, front) own slot t; trial `last` (= live trials - 1) and replayed trials use slot `front`
  ALTRO_DEV CandLayout(int front_, int nlive) {
    last = nlive - 1;
    front = front_ < last ? (front_ < 0 ? 0 : front_) : last;  // (front == last: every live trial has its own slot)
    slots = front + 1;
  }
  ALTRO_DEV int store_slot(int t) const { return t < front ? t : (t == last ? front : -1); }   // -1: not stored by the knot loop
  ALTRO_DEV int read_slot(int t) const { return t < front ? t : front; }
  ALTRO_DEV bool needs_replay(int t) const { return t >= front && t != last; }
};

// Phase 2 of the forward pass for the knots k0, k0 + stride, ...: copy the replayed trial (the accepted
// one, or the last whose rollout succeeded: quirk Q6) out of the candidate scratch, evaluate and store
// the constraint values it leaves in c_, and -- if accepted -- install it as the new trajectory.
// Returns the max violation over those knots.  t_replay < 0: c_ is untouched since the expansion step.
// One knot's term of the gradient measure (ilqr.hpp:574-583): max_i |d_i| / (|u_i| + 1).  The maximiser is picked by
// cross-multiplication, then ONE division.
template <class T, int m>
ALTRO_DEV T grad_term(const T* d, const T* u) {
  T gnum = T(0), gden = T(1);
#pragma unroll
  for (int i = 0; i < m; ++i) {
    const T num = abs_(d[i]), den = abs_(u[i]) + T(1);
    if (num * gden > gnum * den) {
      gnum = num;
      gden = den;
    }
  }
  return gnum / gden;
}

// `cslots`: candidate slots per knot of the scratch (its knot stride is cslots * (n + m)); the replayed trial sits in
// slot cand_slot_of(t_replay, ...) -- see CandLayout.
template <class T, class M, class Ctx>
ALTRO_DEV T forward_phase2(const DevArrays<T>& A, const ProblemDesc* pd, const Ctx& C, int b, int t_replay, bool accepted,
                           int k0, int stride, const T* cand_base, unsigned cand_off0, int cslot, int cslots, T* sXw = nullptr,
                           T* sUw = nullptr, T* rk = nullptr, const T* sKD = nullptr, int kd_stride = 0, int kd_off = 0) {
  constexpr int n = M::n, m = M::m, nm = n + m;
  using R = Rec<T, n, m>;
  const unsigned Bp = A.Bp;
  const int N = A.N;
  const unsigned LSnm = (unsigned)cslots * (unsigned)nm;
  T viol = T(0);
  if (t_replay >= 0) {
    const unsigned rbo = cand_off0 + (unsigned)cslot * (unsigned)nm;
    constexpr int kAhead = 3;  // candidates fetched before the first is used
    for (int kb = k0; kb <= N; kb += kAhead * stride) {
      T xs[kAhead][n], us[kAhead][m];
#pragma unroll
      for (int j = 0; j < kAhead; ++j) {
        const int k = kb + j * stride;
        const T* cand = cand_base + (rbo + (unsigned)(k <= N ? k : N) * LSnm);
#pragma unroll
        for (int i = 0; i < n; ++i) xs[j][i] = cand[i];
#pragma unroll
        for (int i = 0; i < m; ++i) us[j][i] = cand[n + i];  // knot N: never written, replaced below
      }
#pragma unroll
      for (int j = 0; j < kAhead; ++j) {
        const int k = kb + j * stride;
        if (k > N) break;
        if (k == N) {
#pragma unroll
          for (int i = 0; i < m; ++i) us[j][i] = T(0);
        }
        int rb;
        const KnotClass& kc = class_of_knot(A, pd, k, &rb);
        T v;
        knot_cost<T, n, m, true>(C, pd, kc, rb, xs[j], us[j], &v);
        viol = max_(viol, v);
        if (accepted) {
          if (rk && k < N) {
            if (sKD) {
              rk[k] = grad_term<T, m>(sKD + k * kd_stride + kd_off, us[j]);
            } else {  // the feedforward term of the gain record in global memory
              using RS = rec_scalar_t<T, M>;
              const RS* kdr = RECP((const RS*)A.KD, k, (Rec<RS, n, m>::KP)) + R::oD;
              T dv[m];
#pragma unroll
              for (int i = 0; i < m; ++i) dv[i] = (T)kdr[i];
              rk[k] = grad_term<T, m>(dv, us[j]);
            }
          }
          T xr[R::nP], ur[R::mP];
#pragma unroll
          for (int i = 0; i < R::nP; ++i) xr[i] = i < n ? xs[j][i < n ? i : 0] : T(0);
#pragma unroll
          for (int i = 0; i < R::mP; ++i) ur[i] = i < m ? us[j][i < m ? i : 0] : T(0);
          store_rec<T, R::nP>(RECP(A.X, k, R::nP), xr);
          if (k < N) store_rec<T, R::mP>(RECP(A.U, k, R::mP), ur);
          if (sXw) {  // LDS-resident copy of the trajectory (persistent kernel)
            store_rec<T, R::nP>(sXw + k * R::nP, xr);
            if (k < N) store_rec<T, R::mP>(sUw + k * R::mP, ur);
          }
        }
      }
    }
  } else {
    for (int k = k0; k <= N; k += stride) {
      int rb;
      const KnotClass& kc = class_of_knot(A, pd, k, &rb);
      for (int ci = 0; ci < kc.ncon; ++ci)
        for (int i = 0; i < kc.con[ci].p; ++i)
          viol = max_(viol, violation(kc.con[ci].type, SOA(A.cval, rb + kc.con[ci].row_off + i)));
    }
  }
  return viol;
}


// The rollout wave hands (xbar_k, ubar_k) to the two consumer waves through a ring of kFwdSlots LDS slots and
// the workgroup barrier is taken once per PAIR of knots: the producer syncs after the odd knots (and after the
// terminal one), the consumers before the even ones.  Barrier j publishes knots 2j, 2j+1; the producer overwrites
// their slots with knots 2j+4, 2j+5 only after barrier j+1, which the consumers reach after finishing pair j.
// G = knots per barrier: 2 in the batched sweeps (kFwdSlots = 4 slots), kSyncFused in the persistent kernel, whose
// knot loop runs in lock step with the speculative backward pass of the fourth wave: the longer the stretch between
// two barriers, the less the slowest of four waves per stretch costs (and the fewer barriers the recursion pays).
#ifndef ALTRO_SYNC_BATCHED
#define ALTRO_SYNC_BATCHED 2  // knots per workgroup barrier in the batched forward kernel (A/B builds: 4)
#endif
constexpr int kSyncBatched = ALTRO_SYNC_BATCHED;
constexpr int kFwdSlots = 2 * kSyncBatched;
// ... per variant (round 6): the large models' global-source variant meets at EVERY knot -- two hand-off slots instead of four.
// With n + m = 16 a slot is 8 KB; the two it saves are what lets a second two-instance workgroup of the 12-state model share
// the CU's LDS (config 4), and one knot of that model is long enough for a barrier of its own.  SRC: FwdSrc (2 = kSrcGlb).
template <class M, int SRC>
constexpr int fwd_sync_batched() {
  return (SRC == 2 && M::n * M::m >= 12) ? 1 : kSyncBatched;
}
ALTRO_DEV bool producer_syncs_after(int k, int N, int G = 2) { return (k & (G - 1)) == G - 1 || k == N; }
ALTRO_DEV bool consumer_syncs_before(int k, int G = 2) { return (k & (G - 1)) == 0; }
ALTRO_DEV int fwd_slot(int k, int G = 2) { return k & (2 * G - 1); }


// ---------------------------------------------------------------------------------------------------------------------
// How the three waves of the forward pass meet.
//
// HARDWARE (SOFT = false: the batched sweeps and the persistent kernel's lock-step modes): LDS-only workgroup barriers --
// one per stretch of G knots in the knot loop (producer after the stretch, consumers before it), then A, S, V.  Every
// wave of the workgroup has to execute every one of them: the fourth wave of k_sweep_fused<.., kSpecWave>, which runs the
// speculative backward pass, therefore walks in LOCK STEP with the knot loop (the recursion at 825 cycles per knot sets
// the pace of a loop whose own waves need 620).
//
// SOFT (k_sweep_fused<.., kSpecFree>): no hardware barrier anywhere in the forward pass.  The rollout wave publishes a
// stretch by bumping a sequence word in LDS behind the stretch's slot writes (LDS operations of one wave execute in
// order; the release makes the compiler keep that order), the consumers poll it; the consumers report the stretches
// they have finished the same way, and the producer -- two stretches ahead at most: the ring has 2 G slots -- looks at
// those words one stretch before it needs them.  A, S and V are sequence words too.  The fourth wave then runs its
// recursion at its own pace (700 cycles per knot) beside a knot loop that runs at ITS own pace, and all four waves meet
// at the workgroup barrier behind the forward pass.  Sequence numbers grow monotonically over the iterations of the
// persistent kernel (base = iteration * kFwdSeqStride), so nothing is ever reset.  Every poll loop is bounded: a wave
// that never sees its word gives up after kFwdSpinLimit polls and raises the error word (the launch then reports it;
// it cannot hang the GPU).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kFwdSeqStride = 1 << 12;   // > stretches of one forward pass (N / G + 2)
enum FwdSyncWord { kSyPub = 0, kSyErr = 1, kSyCons0 = 2, kSyCons1 = 3, kSyA = 4, kSyS = 5, kSyV0 = 6, kSyV1 = 7,
                   kSyEAhead0 = 8,  // + wave index 0..2: iteration whose expansions-ahead that wave has finished (all modes)
                   kSyWords = 16 };
static_assert(kSyErr - kSyEAhead0 == kEAheadToErrWord, "forward_phase3 raises the error word relative to the E-ahead words");
template <bool SOFT>
struct FwdSync {
  int* w;    // kSyWords ints in LDS, 8-byte aligned (SOFT only)
  int base;  // sequence offset of this forward pass
  // LDS operations of one wavefront execute in issue order, so a sequence word written behind the data it announces
  // needs no s_waitcnt in front of it, and the data reads behind a poll need none either: wavefront-scope fences keep
  // the COMPILER from reordering them, the hardware does not.
  static ALTRO_DEV int peek(const int* p) {
    const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return v;
  }
  static ALTRO_DEV void post(int* p, int v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  ALTRO_DEV void wait_for(int word, int value) const {
    for (int tries = 0; tries < kFwdSpinLimit; ++tries)
      if (peek(w + word) >= value) return;
    post(w + kSyErr, 1);
  }
  // producer: stretch j is in its slots / consumers: before the first read of stretch j
  ALTRO_DEV void publish(int j) const {
    if (SOFT) post(w + kSyPub, base + j + 1);
    else lds_barrier();
  }
  ALTRO_DEV void await(int j) const {
    if (SOFT) wait_for(kSyPub, base + j + 1);
    else lds_barrier();
  }
  // consumer c (0 cost wave, 1 auxiliary wave) has read stretch j / producer: both have (before it rewrites those slots)
  ALTRO_DEV void consumed(int c, int j) const {
    if (SOFT) post(w + kSyCons0 + c, base + j + 1);
  }
  // The producer looks at the consumers' words ONE STRETCH EARLY (right behind a publish: the load's latency hides
  // behind the next stretch's arithmetic); the consumers, faster than the rollout chain, have posted by then, and the
  // poll loop behind a stale look is the rare path.
  ALTRO_DEV long long look_consumed() const {
    if (!SOFT) return 0;
    return __hip_atomic_load(reinterpret_cast<const long long*>(w + kSyCons0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  ALTRO_DEV void await_consumed(int j, long long look) const {
    if (SOFT) {
      const int need = base + j + 1;
      const int c0 = __builtin_amdgcn_readfirstlane((int)(look & 0xffffffffll));
      const int c1 = __builtin_amdgcn_readfirstlane((int)(look >> 32));
      if (c0 >= need && c1 >= need) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        return;
      }
      wait_for(kSyCons0, need);
      wait_for(kSyCons1, need);
    }
  }
  // A: the auxiliary wave's verdicts are in LDS (hardware: all three waves take the barrier)
  ALTRO_DEV void signal_a() const {
    if (SOFT) post(w + kSyA, base + 1);
    else lds_barrier();
  }
  ALTRO_DEV void await_a() const {
    if (SOFT) wait_for(kSyA, base + 1);
    else lds_barrier();
  }
  ALTRO_DEV void pass_a() const {  // the rollout wave has no business at A
    if (!SOFT) lds_barrier();
  }
  // S: the selection is in LDS, this wave's candidate / trajectory stores have drained
  ALTRO_DEV void signal_s() const {
    if (SOFT) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      post(w + kSyS, base + 1);
    } else {
      __syncthreads();
    }
  }
  ALTRO_DEV void await_s() const {
    if (SOFT) wait_for(kSyS, base + 1);
    else __syncthreads();
  }
  // V: the shares of the violation of the two other waves (c = 0 rollout wave, 1 auxiliary wave) are in LDS
  ALTRO_DEV void signal_v(int c) const {
    if (SOFT) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      post(w + kSyV0 + c, base + 1);
    } else {
      __syncthreads();
    }
  }
  ALTRO_DEV void await_v() const {
    if (SOFT) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      wait_for(kSyV0, base + 1);
      wait_for(kSyV1, base + 1);
    } else {
      __syncthreads();
    }
  }
};

// iLQR::RolloutClosedLoop's bound checks (ilqr.hpp:484-495), evaluated by the auxiliary wave so that they
// stay off the rollout wave's serial chain.  Step k of the rollout fails with kStateLimit when
// ||x_{k+1}|| > state_max, else with kControlLimit when ||u_k|| > control_max; the first failing step
// decides.  The rollout wave keeps integrating a failed trial; nothing reads the result.
// The verdicts of all trials of the wave live in wave-uniform bit masks (scalar registers, bit = trial):
// the common case -- no trial past a limit -- costs two compares (which write a scalar mask directly),
// a handful of scalar operations and one scalar branch per knot.
struct BoundMasks {
  unsigned long long ok = ~0ull, state = 0ull, ctrl = 0ull, pend = 0ull;
  // One rollout step: over_x = "x_{k+1} is past its limit", over_u_next = "u_{k+1} is past its limit"
  // (remembered for the next step).  The caller masks out x_0 (never checked) and idle lanes.
  ALTRO_DEV void step(unsigned long long over_x, unsigned long long over_u_next) {
    const unsigned long long hit = (over_x | pend) & ok;
    if (hit != 0ull) {
      state |= over_x & ok;
      ctrl |= pend & ok & ~over_x;
      ok &= ~hit;
    }
    pend = over_u_next;
  }
  ALTRO_DEV bool lane_ok(int bit) const { return ((ok >> bit) & 1ull) != 0ull; }
  ALTRO_DEV int lane_status(int bit) const {
    return ((state >> bit) & 1ull) ? (int)ALTRO_STATE_LIMIT : ((ctrl >> bit) & 1ull) ? (int)ALTRO_CONTROL_LIMIT
                                                                                   : (int)ALTRO_UNSOLVED;
  }
};

// The value lanes 32..63 hold, delivered to lanes 0..31 (v_permlane32_swap: no LDS round trip)
ALTRO_DEV double from_upper_half(double x) {
  const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto bb = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)bb[1], (int)a[1]);
}

// The auxiliary wave of the forward pass: the rollout's bound checks (BoundMasks), the gradient measure of the
// trial (ilqr.hpp:574-583 with the trial's controls) and the candidate stores, for the knots 0..N, then the
// verdicts for the cost wave (flags, gsx).  None of it feeds the cost, so it runs beside the cost wave.
//
// PAIRED (the persistent tail kernel: one instance = 20 trials per workgroup, two thirds of the lanes would idle):
// the rollout wave publishes two knots per barrier, and the two halves of this wave take one each -- lanes 0..31
// knot 2j, lanes 32..63 knot 2j+1 of trial (lane & 31) -- so the wave's instruction stream runs once per PAIR of
// knots.  What is sequential over the knots stays sequential: the bound verdicts are scalar masks stepped in knot
// order, and the gradient measure is summed in the lower half in knot order (the upper half's term arrives through
// v_permlane32_swap), bit-identical to the one-knot-at-a-time loop.
template <class T, class M, bool PAIRED, bool SOFT = false, int GB = kSyncBatched>
ALTRO_DEV void aux_wave_run(int N, const DevOpts& o, const T* sKD, int kd_stride, int kd_off, const T* xch, int lane,
                            bool valid, T* cand_inst, int cand_front, int* flags, double* gsx, bool grad,
                            const FwdSync<SOFT>& sy = FwdSync<SOFT>{nullptr, 0}, const T* sCost = nullptr,
                            double* J0_out = nullptr) {
  constexpr int G = PAIRED ? kSyncFused : GB;
  constexpr int n = M::n, m = M::m, nm = n + m;
  constexpr int LS = kLineSearchLanes;
  const bool check = o.check_forwardpass_bounds != 0;
  const T smax2 = T(o.state_max) * T(o.state_max), umax2 = T(o.control_max) * T(o.control_max);
  const int half = PAIRED ? (lane >> 5) : 0;
  const int col = PAIRED ? (lane & 31) : lane;           // the rollout-wave lane whose trial this lane follows
  const bool mine = PAIRED ? col < LS : valid;  // (a PAIRED workgroup has exactly one, live, instance)
  // the candidate slot of this lane's trial: PAIRED (candidates in LDS) one per trial; batched sweeps: CandLayout
  const CandLayout CL(PAIRED ? LS : cand_front, o.line_search_max_iterations < LS ? o.line_search_max_iterations : LS);
  const int cslot = PAIRED ? col : CL.store_slot(lane % LS);
  const unsigned cstride = (unsigned)(PAIRED ? LS : CL.slots) * (unsigned)nm;
  const bool stores = mine && cslot >= 0;
  T* const candp = cand_inst + (unsigned)(cslot >= 0 ? cslot : 0) * (unsigned)nm;
  BoundMasks bm;
  double gs = 0.0;
  // PAIRED (persistent kernel): this wave also sums the running cost of the current trajectory, J0 = costs_.sum() in
  // knot order (ilqr.hpp:326-334, 516), two knots per trip of its loop -- the wave has the slack, and the hundred
  // dependent additions (with their LDS reads ~6 500 cycles on a wave of their own between two barriers) leave the
  // critical path of the iteration altogether.  The cost wave needs J0 behind barrier A only.
  double J0 = 0.0;
  constexpr int kStep = PAIRED ? 2 : 1;
  for (int k0 = 0; k0 <= N; k0 += kStep) {
    if (PAIRED && sCost) {
      const double c0 = (double)sCost[k0], c1 = (double)sCost[k0 + 1 <= N ? k0 + 1 : N];
      J0 += c0;
      if (k0 + 1 <= N) J0 += c1;  // (wave-uniform)
    }
    if (consumer_syncs_before(k0, G)) {  // (xbar, ubar) of knots k0 .. k0+G-1 of every trial are published
      if (k0 > 0) sy.consumed(1, k0 / G - 1);
      sy.await(k0 / G);
    }
    const int k = k0 + half;                       // k <= N + 1; N is the terminal knot (a state only)
    const bool inner = k < N;
    const T* slot = xch + fwd_slot(k, G) * (nm * kBlock);
    T xb[n], ub[m], d[m];
#pragma unroll
    for (int i = 0; i < n; ++i) xb[i] = slot[i * kBlock + col];
#pragma unroll
    for (int i = 0; i < m; ++i) ub[i] = slot[(n + i) * kBlock + col];  // stale at the terminal knot: masked below
    const int kg = inner ? k : N - 1;
    if (grad) {  // wave-uniform
#pragma unroll
      for (int i = 0; i < m; ++i) d[i] = sKD[kg * kd_stride + kd_off + i];
    }
    if (check) {  // wave-uniform
      T sx = T(0), su = T(0);
#pragma unroll
      for (int i = 0; i < n; ++i) sx += xb[i] * xb[i];
#pragma unroll
      for (int i = 0; i < m; ++i) su += ub[i] * ub[i];
      // ||x||_2 > state_max  <=>  ||x||^2 > state_max^2: no sqrt needed.  x_0 is given, not checked.
      const unsigned long long ox = __ballot(mine && k > 0 && k <= N && sx > smax2);
      const unsigned long long ou = __ballot(mine && inner && su > umax2);
      if (PAIRED) {
        bm.step(ox & 0xffffffffull, ou & 0xffffffffull);
        bm.step(ox >> 32, ou >> 32);
      } else {
        bm.step(ox, ou);
      }
    }
    if (grad) {
      const double r = inner ? (double)grad_term<T, m>(d, ub) : 0.0;  // gs >= +0: adding +0 leaves every bit alone
      gs += r;
      if (PAIRED) gs += from_upper_half(r);
    }
    if (stores && k <= N) {  // idle lanes (and trials without a slot) must not touch the candidates
      T* cand = candp + (unsigned)k * cstride;
#pragma unroll
      for (int i = 0; i < n; ++i) cand[i] = xb[i];
#pragma unroll
      for (int i = 0; i < m; ++i) cand[n + i] = ub[i];  // (the terminal knot's slot has room for the unused u)
    }
  }
  if (J0_out) *J0_out = J0;
  const int bit = PAIRED ? col : lane;
  if (!PAIRED || half == 0) {
    flags[lane] = bm.lane_ok(bit) ? 1 : 0;
    flags[kBlock + lane] = bm.lane_status(bit);
    gsx[lane] = gs;
  }
}

// HOIST (the persistent kernel of problems with circle constraints: one workgroup per CU, registers are free): the
// circles' centres and radii in registers, their multipliers fetched one knot ahead.  The batched sweeps keep the
// runtime loop: the 30 VGPRs would cost them a wave per SIMD.
template <class T, class M, int FK, bool HOIST, bool SOFT = false>
ALTRO_DEV void cost_consumer_run(const CtxL<T>& C, const ProblemDesc* pd, const DevArrays<T>& A, const KnotRun& run,
                                 int kend, const T* xch, int lane, double& J, int G,
                                 const FwdSync<SOFT>& sy = FwdSync<SOFT>{nullptr, 0}) {
  constexpr int n = M::n, m = M::m, nm = n + m;
  const KnotClass& kc = pd->cls[run.cls];
  RunConsts<T, n, m> RC;
  load_run_consts<T, n, m>(C, pd, kc, RC);
  constexpr bool kHasB = (FK == kFastB || FK == kFastCB || FK == kFastBC);
  constexpr bool kHasC = (FK == kFastC || FK == kFastCB || FK == kFastBC);
  constexpr int kCi = (FK == kFastBC) ? 1 : 0;
  constexpr int kBi = (FK == kFastCB) ? 1 : 0;
  int c_p = 0, c_pi = 0, c_off = 0, c_row = 0, b_row = 0;
  if (kHasC) {
    c_p = kc.con[kCi].p;
    c_pi = kc.con[kCi].per_instance;
    c_off = kc.con[kCi].param_off;
    c_row = kc.con[kCi].row_off;
  }
  if (kHasB) b_row = kc.con[kBi].row_off;
  int nrows = kc.nrows, rowbase = run.rowbase, k_begin = run.k_begin;
  // loop invariants that came from the kernel arguments: in scalar registers for the whole loop (pin_s), or the
  // compiler re-loads them from memory at every knot -- a scalar-load round trip right behind the barrier
  pin_s(c_p);
  pin_s(c_pi);
  pin_s(c_off);
  pin_s(c_row);
  pin_s(b_row);
  pin_s(nrows);
  pin_s(rowbase);
  pin_s(k_begin);
  // Jc / (2 rho): the penalty hardly ever changes from knot to knot, so its reciprocal is kept (one IEEE
  // division, ~70 cycles, when it changes -- behind a WAVE-UNIFORM test, or the compiler turns the rare branch
  // into an unconditional division and a select) and the quotient comes from Markstein's correction step, which
  // returns the correctly rounded Jc / (2 rho) -- the same bits -- in 3 operations
  T inv_rho2 = T(0), inv_for = T(0);
  auto div_2rho = [&](T num, T rho) __attribute__((always_inline)) -> T {
    const T den = T(2) * rho;
    const bool stale = den != inv_for;
    if (__ballot(stale) != 0ull) {
      const T fresh = T(1) / den;
      inv_rho2 = stale ? fresh : inv_rho2;
      inv_for = stale ? den : inv_for;
    }
    const T q = num * inv_rho2;
    const T r = fma(-den, q, num);
    return fma(r, inv_rho2, q);
  };
  // multipliers of the bound rows: fetched one knot AHEAD, before the barrier (they do not depend on the hand-off),
  // so that behind the barrier only the slot itself has to make the LDS round trip
  T blam[2 * m], brho = T(1);
  // the circles of the run (obstacle_constraints.hpp:69-127): centres and radii are the same at every knot -- hoisted
  // into registers -- and their multipliers are fetched one knot ahead with the bound rows.  (A runtime loop over the
  // circles with its LDS reads in the body cost three dependent LDS round trips per knot and trial: the cost wave was
  // the critical wave of the knot loop on the obstacle problems.)
  constexpr bool kHoistC = HOIST && kHasC;
  constexpr int kFC = kHoistC ? kMaxFastCircles : 1;
  T ccx[kFC], ccy[kFC], crr[kFC], clam[kFC], crho = T(1);
#pragma unroll
  for (int i = 0; i < kFC; ++i) {
    ccx[i] = ccy[i] = crr[i] = clam[i] = T(0);
    if (kHoistC && i < c_p) {
      ccx[i] = C.par(c_pi, c_off, 3 * i);
      ccy[i] = C.par(c_pi, c_off, 3 * i + 1);
      crr[i] = C.par(c_pi, c_off, 3 * i + 2);
    }
  }
  auto fetch_bound_rows = [&](int k, T* lam_out, T& rho_out, T* clam_out, T& crho_out) __attribute__((always_inline)) {
    const int rb = rowbase + (k - k_begin) * nrows;
    if (kHasB) {
      rho_out = C.pen(rb + b_row);
#pragma unroll
      for (int j = 0; j < 2 * m; ++j) lam_out[j] = C.lam(rb + b_row + j);
    }
    if (kHoistC) {
      crho_out = C.pen(rb + c_row);
#pragma unroll
      for (int i = 0; i < kFC; ++i)
        if (i < c_p) clam_out[i] = C.lam(rb + c_row + i);
    }
  };
  fetch_bound_rows(k_begin, blam, brho, clam, crho);
  for (int k = k_begin; k < kend; ++k) {
    if (consumer_syncs_before(k, G)) {  // (xbar, ubar) of knots k .. k+G-1 of every trial are published
      if (k > 0) sy.consumed(0, k / G - 1);
      sy.await(k / G);
    }
    const T* slot = xch + fwd_slot(k, G) * (nm * kBlock);
    T xb[n], ub[m];
#pragma unroll
    for (int i = 0; i < n; ++i) xb[i] = slot[i * kBlock + lane];
#pragma unroll
    for (int i = 0; i < m; ++i) ub[i] = slot[(n + i) * kBlock + lane];
    const int rb = rowbase + (k - k_begin) * nrows;
    // this knot's multipliers are in registers; the next knot's are requested now (the last request of a run is
    // clamped to its own knot and unused)
    T blam_next[2 * m], brho_next = T(1), clam_next[kFC], crho_next = T(1);
#pragma unroll
    for (int i = 0; i < kFC; ++i) clam_next[i] = T(0);
    fetch_bound_rows(k + 1 < kend ? k + 1 : k, blam_next, brho_next, clam_next, crho_next);
    if (FK == kFastGeneric) {
      J += (double)knot_cost_fast<T, n, m>(C, pd, kc, RC, rb, xb, ub);
    } else {
      T xQx = T(0), uRu = T(0), qx = T(0), ru = T(0);
#pragma unroll
      for (int i = 0; i < n; ++i) {
        xQx += xb[i] * (RC.Qd[i] * xb[i]);
        qx += RC.q[i] * xb[i];
      }
#pragma unroll
      for (int i = 0; i < m; ++i) {
        uRu += ub[i] * (RC.Rd[i] * ub[i]);
        ru += RC.r[i] * ub[i];
      }
      T Jk = T(0.5) * xQx + T(0.5) * uRu + qx + ru + RC.c;
      auto circle_term = [&]() {
        const T rho = kHoistC ? crho : C.pen(rb + c_row);
        T a = T(0), bsum = T(0);
        if constexpr (kHoistC) {
#pragma unroll
          for (int i = 0; i < kFC; ++i)
            if (i < c_p) {  // wave-uniform
              T dx = xb[0] - ccx[i];
              T dy = xb[1] - ccy[i];
              T rr = crr[i];
              T c = circle_value(dx, dy, rr);
              T lam = clam[i];
              T lp = dual_proj(1, lam - rho * c);
              a += lp * lp;
              bsum += lam * lam;
            }
        } else {
          for (int i = 0; i < c_p; ++i) {
            T dx = xb[0] - C.par(c_pi, c_off, 3 * i);
            T dy = xb[1] - C.par(c_pi, c_off, 3 * i + 1);
            T rr = C.par(c_pi, c_off, 3 * i + 2);
            T c = circle_value(dx, dy, rr);
            T lam = C.lam(rb + c_row + i);
            T lp = dual_proj(1, lam - rho * c);
            a += lp * lp;
            bsum += lam * lam;
          }
        }
        T Jc = a - bsum;
        Jk += div_2rho(Jc, rho);
      };
      auto bound_term = [&]() {
        T a = T(0), bsum = T(0);
#pragma unroll
        for (int j = 0; j < m; ++j) {
          T c = RC.bnd[j] - ub[j];
          T lp = dual_proj(1, blam[j] - brho * c);
          a += lp * lp;
          bsum += blam[j] * blam[j];
        }
#pragma unroll
        for (int j = 0; j < m; ++j) {
          T c = ub[j] - RC.bnd[m + j];
          T lp = dual_proj(1, blam[m + j] - brho * c);
          a += lp * lp;
          bsum += blam[m + j] * blam[m + j];
        }
        T Jc = a - bsum;
        Jk += div_2rho(Jc, brho);
      };
      if (FK == kFastB) bound_term();
      if (FK == kFastC) circle_term();
      if (FK == kFastCB) {
        circle_term();
        bound_term();
      }
      if (FK == kFastBC) {
        bound_term();
        circle_term();
      }
      J += (double)Jk;
    }
    if (kHasB) {
      brho = brho_next;
#pragma unroll
      for (int j = 0; j < 2 * m; ++j) blam[j] = blam_next[j];
    }
    if (kHoistC) {
      crho = crho_next;
#pragma unroll
      for (int i = 0; i < kFC; ++i) clam[i] = clam_next[i];
    }
  }
}

// Phase 0 of the two-wave forward pass: copy the read-only inputs of up to kBlock / 20 instances into
// LDS with `nthreads` threads (tt = index of this thread).  kd_mode: kKdNone leaves the gains alone (the fused
// sweep kernel has the backward pass write them straight into LDS); kKdFull stages the whole gain records;
// kKdFeedforward only d (records of mP elements): the models whose gain records would fill the LDS (n = 12:
// 83 KB per instance) read K from global memory in the rollout wave instead, so that three instances share a
// workgroup instead of one.
enum KdMode { kKdNone = 0, kKdFull = 1, kKdFeedforward = 2 };
template <class T, class M>
ALTRO_DEV void forward2_stage(const DevArrays<T>& A, const ProblemDesc* pd, const FwdLds<T>& L, unsigned char* smem_raw,
                              T* sPool, int per_wave, int all, int tt, int nthreads, int kd_mode,
                              bool with_traj = true, int bid = -1, int b_fixed = -1) {
  if (bid < 0) bid = blockIdx.x;
  const bool with_kd = kd_mode != kKdNone;
  constexpr int LS = kLineSearchLanes;
  using R = Rec<T, M::n, M::m>;
  const unsigned Bp = A.Bp;
  const int N = A.N;
  {
    for (int i = tt; i < pd->npool; i += nthreads) sPool[i] = A.pool[i];
    // Every thread of the workgroup copies for every instance (wave-uniform instance index).  Memory
    // latency (~1 us from HBM / Infinity Cache, more when 500 workgroups start together) dwarfs the
    // copy itself, so ALL loads -- six arrays, up to three instances -- are issued before the first
    // LDS store: one round trip for the whole block.
    using V = typename VecOf<T>::type;
    constexpr int VN = R::V;
    const int kStride = nthreads;
    constexpr int D = 2;  // items per thread, array and instance in flight (N = 100: two rounds)
    const int perX = R::nP / VN, perU = R::mP / VN, perK = (kd_mode == kKdFeedforward ? R::mP : R::KP) / VN;
    const int kd_first = kd_mode == kKdFeedforward ? R::oD : 0;  // first record element that is staged
    const int cX = with_traj ? (N + 1) * perX : 0, cU = with_traj ? N * perU : 0, cK = N * perK, cR = L.nR, cS = L.nS;
    int cmax = (with_kd && cK > cX) ? cK : cX;
    cmax = cmax > cR ? cmax : cR;
    cmax = cmax > cS ? cmax : cS;
    constexpr int G = kBlock / LS;  // instances per wave at most
    for (int i0 = tt; i0 < cmax; i0 += kStride * D) {
      V vx[G][D], vu[G][D], vk[G][D];
      T sl[G][D], sp[G][D], si[G][D];
      int bgs[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        // (b_fixed: the persistent kernel names its instance itself -- a twin workgroup works on a shadow column)
        bgs[g] = g < per_wave ? (b_fixed >= 0 ? b_fixed : instance_of_slot(A, bid * per_wave + g, all)) : -1;
        if (bgs[g] < 0) continue;
        const int b = bgs[g];  // RECP / SOA address this instance
        auto ldrec = [&](const T* src, int per, int cnt, int EP, int vi) -> V {
          vi = vi < cnt ? vi : cnt - 1;  // clamped: the load is unconditional, the store is not
          const int k = vi / per, w = vi - k * per;
          return *reinterpret_cast<const V*>(RECP(src, k, EP) + w * VN);
        };
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const int i = i0 + j * kStride;
          if (with_traj) {
            vx[g][j] = ldrec(A.X, perX, cX, R::nP, i);
            vu[g][j] = ldrec(A.U, perU, cU, R::mP, i);
          }
          if (with_kd) {
            // gain record: stored as RS, staged as T (element offsets coincide, see load_rec_as)
            using RS = rec_scalar_t<T, M>;
            int vi = i < cK ? i : cK - 1;
            const int k = vi / perK, w = vi - k * perK;
            const RS* src = RECP((const RS*)A.KD, k, (Rec<RS, M::n, M::m>::KP)) + kd_first + w * VN;
            T* e = reinterpret_cast<T*>(&vk[g][j]);
#pragma unroll
            for (int q = 0; q < VN; ++q)  // (d-only: the last vector of the padded block may run past d; never past the record)
              e[q] = (kd_first + w * VN + q < Rec<RS, M::n, M::m>::KP) ? (T)src[(kd_first + w * VN + q < Rec<RS, M::n, M::m>::KP) ? q : 0] : T(0);
          }
          sl[g][j] = cR > 0 ? SOA(A.lam, i < cR ? i : cR - 1) : T(0);
          sp[g][j] = cR > 0 ? SOA(A.pen, i < cR ? i : cR - 1) : T(0);
          si[g][j] = cS > 0 ? SOA(A.ipool, i < cS ? i : cS - 1) : T(0);
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (bgs[g] < 0) continue;
        T* gX = reinterpret_cast<T*>(smem_raw) + g * L.total();
        T* gU = gX + L.nX;
        T* gKD = gU + L.nU;
        T* gLam = gKD + L.nKD;
        T* gPen = gLam + L.rowsP();
        T* gIp = gPen + L.rowsP();
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const int i = i0 + j * kStride;
          if (i < cX) *reinterpret_cast<V*>(gX + i * VN) = vx[g][j];
          if (i < cU) *reinterpret_cast<V*>(gU + i * VN) = vu[g][j];
          if (with_kd && i < cK) *reinterpret_cast<V*>(gKD + i * VN) = vk[g][j];
          if (i < cR) {
            gLam[i] = sl[g][j];
            gPen[i] = sp[g][j];
          }
          if (i < cS) gIp[i] = si[g][j];
        }
      }
    }
  }
}

// Body of the two-wave forward pass (128 threads).  pd: the description in the kernel arguments --
// wave-uniform accesses become scalar loads that stay in SGPRs across the serial loop.  pdg: the same
// bytes in global memory, for the per-lane (divergent) indexing of phases 2 and 3; indexing the
// by-value copy that way would force it into scratch.  FUSED: called by k_sweep_fused with the LDS
// block already filled; fh = {J0, dV0, dV1, initial_cost} handed over in LDS.
// SRC: where the rollout wave's per-knot inputs (xbar, ubar, K, d) come from in the batched sweeps --
//   kSrcLds  staged in LDS with everything else (17.7 KB per unicycle instance: two workgroups per CU);
//   kSrcKdg  large models: K from global memory one knot ahead, the rest staged;
//   kSrcGlb  small models: all four from global memory TWO knots ahead in three rotating register sets (L2-resident:
//            the backward kernel has just written the gains); LDS keeps only the multipliers and parameters, so four
//            workgroups fit a CU and the staging phase shrinks with it.
enum FwdSrc { kSrcLds = 0, kSrcKdg = 1, kSrcGlb = 2 };
// E elements of a 16-byte aligned record: 16-byte loads for the pairs, one 8-byte load for an odd last element
template <class T, int E>
ALTRO_DEV void load_elems(const T* p, T* out) {
  static_assert(sizeof(T) == 8, "fp64 records");
  using V = typename VecOf<T>::type;
#pragma unroll
  for (int i = 0; i + 1 < E; i += 2) {
    const V v = *reinterpret_cast<const V*>(p + i);
    out[i] = v.x;
    out[i + 1] = v.y;
  }
  if (E & 1) out[E - 1] = p[E - 1];
}
#ifndef ALTRO_RG_AHEAD
#define ALTRO_RG_AHEAD 2
#endif
constexpr int kRgAhead = ALTRO_RG_AHEAD;  // knots of prefetch distance of the kSrcGlb rollout wave (register sets - 1)
// ... for the large models (round 6: kSrcGlb for the 12-state model, whose staged trajectory is what keeps a second workgroup
// off the CU): one knot, two register sets -- a third set of 12 + 4 doubles and 52 gain elements does not fit 256 registers,
// and one knot of its RK4 (~2 us) covers an L2 round trip
template <class M>
constexpr int rg_ahead() {
  return M::n * M::m >= 12 ? 1 : kRgAhead;
}
// HOISTC: the circle layouts of the cost wave keep centres, radii and multipliers in registers (cost_consumer_run).
// Only the persistent kernel's variant for problems that HAVE circle constraints is built that way: the kernel sits at
// the register limit, and code of constraint kinds a problem does not have still moves the allocation of its hot
// loops (measured: +4 % per iteration on the obstacle-free config 2 when its circle arms grew).
// spec != nullptr (k_sweep_fused with a fourth wave): that wave runs the speculative backward pass of the next
// iteration (backward_mfma_body<.., SPEC>) beside the three forward waves and joins every workgroup barrier of theirs.
template <class T>
struct FwdSpec {
  T* sKD2;        // second gain block (LDS)
  int junk2;      // junk slots behind it, relative to sKD2
  double* fh2;    // hand-over values of the speculative pass
  double* inbox;  // {rho, drho} the pass assumed
  bool armed;     // speculate in this forward pass (the previous line search was rejected: a streak is likely)
};
template <class T, class M, bool FUSED, int SRC = kSrcLds, bool HOISTC = false, bool SOFT = false, bool SEG = true>
ALTRO_DEV void forward2_body(const DevArrays<T>& A, const ProblemDesc* __restrict__ pdg, const ProblemDesc* pd,
                             const DevOpts& o, int mode, int all, int per_wave, unsigned char* smem_raw,
                             const double* fh, int* active_out = nullptr, T* sCand = nullptr, double* ff = nullptr,
                             const FwdSpec<T>* spec = nullptr, const T* alpha_tab = nullptr,
                             const FwdSync<SOFT>& sy = FwdSync<SOFT>{nullptr, 0}, const T* sCost = nullptr,
                             double* fhw = nullptr, const int* eahead_words = nullptr, int eahead_waves = 0,
                             int eahead_tag = 0, int b_fixed = -1, int bid_fixed = -1) {
  static_assert(!SOFT || FUSED, "software synchronisation is a mode of the persistent kernel");
  constexpr int n = M::n, m = M::m, nm = n + m;
  constexpr int LS = kLineSearchLanes;
  using R = Rec<T, n, m>;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int grp = lane / LS;
  const int t = lane - grp * LS;
  // (batched kernel: the workgroups that share cache lines of the instance-minor arrays run on one XCD, see xcd_block)
  // (bid_fixed: the device-side sweep loop, k_sweep_loop -- a persistent workgroup names its window of the list itself)
  const int bid = bid_fixed >= 0 ? bid_fixed : (FUSED ? (int)blockIdx.x : xcd_block((int)blockIdx.x, (int)gridDim.x, A.xcd_remap));
  const int b0 = (grp < per_wave) ? (b_fixed >= 0 ? b_fixed : instance_of_slot(A, bid * per_wave + grp, all)) : -1;
  const int N = A.N;
  const bool valid = b0 >= 0;
  if (__ballot(valid) == 0ull) return;  // every wave takes the same decision
  const int b = valid ? b0 : 0;

  // ---- phase 0: stage the instance's read-only inputs in LDS (both waves copy) ------------------
  // KDG: the feedback gains stay in global memory (read by the rollout wave one knot ahead), LDS keeps d only
  constexpr bool KDG = SRC == kSrcKdg, RG = SRC == kSrcGlb;
  static_assert(!(FUSED && SRC != kSrcLds), "the fused kernel keeps the gain records in LDS");
  constexpr int kKdStride = KDG ? R::mP : R::KP, kKdOff = KDG ? 0 : R::oD;
  const FwdLds<T> L{RG ? 0 : (N + 1) * R::nP, RG ? 0 : N * R::mP, RG ? 0 : N * kKdStride, pd->total_rows, pd->nslots, R::V};
  T* sm = reinterpret_cast<T*>(smem_raw) + (grp < per_wave ? grp : 0) * L.total();
  T* sX = RG ? nullptr : sm;  // (RG: nothing of the trajectory or the gains is staged; phases 2 and 3 read d and
  T* sU = RG ? nullptr : sm + L.nX;  //  ubar from global memory)
  T* sKD = RG ? nullptr : sm + L.nX + L.nU;
  T* sLam = sm + L.nX + L.nU + L.nKD;
  T* sPen = sLam + L.rowsP();
  T* sIp = sPen + L.rowsP();
  T* sPool = reinterpret_cast<T*>(smem_raw) + per_wave * L.total();
  T* xch = sPool + L.padv(pd->npool);              // [2][nm][64] hand-off slots
  constexpr int G = FUSED ? kSyncFused : fwd_sync_batched<M, SRC>();  // knots per workgroup barrier of the knot loop (2 G hand-off slots)
  int* flags = reinterpret_cast<int*>(xch + 2 * G * nm * kBlock);  // [2][64]: ok, status of each trial
  double* gsx = reinterpret_cast<double*>(flags + 2 * kBlock);   // [64]: gradient measure of each trial
  if (!FUSED) {
    forward2_stage<T, M>(A, pd, L, smem_raw, sPool, per_wave, all, threadIdx.x, kFwdWaves * kBlock,
                         RG ? kKdNone : (KDG ? kKdFeedforward : kKdFull), !RG, bid);
    __syncthreads();
  }

  // The gradient measure of the accepted trial (ilqr.hpp:574-583 with the trial's controls).  FUSED: summed knot by
  // knot by the auxiliary wave for every trial -- its two half-waves have the slack.  Batched sweeps, where that wave
  // is the critical one: only for the winner, in phase 2 -- each knot's term into the hand-off slots (free by then),
  // summed in knot order afterwards: the same additions in the same order.
  const bool grad_in_loop = FUSED || (!RG && 16 + per_wave * N > 2 * G * nm * kBlock);  // (RG: the engine checked)
  T* const rk = grad_in_loop ? nullptr : xch + 16 + (grp < per_wave ? grp : 0) * N;

  // (the persistent kernel keeps the serial chain free of what does not change between its iterations: the initial
  //  state is knot 0 of the LDS-resident trajectory -- a rollout never moves it -- and the step lengths, 19 dependent
  //  divisions for the last lane, come from a table the kernel fills once)
  T x0[R::nP];
  load_rec<T, R::nP>(FUSED ? sX : A.x0 + (size_t)b * R::nP, x0);
  const T hh = T(pd->hstep);
  const int ls_max = o.line_search_max_iterations;
  // this lane's step length: alpha /= decrease_factor, t times (ilqr.hpp:544)
  T alpha = T(1);
  if (FUSED && alpha_tab) {
    alpha = alpha_tab[t];
  } else {
    for (int i = 0; i < t; ++i) alpha /= T(o.line_search_decrease_factor);
  }

  // candidate scratch of the batched kernel: which trials own a slot (CandLayout); FUSED keeps all LS in LDS
  const CandLayout CL(FUSED ? LS : A.cand_front, ls_max < LS ? ls_max : LS);
  const unsigned cand_inst_off = FUSED ? 0u : (unsigned)b * (unsigned)(N + 1) * (unsigned)CL.slots * (unsigned)nm;
  // behind barrier S: does an instance of this workgroup need its winner replayed?  (Every wave holds the same
  // (valid, grp) pattern over its lanes and reads the same selection words: the answer is workgroup-uniform.)
  auto wg_replays = [&]() __attribute__((always_inline)) -> bool {
    if constexpr (FUSED) return false;
    const int* selr = reinterpret_cast<const int*>(xch);
    const int trep = valid ? selr[2 * grp] : -1;
    return __ballot(trep >= 0 && CL.needs_replay(trep)) != 0ull;
  };

  // Phase 2 of the persistent kernel (one instance per workgroup): the knots of the instance over all 64 lanes of the
  // three forward waves -- one knot per lane for N = 100 instead of two per line-search lane -- each knot's copy and
  // constraint evaluation exactly as in the batched kernels; returns the wave's share of the violation (a maximum:
  // exact in any order).  `wi` = 0, 1, 2: which third of the lanes this wave is.
  auto phase2_all_lanes = [&](int wi, int t_rep, bool acc) __attribute__((always_inline)) -> T {
    const int bF = __builtin_amdgcn_readfirstlane(b);  // lane 0 of every wave holds the workgroup's instance
    const unsigned Bp = A.Bp;
    (void)Bp;
    CtxL<T> CF(A, bF, sPool, sIp, sLam, sPen);  // (one instance per workgroup: the same LDS block for every lane)
    T v = forward_phase2<T, M>(A, pdg, CF, bF, t_rep, acc, wi * kBlock + lane, kFwdWaves * kBlock, sCand, 0u, t_rep, LS, sX, sU,
                               nullptr, sKD, kKdStride, kKdOff);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = max_(v, __shfl_xor(v, off));
    return v;
  };
  if (wave == 0) {
    // ================= rollout wave: iLQR::RolloutClosedLoop (ilqr.hpp:468-499) =================
    // (the state / control limit checks of the reference run in the cost wave: RolloutBounds)
    T xb[n];
    T trig_s = T(0), trig_c = T(1);
#pragma unroll
    for (int i = 0; i < n; ++i) xb[i] = x0[i];
    // One knot of the rollout.  The nominal knot (xbar, ubar, K, d) comes from LDS one knot AHEAD: the
    // reads for knot k + 1 are issued before the arithmetic of knot k, so their latency hides behind
    // it; two register sets alternate (the loop is unrolled by two, no copies).
    // (kSrcGlb keeps the gain record in its storage type -- fp32 under WithRec32 -- while it is in flight and converts
    //  at use: a third fewer registers per set)
    // (round 6: kSrcKdg too -- the 12-state model's K is 48 of the 52 elements of a record, and two register sets of it as
    //  doubles were 208 of the rollout wave's 480 registers)
    using RSn = rec_scalar_t<T, M>;
    constexpr bool kKdInStorageType = RG || KDG;
    using KdT = std::conditional_t<kKdInStorageType, RSn, T>;
    struct Nominal {
      T xk[R::nP], uk[R::mP];
      KdT kd[kKdInStorageType ? (int)Rec<RSn, n, m>::KP : (int)R::KP];
    };
    auto fetch = [&](int k, Nominal& q) __attribute__((always_inline)) {
      const int kc = k < N ? k : N - 1;
      if constexpr (RG) {
        using RS = rec_scalar_t<T, M>;
        using RR = Rec<RS, n, m>;
        const unsigned Bp = A.Bp;
        // (exactly n / m elements: a padded element would hold a register pair for the two knots the load is in flight,
        //  in each of the three sets -- the difference between two and three waves per SIMD for the unicycle)
        load_elems<T, n>(RECP(A.X, kc, R::nP), q.xk);
        load_elems<T, m>(RECP(A.U, kc, R::mP), q.uk);
        load_rec<RS, RR::KP>(RECP((const RS*)A.KD, kc, RR::KP), q.kd);
      } else {
        load_rec<T, R::nP>(sX + kc * R::nP, q.xk);
        load_rec<T, R::mP>(sU + kc * R::mP, q.uk);
        if constexpr (KDG) {
          using RS = rec_scalar_t<T, M>;
          using RR = Rec<RS, n, m>;
          load_rec<RS, RR::KP>((const RS*)A.KD + ((size_t)(unsigned)kc * (unsigned)A.Bp + (unsigned)b) * RR::KP, q.kd);
        } else {
          load_rec<T, R::KP>(sKD + kc * R::KP, q.kd);
        }
      }
    };
    const long long st_w0 = ALTRO_STAMP_T0();
    long long cons_look = 0;  // (software synchronisation) the consumers' progress words as of the last publish
    // (REPLAY -- batched kernel only, see CandLayout: the same knot for the trial phase 2 is about to read, whose depth owned
    //  no candidate slot: nothing is handed to the consumer waves, lane t = 0 of the instance writes the shared slot)
    T* rep_cand = nullptr;  // this lane's candidate slot of knot 0 in a replay (null: the lane stores nothing)
    const unsigned rep_stride = (unsigned)CL.slots * (unsigned)nm;
    auto knot = [&](auto replay_tag, int k, const Nominal& cur, Nominal& nxt) __attribute__((always_inline)) {
      constexpr bool REPLAY = decltype(replay_tag)::value;
      T ub[m], xn[n];
      fetch(k + (RG ? rg_ahead<M>() : 1), nxt);
#pragma unroll
      for (int i = 0; i < m; ++i) {
        T sacc = T(0);
#pragma unroll
        for (int l = 0; l < n; ++l) sacc += (T)cur.kd[R::oK + i + l * m] * (xb[l] - cur.xk[l]);
        ub[i] = cur.uk[i] + sacc + (T)cur.kd[R::oD + i] * alpha;
      }
      if constexpr (REPLAY) {
        if (rep_cand) {
          T* cand = rep_cand + (unsigned)k * rep_stride;
#pragma unroll
          for (int i = 0; i < n; ++i) cand[i] = xb[i];
#pragma unroll
          for (int i = 0; i < m; ++i) cand[n + i] = ub[i];
        }
      } else {
        T* slot = xch + fwd_slot(k, G) * (nm * kBlock);
        // (software synchronisation: the slots of stretch j are those of stretch j - 2, which both consumers must have read)
        if (SOFT && (k & (G - 1)) == 0 && k >= 2 * G) sy.await_consumed(k / G - 2, cons_look);
#pragma unroll
        for (int i = 0; i < n; ++i) slot[i * kBlock + lane] = xb[i];
#pragma unroll
        for (int i = 0; i < m; ++i) slot[(n + i) * kBlock + lane] = ub[i];
      }
      if constexpr (M::kHasCarriedTrig) {
        // sin / cos of the heading ride along from the previous step (see rk4_fused_sc)
        if ((k % M::kTrigResync) == 0) sincos_(xb[2], &trig_s, &trig_c);
        M::template rk4_fused_sc<T, false>(xb, ub, hh, xn, trig_s, trig_c);
      } else if constexpr (model_knot_path<M>::value || !M::kHasFusedRk4) {
        // (round 4: per-knot steps, times and models -- Trajectory::SetStep / SetTime, time-varying or discrete user
        //  dynamics, a model per knot -- run on this kernel too: three wave-uniform scalar loads per knot.  Models with a
        //  hand-fused RK4 keep one step as a loop invariant; the engine sends their per-knot trajectories to k_forward.)
        discrete_step<T, M>(xb, ub, A.hk ? T(A.hk[k]) : hh, xn, time_of(A, k), model_of(A, k));
      } else {
        discrete_step<T, M>(xb, ub, hh, xn);
      }
#pragma unroll
      for (int i = 0; i < n; ++i) xb[i] = xn[i];
      if constexpr (!REPLAY) {
        if (producer_syncs_after(k, N, G)) {
          sy.publish(k / G);
          if (SOFT) cons_look = sy.look_consumed();
        }
      }
    };
    auto all_knots = [&](auto replay_tag) __attribute__((always_inline)) {
      if constexpr (RG && rg_ahead<M>() == 2) {
        // three register sets: knot k uses set k % 3 and refills the set of knot k - 1 with knot k + 2
        Nominal q0, q1, q2;
        fetch(0, q0);
        fetch(1, q1);
        int k = 0;
        for (; k + 2 < N; k += 3) {
          knot(replay_tag, k, q0, q2);
          knot(replay_tag, k + 1, q1, q0);
          knot(replay_tag, k + 2, q2, q1);
        }
        if (k < N) knot(replay_tag, k, q0, q2);
        if (k + 1 < N) knot(replay_tag, k + 1, q1, q0);
      } else {
        Nominal qa, qb;
        fetch(0, qa);
        int k = 0;
        for (; k + 1 < N; k += 2) {
          knot(replay_tag, k, qa, qb);
          knot(replay_tag, k + 1, qb, qa);
        }
        if (k < N) knot(replay_tag, k, qa, qb);
      }
    };
    all_knots(std::false_type{});
    // final hand-off: x_N
    T* slot = xch + fwd_slot(N, G) * (nm * kBlock);
    if (SOFT && (N & (G - 1)) == 0 && N >= 2 * G) sy.await_consumed(N / G - 2, cons_look);
#pragma unroll
    for (int i = 0; i < n; ++i) slot[i * kBlock + lane] = xb[i];
    sy.publish(N / G);  // barrier N (producer_syncs_after(N, N))
    ALTRO_STAMP_ADD(0, st_w0);
    const long long st_w0b = ALTRO_STAMP_T0();
    sy.pass_a();        // barrier A (auxiliary wave -> cost wave)
    // phase 2 is shared by all waves: wait for the selection, take every third block of knots
    sy.await_s();  // barrier S
    if constexpr (!FUSED) {
      // REPLAY (CandLayout): the trial phase 2 will read had no candidate slot -- a winner deeper than `front`, rare -- so
      // the rollout runs once more for it.  All lanes of the instance follow the same trial (same step length, same
      // operations as the lane that ran it first: same bits), lane t = 0 writes the shared slot; instances of this
      // workgroup that need no replay ride along without storing.  The other two waves wait at the extra barrier.
      const int* selr = reinterpret_cast<const int*>(xch);
      const int trep = valid ? selr[2 * grp] : -1;
      const bool need = trep >= 0 && CL.needs_replay(trep);
      if (__ballot(need) != 0ull) {
        if (need) {
          alpha = T(1);
          for (int i = 0; i < trep; ++i) alpha /= T(o.line_search_decrease_factor);
        }
        T x0r[R::nP];
        load_rec<T, R::nP>(A.x0 + (size_t)b * R::nP, x0r);
#pragma unroll
        for (int i = 0; i < n; ++i) xb[i] = x0r[i];
        trig_s = T(0);
        trig_c = T(1);
        rep_cand = (need && t == 0) ? A.trial + (cand_inst_off + (unsigned)CL.front * (unsigned)nm) : nullptr;
        all_knots(std::true_type{});
        if (rep_cand) {
          T* cand = rep_cand + (unsigned)N * rep_stride;
#pragma unroll
          for (int i = 0; i < n; ++i) cand[i] = xb[i];
        }
        __syncthreads();  // (drains the stores: s_waitcnt vmcnt(0) in front of the barrier) -- "R"
      }
    }
    ALTRO_STAMP_ADD(1, st_w0b);
    const long long st_w0c = ALTRO_STAMP_T0();
    {
      const int* sel = reinterpret_cast<const int*>(xch);
      T* vpart = xch + 8;
      T viol = T(0);
      if (FUSED) {
        // one instance per workgroup: its knots over ALL lanes of the three waves (one knot per lane for N = 100)
        viol = phase2_all_lanes(0, sel[0], sel[1] != 0);
        if (lane == 0) vpart[0] = viol;
      } else if (valid) {
        CtxL<T> C0(A, b, sPool, sIp, sLam, sPen);
        const int trep = sel[2 * grp];
        viol = forward_phase2<T, M>(A, pdg, C0, b, trep, sel[2 * grp + 1] != 0, t, kFwdWaves * LS, FUSED ? sCand : A.trial,
                                    cand_inst_off, CL.read_slot(trep), CL.slots, FUSED ? sX : nullptr, FUSED ? sU : nullptr, rk,
                                    sKD, kKdStride, kKdOff);
        T vm = viol;
        for (int j = 0; j < LS; ++j) vm = max_(vm, __shfl(viol, grp * LS + j));
        if (t == 0) vpart[grp] = vm;
      }
    }
    sy.signal_v(0);  // barrier V
    ALTRO_STAMP_ADD(2, st_w0c);
    return;
  }

  // candidate scratch [k][trial][x|u]: in global memory (instance-major), or -- FUSED, one instance per
  // workgroup and per CU -- in LDS, so that the knot loop issues no global store at all
  T* const cand_base = FUSED ? sCand : A.trial;
  const unsigned cand_off0 = cand_inst_off;
  if (wave == 2) {
    // ================= auxiliary wave: bound checks, gradient measure, candidate stores ===========
    const long long st_w2 = ALTRO_STAMP_T0();
    double J0_run = 0.0;
    aux_wave_run<T, M, FUSED, SOFT, G>(N, o, sKD, kKdStride, kKdOff, xch, lane, valid, cand_base + cand_off0, A.cand_front, flags,
                                    gsx, grad_in_loop, sy, FUSED ? sCost : nullptr, &J0_run);
    if (FUSED && lane == 0) {
      // J0 of the expansion step and, on the first iteration of an inner solve, stats_.initial_cost (ilqr.hpp:298);
      // ff[4] / ff[5]: LDS mirrors of initial_cost / need_init_cost (phase 3 sets ff[5] when a new inner solve begins)
      A.J0[b] = J0_run;
      double ic = ff[4];
      if (ff[5] != 0.0) {
        ic = J0_run;
        A.initial_cost[b] = J0_run;
        A.need_init_cost[b] = 0;
      }
      fhw[0] = J0_run;
      fhw[3] = ic;
      ff[4] = ic;
      ff[5] = 0.0;
    }
    ALTRO_STAMP_ADD(7, st_w2);
    sy.signal_a();  // barrier A
    sy.await_s();   // barrier S
    if (wg_replays()) __syncthreads();  // barrier R: the rollout wave has rewritten the shared candidate slot
    {
      const int* sel = reinterpret_cast<const int*>(xch);
      T* vpart2 = xch + 12;
      if (FUSED) {
        const T viol = phase2_all_lanes(2, sel[0], sel[1] != 0);
        if (lane == 0) vpart2[0] = viol;
      } else if (valid) {
        CtxL<T> C2(A, b, sPool, sIp, sLam, sPen);
        const int trep = sel[2 * grp];
        const T viol = forward_phase2<T, M>(A, pdg, C2, b, trep, sel[2 * grp + 1] != 0, t + 2 * LS, kFwdWaves * LS, cand_base,
                                            cand_off0, CL.read_slot(trep), CL.slots, FUSED ? sX : nullptr, FUSED ? sU : nullptr,
                                            rk, sKD, kKdStride, kKdOff);
        T vm = viol;
        for (int j = 0; j < LS; ++j) vm = max_(vm, __shfl(viol, grp * LS + j));
        if (t == 0) vpart2[grp] = vm;
      }
    }
    sy.signal_v(1);  // barrier V
    return;
  }

  if constexpr (FUSED && M::n <= 3 && M::m <= 2) {
    if (wave == 3) {
      // ============ fourth wave: the backward pass of the NEXT iteration, assuming this line search fails ============
      // (ilqr.hpp:550: a rejected step raises the regularisation; nothing else changes)
      double rho_in = fh[4], drho_in = fh[5];
      increase_reg(o, &rho_in, &drho_in);
      if (lane == 0) {
        spec->inbox[0] = rho_in;
        spec->inbox[1] = drho_in;
        spec->fh2[7] = 0.0;
      }
      int nbar = 0;
      const long long st_w3 = ALTRO_STAMP_T0();
      // (not armed: the fourth wave just keeps the barrier count)
      if (spec->armed)
        backward_mfma_body<T, M, false, true, true>(A, o, 0, lane, blockIdx.x, nullptr, spec->sKD2, spec->junk2, spec->fh2,
                                                    rho_in, drho_in, SOFT ? nullptr : &nbar, b_fixed);
      // (the pass placed its barriers itself; whatever is left of the N / 2 + 1 of the knot loop and A, S, V.  With
      //  software synchronisation the forward waves take no hardware barrier: the recursion ran at its own pace)
      if (spec->armed) ALTRO_STAMP_ADD(8, st_w3);
      if (!SOFT)
        for (const int bars = N / G + 1 + 3; nbar < bars; ++nbar) __builtin_amdgcn_s_barrier();
      else
        sy.await_s();  // (the selection is published: the caller reads it -- expansions ahead, k_sweep_fused)
      return;
    }
  }
  // ===================== cost wave: iLQR::Cost per trial + everything after ======================
  CtxL<T> C(A, b, sPool, sIp, sLam, sPen);
  const long long st_w1 = ALTRO_STAMP_T0();
  double J0 = FUSED ? 0.0 : A.J0[b];  // (FUSED: summed by the auxiliary wave during the knot loop, read behind barrier A)
  const double dV0 = FUSED ? fh[1] : A.dV0[b], dV1 = FUSED ? fh[2] : A.dV1[b];
  InstPre pre = load_inst_pre(A, b);  // consumed by the state machine at the very end
  if (FUSED) {
    pre.rho_reg = fh[4];
    pre.drho = fh[5];
  }
  double J = 0.0;
  for (int r = 0; r < pd->nruns; ++r) {
    const KnotRun run = pd->runs[r];
    const int kend = run.k_end < N ? run.k_end : N;
#define ALTRO_RUN(FK) cost_consumer_run<T, M, FK, HOISTC, SOFT>(C, pd, A, run, kend, xch, lane, J, G, sy)
    switch (run.fast) {
      case kFastNone: ALTRO_RUN(kFastNone); break;
      case kFastB: ALTRO_RUN(kFastB); break;
      case kFastCB: ALTRO_RUN(kFastCB); break;
      case kFastBC: ALTRO_RUN(kFastBC); break;
      case kFastC: ALTRO_RUN(kFastC); break;
      default: ALTRO_RUN(kFastGeneric); break;
    }
#undef ALTRO_RUN
  }
  if (consumer_syncs_before(N, G)) {  // barrier N: terminal state and rollout outcome
    if (N > 0) sy.consumed(0, N / G - 1);
    sy.await(N / G);
  }
  {
    const T* slot = xch + fwd_slot(N, G) * (nm * kBlock);
    T xN[n], uz[m];
#pragma unroll
    for (int i = 0; i < n; ++i) xN[i] = slot[i * kBlock + lane];
#pragma unroll
    for (int i = 0; i < m; ++i) uz[i] = T(0);
    const KnotRun runN = pd->runs[pd->nruns - 1];  // the terminal knot closes the last run
    const KnotClass& kcN = pd->cls[runN.cls];
    J += (double)knot_cost<T, n, m, false>(C, pd, kcN, runN.rowbase + (N - runN.k_begin) * kcN.nrows, xN, uz, nullptr);
  }
  sy.await_a();  // barrier A: the auxiliary wave's verdicts (FUSED: and the running cost)
  if (FUSED) {
    J0 = fh[0];
    pre.initial_cost = fh[3];
  }
  ALTRO_STAMP_ADD(3, st_w1);
  const long long st_w1b = ALTRO_STAMP_T0();
  const bool ok = flags[lane] != 0;
  const int st = flags[kBlock + lane];
  const double gs = gsx[lane];
  // ---- acceptance test (ilqr.hpp:528-542) and selection of the first accepted trial --------------
  const bool live = valid && (t < ls_max);
  const double expected = -(double)alpha * (dV0 + (double)alpha * dV1);
  const double z = (expected > 0.0) ? (J0 - J) / expected : -1.0;
  const bool acc = live && ok && o.line_search_lower_bound <= z && z <= o.line_search_upper_bound && J < J0;
  const unsigned long long accm = __ballot(acc);
  const unsigned long long okm = __ballot(live && ok);
  const unsigned gmask = (1u << LS) - 1u;
  const unsigned acc_g = (unsigned)(accm >> (grp * LS)) & gmask;
  const unsigned ok_g = (unsigned)(okm >> (grp * LS)) & gmask;
  const int nlive = ls_max < LS ? ls_max : LS;
  bool accepted = false;
  T alpha_sel = T(0);
  double J_sel = J0, z_sel = -1.0, g_sel = 0.0;
  int t_replay = -1;
  int last_status = ALTRO_UNSOLVED;
  if (acc_g) {
    const int tsel = __ffs(acc_g) - 1;
    const int src = grp * LS + tsel;
    alpha_sel = __shfl(alpha, src);
    J_sel = __shfl(J, src);
    z_sel = __shfl(z, src);
    g_sel = __shfl(gs, src);
    accepted = true;
    last_status = ALTRO_UNSOLVED;  // the accepted rollout was the last one run (ilqr.hpp:497)
    t_replay = tsel;
  } else {
    // the serial loop ran all trials; c_ holds the constraint values of the last trial whose rollout
    // succeeded (quirk Q6), and status_ is the outcome of the very last rollout
    last_status = __shfl(st, grp * LS + (nlive - 1));
    if (ok_g) t_replay = 31 - __clz(ok_g);
  }
  // ---- phase 2: copy the winner into Z_, evaluate the c_ it leaves behind; the knots are spread over
  //      the lanes of BOTH waves (the rollout wave has nothing else left to do)
  {
    int* sel = reinterpret_cast<int*>(xch);  // the hand-off slots are free now
    if (valid && t == 0) {
      sel[2 * grp] = t_replay;
      sel[2 * grp + 1] = accepted ? 1 : 0;
    }
  }
  sy.signal_s();  // barrier S: selection visible, candidate stores of this wave drained
  if (wg_replays()) __syncthreads();  // barrier R (see the rollout wave)
  ALTRO_STAMP_ADD(4, st_w1b);
  const long long st_w1c = ALTRO_STAMP_T0();
  T viol = T(0);
  if (FUSED) {
    // (the selection lives in the instance's 20 lanes: every lane of the wave takes lane 0's)
    viol = phase2_all_lanes(1, __builtin_amdgcn_readfirstlane(t_replay), __builtin_amdgcn_readfirstlane(accepted ? 1 : 0) != 0);
  } else if (valid) {
    viol = forward_phase2<T, M>(A, pdg, C, b, t_replay, accepted, t + LS, kFwdWaves * LS, cand_base, cand_off0,
                                CL.read_slot(t_replay), CL.slots, FUSED ? sX : nullptr, FUSED ? sU : nullptr, rk, sKD, kKdStride,
                                kKdOff);
    T vm = viol;
    for (int j = 0; j < LS; ++j) vm = max_(vm, __shfl(viol, grp * LS + j));
    viol = vm;
  }
  sy.await_v();  // barrier V: the other wave's share of the violation
  ALTRO_STAMP_ADD(5, st_w1c);
  const long long st_w1d = ALTRO_STAMP_T0();
  if (!valid) return;
  viol = max_(max_(viol, (xch + 8)[grp]), (xch + 12)[grp]);
  if (!grad_in_loop && accepted) {
    double gsum = 0.0;
    for (int k = 0; k < N; ++k) gsum += (double)rk[k];
    g_sel = gsum;
  }
  forward_phase3<T, M, !FUSED, SEG>(A, pdg, o, mode, b, grp, t, accepted, (double)alpha_sel, J_sel, z_sel, g_sel, last_status,
                       (double)viol, sKD, sU, pre, FUSED ? active_out : nullptr, FUSED ? sLam : nullptr,
                       FUSED ? sPen : nullptr, FUSED ? ff : nullptr, kKdStride, kKdOff, eahead_words, eahead_waves, eahead_tag);
  ALTRO_STAMP_ADD(6, st_w1d);
}

// Waves per SIMD the kernel is compiled for (second argument of HIP's __launch_bounds__).  EXPERIMENT SWITCH, default 1 = the
// compiler's own choice: the variant of the small models that reads its inputs from global memory exists for occupancy, and
// with fp64 records in flight it lands on 171 VGPRs -- three above the step to three waves per SIMD.  Built with
// -DALTRO_FWD_GLB_WAVES=3 it has 168 VGPRs + 20 B of scratch, the engine then picks it for config 2 (nine instances per CU
// instead of six) -- and config 2 runs 6.27 - 6.33 ms against 6.31 - 6.38 ms: the forward pass of the full batch is not
// bound by its occupancy (profiles/r05_experiments.txt #5).
#ifndef ALTRO_FWD_GLB_WAVES
#define ALTRO_FWD_GLB_WAVES 1
#endif
template <class M, int SRC>
constexpr int fwd_min_waves() {
  // (round 6: the large models' global-source variant is compiled for two waves per SIMD -- two workgroups per CU)
  return (SRC == kSrcGlb && M::n <= 4) ? ALTRO_FWD_GLB_WAVES : ((SRC == kSrcGlb && M::n * M::m >= 12) ? 2 : 1);
}
template <class T, class M, int SRC>
__global__ __launch_bounds__(kFwdWaves * kBlock, (fwd_min_waves<M, SRC>())) void k_forward2(DevArrays<T> A, const ProblemDesc* __restrict__ pdg,
                                                         const ProblemDesc pd_arg, DevOpts o, int mode, int all,
                                                         int per_wave) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  forward2_body<T, M, false, SRC>(A, pdg, &pd_arg, o, mode, all, per_wave, smem_raw, nullptr);
}

// iLQR::UpdateExpansions of one instance by `nthreads` threads, reading the trajectory, multipliers and
// parameters from the LDS block of the forward pass and the problem description from the kernel
// arguments: the knots of a run share their class, so each run is handed to whole wavefronts
// (wave-uniform class -> scalar loads) and different runs go to different waves where they fit.  No
// dependent global load is left on the path; the knot costs also go to sCost[k] (LDS).
template <class T, class M>
ALTRO_DEV void expansion_from_lds(const DevArrays<T>& A, const ProblemDesc* pd, const CtxL<T>& C, const T* sX,
                                  const T* sU, T* sCost, int b, int tid, int nthreads) {
  constexpr int n = M::n, m = M::m;
  using R = Rec<T, n, m>;
  const int N = A.N;
  const unsigned Bp = A.Bp;
  int toff = 0;
  for (int r = 0; r < pd->nruns; ++r) {
    const KnotRun run = pd->runs[r];
    const KnotClass& kc = pd->cls[run.cls];
    const int cnt = run.k_end - run.k_begin;
    int j = tid - toff;
    if (j < 0) j += nthreads;
    for (; j < cnt; j += nthreads) {
      const int k = run.k_begin + j;
      const int rb = run.rowbase + j * kc.nrows;
      T xr[R::nP], ur[R::mP];
      load_rec<T, R::nP>(sX + k * R::nP, xr);
#pragma unroll
      for (int i = 0; i < R::mP; ++i) ur[i] = T(0);
      if (k < N) load_rec<T, R::mP>(sU + k * R::mP, ur);
      T E[R::EP];
#pragma unroll
      for (int e = 0; e < R::EP; ++e) E[e] = T(0);
      const T J = knot_cost_expansion<T, n, m>(C, pd, kc, rb, xr, ur, E + R::oLx, E + R::oLu, E + R::oLxx, E + R::oLxu,
                                               E + R::oLuu);
      A.costs[(unsigned)k * Bp + (unsigned)b] = J;
      sCost[k] = J;
      if (k < N) discrete_jacobian<T, M>(xr, ur, T(pd->hstep), E + R::oAB);
      using RS = rec_scalar_t<T, M>;
      using RR = Rec<RS, n, m>;
      store_rec_as<T, RS, R::EP, RR::EP, R::eE>(RECP((RS*)A.EXP, k, RR::EP), E);
    }
    toff = (toff + ((cnt + kBlock - 1) / kBlock) * kBlock) % nthreads;
  }
}

// -------------------------------------------------------------------------------------------------
// One whole iLQR iteration of ONE instance per workgroup (128 threads), for the long tail of a batched
// solve: a few dozen stragglers iterate ~100 times after everyone else has converged, and each sweep
// is then a pure latency chain.  Fusing the three kernels removes two kernel boundaries, the staging
// of the gains (the backward wave writes them straight into the forward pass's LDS block), and hides
// the rest of the staging and the running-cost sum behind the backward recursion:
//   S  X, U, lambda, rho, parameters -> LDS (all threads, one memory round trip)
//   E  expansions of the 101 knots over the 192 threads from the LDS block (records to global
//      memory, visible to the workgroup after the barrier; knot costs also to LDS)
//   B  wave 0: MFMA backward pass (one of the four 4x4 blocks carries the instance), gains -> LDS;
//      wave 1, meanwhile: J0 = sum of the knot costs in order
//   F  all waves: the three-wave forward pass on the LDS block (forward2_body)
// Same device code as the separate kernels, hence the same numbers.  n = 3, m = 2 (fp64 or fp32 storage).
// persistent != 0: instances are independent, so the workgroup simply keeps iterating until ITS
// instance is finished (the AL state machine of phase 3 says so) -- no further launches, no host in
// the loop; *sweeps_out receives the largest number of iterations any workgroup ran.
// -------------------------------------------------------------------------------------------------
// SPEC: a FOURTH wave (the CU has four SIMDs, the three forward waves occupy three) runs the backward pass of the
// next iteration beside the forward pass of this one, on the assumption that the line search rejects every trial:
// then trajectory, multipliers and expansions stay what they are and only the regularisation -- known in advance --
// changes.  If the forward pass does end that way (and the inner solve goes on), the next iteration recomputes its
// expansions as always, takes the gains and the expected decrease from the speculative pass instead of running the
// recursion again, and goes straight to its forward pass; otherwise the speculation is dropped.  Every straggler of
// the tail sits in exactly that regime (a line search that rejects all 20 trials, iteration after iteration), so the
// two serial chains of an iteration overlap: 64 -> ~40 us.  Same arithmetic on the same inputs: same bits
// (ALTRO_HIP_SPECULATION=off runs the three-wave kernel).
// (Rounds 2 - 5 also had a HELPER mode -- the speculative pass in a workgroup of its own, one wave on any CU with a free SIMD,
//  results through global memory with release / acquire flags: it won with a few dozen stragglers and lost on a full tail
//  (config 2 5.27 -> 5.52 ms), was never a default, and went in round 6 with its two kernel variants: git history.)
// SPEC = kSpecFree: the fourth wave again, but the three forward waves synchronise through sequence words in LDS instead of
// workgroup barriers (FwdSync<true>), so the recursion and the knot loop each run at their own pace instead of in lock step.
enum SpecMode { kSpecOff = 0, kSpecWave = 1, kSpecFree = 3 };
ALTRO_DEV constexpr bool spec_has_wave4(int spec) { return spec == kSpecWave || spec == kSpecFree; }

// -------------------------------------------------------------------------------------------------
// TWIN WORKGROUPS (round 5): the second half of a straggler's rejection streak, computed beside the first half.
//
// What the tail of a batched solve is made of (profiles/r04_experiments.txt #5, scripts/probe_stragglers.py): ~2 % of
// the instances reject every trial of every line search for exactly max_iterations_inner iterations.  A rejected
// iteration changes nothing but the regularisation -- by a rule known in advance (ilqr.hpp:550, :770-786) -- and the
// iteration counters, so the state ENTERING iteration j + K of such a streak is known at iteration j, K steps of the
// scalar rule away, PROVIDED every iteration in between is rejected as well.  The reference walks those iterations one
// after the other on one core; here one straggler owns one CU for ~100 x 39 us while more than half of the CUs idle.
//
// So the launch carries, behind its `base` primary workgroups, one TWIN workgroup per slot.  A twin gets a CU when the
// quick instances have left; it reads the snapshot its primary publishes at the end of every iteration of a streak
// (counters, regularisation), CLAIMS the iterations from `start` on -- about half of what is left --, clones the
// instance into a shadow column of the per-instance arrays (index col0 + slot: every array of DevArrays is allocated
// that much wider), sets the entering state of iteration `start` there (counters advanced, regularisation stepped
// through the increase / decrease rule, cost_prev = cost_cur) and runs the SAME code as every workgroup of this kernel
// on that column: every iteration of the instance is still computed, with the inputs the sequential order would have
// given it -- by two workgroups side by side instead of one.  The primary, arriving at `start`, compares what it holds
// with what the twin assumed (bitwise: regularisation, counters, an unbroken streak since the snapshot): equal -> it hands
// the instance over and leaves, and the twin, when its clone is finished, copies the column back over the instance's
// own; different (an accepted step, an end of the inner solve, a Cholesky retry that moved the regularisation) -> it
// refuses, goes on alone, and the twin drops its work.  Either way the result is the sequential one, bit for bit
// (tests/test_fused_gpu.py::test_twin_workgroups_are_bit_identical); ALTRO_HIP_TWIN=0 launches no twins.
//
// Mailbox (global memory, one per slot, 64-bit words, relaxed agent-scope atomics = coherent across the XCDs' L2s; what
// travels through ordinary memory -- the clone's source, the column copied back -- is ordered by ONE agent-scope release
// of the primary per streak / hand-over and one acquire of the twin).  Every wait is bounded; a twin that gives up
// revokes its claim with a compare-and-swap against the primary's hand-over, so an instance always has exactly one owner.
// -------------------------------------------------------------------------------------------------
struct TwinCtl {
  unsigned long long* box;  // [cap][kTwWords], zeroed by the host before the launch
  unsigned long long* state;  // [cap] one word per slot for the pool's scan: kTwsSnap | kTwsClosed | kTwsLocked
  int base;                 // first twin block of the launch (0: no twins)
  int cap;                  // slots that own a mailbox and a shadow column
  int col0;                 // first shadow column
  int lag;                  // iterations the primary is expected to advance while a twin clones and stages (kTwinLag)
  int debug;                // stamp the mailbox (ALTRO_HIP_TWIN_DEBUG)
};
enum TwinStamp { kTsPStart = 0, kTsPSnap = 1, kTsPClaim = 2, kTsPHand = 3, kTsPLoopsAtSnap = 4, kTsTStart = 5, kTsTGo = 6, kTsTCloned = 7,
                 kTsTFirst = 8, kTsTDone = 9, kTsTVerdict = 10, kTsTCommit = 11, kTsTLoops = 12, kTsPEnd = 13 };
enum TwinWord {
  kTwSeq = 0,        // version of the newest snapshot (0: none yet)
  kTwSnap = 1,       // two snapshot buffers of 4 words: (it_inner << 32 | it_total), rho, drho, primary's loop count
  kTwClaim = 9,      // (start_it_inner << 32 | start_it_total), written last by the twin
  kTwClaimRho = 10,  // regularisation the twin assumes to enter iteration `start`
  kTwClaimDrho = 11,
  kTwClaimSnap = 12, // it_total of the snapshot the claim was derived from
  kTwHand = 13,      // 0 open, kTwOk / kTwRefused (primary), kTwRevoked (twin): set once, by compare-and-swap
  kTwHandLoops = 14, // the primary's loop count at the hand-over
  kTwWhy = 15,       // diagnostics (ALTRO_HIP_TWIN_DEBUG): why a claim was refused / what the twin did last
  kTwStamp = 16,     // diagnostics: 100 MHz wall-clock stamps of the two workgroups (TwinStamp), written when tw.debug is set
  kTwWords = 32
};
constexpr unsigned long long kTwOk = 1, kTwRefused = 2, kTwRevoked = 3;
// state word of a slot (contiguous array: a wavefront of the pool looks at 64 slots with one coalesced load):
//   kTwsSnap    the primary has a snapshot of a running streak on display (cleared when the streak breaks)
//   kTwsClosed  the primary has finished, refused or handed over: nothing more to come from this slot
//   kTwsLocked  a twin has taken the slot (compare-and-swap kTwsSnap -> kTwsSnap | kTwsLocked), or has found it not worth it
constexpr unsigned long long kTwsSnap = 1, kTwsClosed = 2, kTwsLocked = 4;
constexpr int kTwinMinRemaining = 12;   // iterations left in a streak below which a twin is not worth its start-up
// iterations of the primary's share that pay for the twin's start-up (claim seen, clone, one unspeculated iteration): the twin
// takes the iterations from snapshot + (R + lag) / 2 + 1 on.  Round 6 (mailbox stamps of config 2, ALTRO_HIP_TWIN_DEBUG): with 3
// the twins finished 15 - 140 us before their primaries handed over; with 1 the two sides end together (persistent launch 2.42 -
// 2.48 -> 2.33 - 2.42 ms, config 3's 7.4 -> 6.6 ms; A/B builds -DALTRO_TWIN_LAG=n)
#ifndef ALTRO_TWIN_LAG
#define ALTRO_TWIN_LAG 1
#endif
constexpr int kTwinLag = ALTRO_TWIN_LAG;
// polls (~3 us each) a twin waits for its primary to publish a streak.  The primary closes the mailbox when it finishes, so the
// wait ends with the primary at the latest; the bound only guards against a primary that never runs.  (Round 5, first version:
// 200 polls = 0.6 ms.  88 of ~100 stragglers of config 2 got their twin -- and the launch was as long as before: it ends with
// its slowest instance, and the twelve whose twin had given up before their streak began still walked all 100 iterations alone.)
constexpr int kTwinIdlePolls = 1 << 15;
constexpr int kTwinHandPolls = 1 << 16; // polls a finished twin waits for the primary's verdict (the primary answers at `start`)
ALTRO_DEV unsigned long long tw_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
ALTRO_DEV void tw_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a thread's earlier mailbox stores have left the CU before its later ones are issued
ALTRO_DEV void tw_order() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
ALTRO_DEV bool tw_cas(unsigned long long* p, unsigned long long expect, unsigned long long v) {
  return __hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
ALTRO_DEV unsigned long long tw_bits(double x) { return (unsigned long long)__double_as_longlong(x); }
ALTRO_DEV double tw_dbl(unsigned long long x) { return __longlong_as_double((long long)x); }

template <class T, class M, bool CIRC, int SPEC, bool SEG = false>
__global__ __launch_bounds__((spec_has_wave4(SPEC) ? kFwdWaves + 1 : kFwdWaves) * kBlock) void k_sweep_fused(
    DevArrays<T> A, const ProblemDesc* __restrict__ pdg, const ProblemDesc pd_arg, DevOpts o, int mode, int persistent,
    int* sweeps_out, TwinCtl tw) {
  constexpr int kThreads = (spec_has_wave4(SPEC) ? kFwdWaves + 1 : kFwdWaves) * kBlock;
  constexpr bool kWave4 = spec_has_wave4(SPEC), kSoft = SPEC == kSpecFree;
  using R = Rec<T, M::n, M::m>;
  constexpr int nm = M::n + M::m;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const ProblemDesc* pd = &pd_arg;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (blockIdx.x == 0 && tid == 0) publish_count(A);
  // (twin workgroups, see TwinCtl: blocks [base, ...) of the launch are a POOL -- each serves whichever slot of the launch
  //  publishes a streak first -- and block base + t owns shadow column col0 + t)
  const bool is_twin = tw.base > 0 && (int)blockIdx.x >= tw.base;
  const int tslot = is_twin ? (int)blockIdx.x - tw.base : -1;
  int slot = is_twin ? -1 : (int)blockIdx.x;
  int b_real = is_twin ? -1 : instance_of_slot(A, slot, 0);
  if (!is_twin && b_real < 0) return;  // uniform over the workgroup
  if (is_twin && (tslot >= tw.cap || !persistent)) return;
  unsigned long long* box = (!is_twin && tw.base > 0 && slot < tw.cap && persistent) ? tw.box + (size_t)slot * kTwWords : nullptr;
  int b = b_real;  // (a twin switches to its shadow column once it has claimed its share of the iterations)
  const int N = A.N;
  const unsigned Bp = A.Bp;
  // LDS: the forward block of one instance, then {pool, hand-off slots, flags} exactly as k_forward2
  // lays them out, then the fused extras: fh[4] and a junk slot per lane for the backward wave
  const FwdLds<T> L{(N + 1) * R::nP, N * R::mP, N * R::KP, pd->total_rows, pd->nslots, R::V};
  T* sm = reinterpret_cast<T*>(smem_raw);
  T* sKDf = sm + L.nX + L.nU;
  T* sPool = sm + L.total();
  T* xch = sPool + L.padv(pd->npool);
  int* flags = reinterpret_cast<int*>(xch + 2 * kSyncFused * nm * kBlock);
  double* fh = reinterpret_cast<double*>(flags + 2 * kBlock) + kBlock;  // behind the gradient slots
  const int fused_junk = (int)(reinterpret_cast<T*>(fh + 6) - sKDf);  // one junk slot per lane, in units of T

  int* active_flag = reinterpret_cast<int*>(fh + 6 + kBlock);
  double* ff = fh + 6 + kBlock + 2;                          // {rejected, rho, drho, inner_done} of the last iteration
  T* sCand = reinterpret_cast<T*>(fh + 6 + kBlock + 2 + 16);  // [N+1][20][n+m] line-search candidates (128-byte phase kept)
  // the speculative backward pass: second gain block + junk slots, hand-over values, assumed regularisation
  T* sKD2 = sCand + (size_t)(N + 1) * kLineSearchLanes * nm;
  double* fh2 = reinterpret_cast<double*>(sKD2 + N * R::KP + kBlock);
  FwdSpec<T> spec{sKD2, N * R::KP, fh2, fh2 + 8, false};
  T* const alpha_tab = reinterpret_cast<T*>(fh2 + 12);      // [20] step lengths of the line-search lanes (ilqr.hpp:544)
  int* const sync_words = reinterpret_cast<int*>(fh2 + 12 + kLineSearchLanes);  // [kSyWords] FwdSync<true> (kSpecFree), E ahead
  if (tid < kSyWords) sync_words[tid] = 0;  // (visible behind the staging barrier of the first iteration)
  T* const sCost = reinterpret_cast<T*>(fh2 + 12 + kLineSearchLanes + kSyWords / 2);  // [N + 1] knot costs of the expansion step
  T* const sCvalAhead = sCost + ((N + 2) & ~1);  // [total_rows] constraint values of the expansions computed ahead
  if (tid < kLineSearchLanes) {
    T alpha = T(1);
    for (int i = 0; i < tid; ++i) alpha /= T(o.line_search_decrease_factor);
    alpha_tab[tid] = alpha;
  }
  bool adopt = false;
  bool prev_rej = false;
  double prev_rho = -1.0, prev_drho = -1.0;
  int skipped = 0;
  int loops = 0;
  bool e_done = false;  // this iteration's expansions were computed beside phase 3 of the previous one
#ifdef ALTRO_STAMPS
  if (tid < 32) g_stamp_acc[tid] = 0;
  int spec_iters = 0;
#endif
  // ---- twin workgroups (TwinCtl): the primary's bookkeeping, the twin's claim and clone ----
  int tw_streak = 0;               // primary: consecutive rejected iterations during which the inner solve went on
  int tw_break_total = -1;         // primary: it_total left by the last iteration that was NOT one of those
  unsigned long long tw_ver = 0;   // primary, thread 0: version of the newest snapshot
  bool tw_closed = false;          // primary, thread 0: the mailbox is out of use
  bool tw_shown = false;           // primary, thread 0: the slot's state word says "snapshot on display"
  unsigned long long tw_claim = 0, tw_claim_rho = 0, tw_claim_drho = 0;  // primary, thread 0: the twin's claim, once seen
  int tw_claim_snap = 0;
  int tw_loops0 = 0;               // twin: iterations the primary ran before the hand-over
  auto stamp = [&](int which, long long value = -1) __attribute__((always_inline)) {
    if (box && tw.debug) tw_store(box + kTwStamp + which, (unsigned long long)(value >= 0 ? value : wall_clock64()));
  };
  const long long tstart_clock = tw.debug ? wall_clock64() : 0;
  if (tid == 0 && !is_twin) stamp(kTsPStart);
  // report of this workgroup to the host (longest chain of iterations, units processed): both ways out of the kernel
  auto report = [&](int chain_loops, int units) __attribute__((always_inline)) {
    if (sweeps_out && tid == 0) {
      atomicMax(sweeps_out, chain_loops);      // longest chain of iterations of one instance inside this launch
      atomicAdd(sweeps_out + 1, units);        // (instance, iteration) units processed by this launch
      atomicMax(sweeps_out + 6, units + skipped);  // most iterations any ONE workgroup ran (a twin or its primary: their share)
      // (a column of a split streak that the sweeps handed over sits behind the batch; the tail launch carries
      //  the first shadow column in seg_lo and the columns per chain in seg_hi, which give the chain that owns it)
      int chain = 0;
      if (A.chain_size) {
        chain = (SEG && A.seg_end && b_real >= A.B && A.seg_hi > 0 && b_real >= A.seg_lo) ? (b_real - A.seg_lo) / A.seg_hi : b_real / A.chain_size;
        chain = chain < kMaxSweepChains - 1 ? chain : kMaxSweepChains - 1;
      }
      atomicMax(sweeps_out + 2, A.chain_base[chain] + chain_loops);  // ... counted from the first sweep of the solve
      if (kSoft && sync_words[kSyErr] != 0) atomicMax(sweeps_out + 3, 1);  // a wave gave up waiting for a sequence word
    }
  };
  if (is_twin) {
    // ---- find work: a slot whose primary has published a streak and that no other twin serves ----
    if (wave == 0) {
      const int cnt_slots = A.act_count ? *A.act_count : A.act_count_const;
      const int nsl = cnt_slots < tw.cap ? cnt_slots : tw.cap;
      int chosen = -1;
      for (int tries = 0; tries < kTwinIdlePolls && chosen < 0; ++tries) {
        int waiting = 0;  // primaries that may still offer something: alive (or not yet dispatched) and unclaimed
        for (int s0 = 0; s0 < nsl && chosen < 0; s0 += kBlock) {
          const int sc = s0 + lane;
          bool open = false, cand = false;
          if (sc < nsl) {
            const unsigned long long st = tw_load(tw.state + sc);
            open = (st & (kTwsClosed | kTwsLocked)) == 0;
            cand = st == kTwsSnap;
          }
          waiting += __popcll(__ballot(open));
          unsigned long long m = __ballot(cand);
          while (m != 0 && chosen < 0) {
            const int s2 = s0 + (__ffsll((long long)m) - 1);
            m &= m - 1;
            unsigned long long* bx = tw.box + (size_t)s2 * kTwWords;
            int got = 0;  // 0: lost the lock, 1: claimed, 2: locked but nothing to claim (kept locked), 3: snapshot gone (unlocked)
            if (lane == 0 && tw_cas(tw.state + s2, kTwsSnap, kTwsSnap | kTwsLocked)) {
              got = 3;
              unsigned long long cnt = 0, rb = 0, db = 0;
              bool have = false;
              for (int r = 0; r < 8 && !have; ++r) {
                const unsigned long long ver = tw_load(bx + kTwSeq);
                if (ver == 0) break;  // the streak broke meanwhile
                const unsigned long long* buf = bx + kTwSnap + 4 * (ver & 1);
                cnt = tw_load(buf);
                rb = tw_load(buf + 1);
                db = tw_load(buf + 2);
                const unsigned long long ver2 = tw_load(bx + kTwSeq);
                have = ver2 == ver || ver2 == ver + 1;  // (buffer ver & 1 is only rewritten by version ver + 2)
              }
              if (have) {
                const int it_in = (int)(cnt >> 32), it_tot = (int)(cnt & 0xffffffffull);
                // iterations until a cap ends the streak at the latest (ilqr.hpp:600-611)
                const int r1 = o.max_iterations_inner - it_in, r2 = o.max_iterations_total - it_tot;
                const int R = r1 < r2 ? r1 : r2;
                if (R < kTwinMinRemaining) {
                  got = 2;  // not worth a twin's start-up: stays locked, nobody else tries
                } else {
                  const int ahead = (R + tw.lag) / 2 + 1;  // iterations the primary keeps, counted from the snapshot
                  // the regularisation entering iteration `start`: every iteration in between runs its backward pass
                  // (DecreaseRegularization, ilqr.hpp:440) and rejects its line search (IncreaseRegularization, :550)
                  double rho = tw_dbl(rb), drho = tw_dbl(db);
                  for (int jj = 0; jj < ahead; ++jj) {
                    decrease_reg(o, &rho, &drho);
                    increase_reg(o, &rho, &drho);
                  }
                  tw_store(bx + kTwClaimRho, tw_bits(rho));
                  tw_store(bx + kTwClaimDrho, tw_bits(drho));
                  tw_store(bx + kTwClaimSnap, (unsigned long long)(unsigned)it_tot);
                  tw_order();
                  tw_store(bx + kTwClaim, ((unsigned long long)(unsigned)(it_in + ahead) << 32) | (unsigned long long)(unsigned)(it_tot + ahead));
                  ff[8] = (double)(it_in + ahead);
                  ff[9] = (double)(it_tot + ahead);
                  ff[10] = rho;
                  ff[11] = drho;
                  if (sweeps_out) atomicAdd(sweeps_out + 5, 1);  // claims of this launch
                  got = 1;
                }
              }
              if (got == 3) __hip_atomic_fetch_and(tw.state + s2, ~kTwsLocked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            got = __shfl(got, 0);
            if (got == 1) chosen = s2;
          }
        }
        if (chosen < 0) {
          if (waiting == 0) break;  // every primary has finished or has its twin
          __builtin_amdgcn_s_sleep(64);
        }
      }
      if (lane == 0) {
        ff[15] = (double)chosen;
        // (what the primary's workgroup stored before it published -- the trajectory of its last accepted step, the
        //  multipliers -- may sit in another XCD's L2: its release is matched by this acquire)
        if (chosen >= 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    __syncthreads();
    slot = (int)ff[15];
    if (slot < 0) return;
    box = tw.box + (size_t)slot * kTwWords;
    b_real = instance_of_slot(A, slot, 0);
    if (tid == 0) {
      stamp(kTsTStart, tstart_clock);
      stamp(kTsTGo);
    }
    // ---- clone the instance into the shadow column ----
    const int bT = tw.col0 + tslot;
    {
      using R_ = Rec<T, M::n, M::m>;
      for (int i = tid; i < (N + 1) * R_::nP; i += kThreads) {
        const int k = i / R_::nP, e = i - k * R_::nP;
        A.X[((size_t)(unsigned)k * Bp + (unsigned)bT) * R_::nP + e] = A.X[((size_t)(unsigned)k * Bp + (unsigned)b_real) * R_::nP + e];
      }
      for (int i = tid; i < N * R_::mP; i += kThreads) {
        const int k = i / R_::mP, e = i - k * R_::mP;
        A.U[((size_t)(unsigned)k * Bp + (unsigned)bT) * R_::mP + e] = A.U[((size_t)(unsigned)k * Bp + (unsigned)b_real) * R_::mP + e];
      }
      if (tid < R_::nP) A.x0[(size_t)bT * R_::nP + tid] = A.x0[(size_t)b_real * R_::nP + tid];
      auto column = [&](T* arr, int rows) __attribute__((always_inline)) {
        for (int r = tid; r < rows; r += kThreads) arr[(unsigned)r * Bp + (unsigned)bT] = arr[(unsigned)r * Bp + (unsigned)b_real];
      };
      column(A.costs, N + 1);
      column(A.lam, pd->total_rows);
      column(A.pen, pd->total_rows);
      column(A.cval, pd->total_rows);
      column(const_cast<T*>(A.ipool), pd->nslots);
      if (tid == 0) {
        // per-instance solver state: as the primary left it ...
        const double c0 = A.dV0[b_real], c1 = A.dV1[b_real], c2 = A.J0[b_real], c3 = A.initial_cost[b_real], c4 = A.cost_cur[b_real],
                     c5 = A.dJ[b_real], c6 = A.grad[b_real], c7 = A.viol[b_real], c8 = A.penmax[b_real], c9 = A.alpha[b_real],
                     c10 = A.z[b_real], c11 = A.reg_log[b_real];
        const int i0 = A.status[b_real], i1 = A.status_al[b_real], i2 = A.it_outer[b_real], i3 = A.phase[b_real],
                  i4 = A.need_init_cost[b_real];
        A.dV0[bT] = c0; A.dV1[bT] = c1; A.J0[bT] = c2; A.initial_cost[bT] = c3; A.cost_cur[bT] = c4;
        A.dJ[bT] = c5; A.grad[bT] = c6; A.viol[bT] = c7; A.penmax[bT] = c8; A.alpha[bT] = c9; A.z[bT] = c10; A.reg_log[bT] = c11;
        A.status[bT] = i0; A.status_al[bT] = i1; A.it_outer[bT] = i2; A.phase[bT] = i3; A.need_init_cost[bT] = i4;
        // ... but for what the rejected iterations up to `start` will have changed: counters, regularisation, and the
        // previous cost, which every iteration of a streak sets to the (unchanged) current one (solver_stats.cpp:54-66)
        if (SEG && A.seg_end) {
          A.seg_end[bT] = kSegNoEnd;
          A.seg_next[bT] = -1;
          A.seg_flag[bT] = 0;
          A.seg_streak[bT] = 2;
        }
        A.cost_prev[bT] = c4;
        A.it_inner[bT] = (int)ff[8];
        A.it_total[bT] = (int)ff[9];
        A.rho_reg[bT] = ff[10];
        A.drho[bT] = ff[11];
      }
    }
    b = bT;
    __syncthreads();
    if (tid == 0) stamp(kTsTCloned);
    if (tid == 0) ff[12] = tw_load(box + kTwHand) == 0 ? 1.0 : 0.0;  // (refused already -- the primary broke its streak or finished)
    __syncthreads();
    if (ff[12] == 0.0) return;
  }
  if (wave == 2 && lane == 0) {  // LDS mirrors of initial_cost / need_init_cost (read by this same lane: forward2_body)
    ff[4] = A.initial_cost[b];
    ff[5] = A.need_init_cost[b] ? 1.0 : 0.0;
  }
  for (;;) {
    const long long st_it = ALTRO_STAMP_T0();
    // ---- S: X, U, lambda, rho, parameters -> LDS (all threads).  Only once: phases 2 and 3 of the
    //      forward pass keep the LDS copies current from then on ----
    if (loops == 0) {
      forward2_stage<T, M>(A, pd, L, smem_raw, sPool, 1, 0, tid, kThreads, kKdNone, true, -1, b);
      __syncthreads();
    }
    // (the two per-instance scalars the running-cost wave needs after E: requested now, their memory latency -- two
    //  dependent round trips -- runs beside the expansions instead of behind them)
    // ---- E: expansions from the LDS block (unless they were computed AHEAD, beside phase 3 of the previous iteration:
    //      see the end of the loop) ----
    const CtxL<T> CE(A, b, sPool, sm + L.nX + L.nU + L.nKD + 2 * L.rowsP(), sm + L.nX + L.nU + L.nKD,
                     sm + L.nX + L.nU + L.nKD + L.rowsP());
    if (!e_done) {
      expansion_from_lds<T, M>(A, pd, CE, sm, sm + L.nX, sCost, b, tid, kThreads);
      if (adopt) ALTRO_STAMP_ADD(17 + wave, st_it);  // (17..20: each wave's own share of E)
      // drains the stores: the records are in L2 for the backward wave, the costs in LDS.  (A speculated iteration runs
      // no backward pass of its own, and the records it just rewrote are bit for bit the ones the fourth wave will read:
      // only the LDS traffic has to settle -- the two microseconds of store acknowledgements stay off the chain.)
      if (SPEC && adopt) lds_barrier(); else __syncthreads();
    }
    if (wave == 0 && adopt) ALTRO_STAMP_ADD(9, st_it);
    const long long st_b = ALTRO_STAMP_T0();

    // speculate only in a streak of rejections (ff: phase 3 of the previous iteration, rewritten by this one's): a
    // converging instance accepts its steps, and a recursion beside its forward pass would only slow that down
    const bool armed = SPEC != kSpecOff && loops > 0 && ff[0] != 0.0;
    if (SPEC && adopt) {
      const long long st_cp = ALTRO_STAMP_T0();
      // ---- B was run ahead (fourth wave) during the previous forward pass: take its results ----
      for (int i = tid; i < N * R::KP; i += kThreads) sKDf[i] = sKD2[i];
      if (tid == 0) {
        double h[6];
        h[0] = fh2[1]; h[1] = fh2[2]; h[2] = fh2[4]; h[3] = fh2[5]; h[4] = fh2[6];  // (the fourth wave's hand-over values)
        fh[1] = h[0];
        fh[2] = h[1];
        fh[4] = h[2];
        fh[5] = h[3];
        A.dV0[b] = h[0];
        A.dV1[b] = h[1];
        A.reg_log[b] = h[4];  // stats_.Log("reg", rho_)
        A.rho_reg[b] = h[2];
        A.drho[b] = h[3];
      }
      if (wave == 0) ALTRO_STAMP_ADD(16, st_cp);
    } else if (wave == 0) {
      // ---- B ----
      backward_mfma_body<T, M, false, true>(A, o, 0, lane, blockIdx.x, nullptr, sKDf, fused_junk, fh, 0.0, 0.0, nullptr, b);
      // (its own LDS writes of fh[4], fh[5]: program order)
    }
    // (the running cost J0 of the expansion step is summed by the auxiliary wave during the forward pass: aux_wave_run)
    if (SPEC && adopt) lds_barrier(); else __syncthreads();
    if (wave == 0 && adopt) ALTRO_STAMP_ADD(10, st_b);
    const long long st_f = ALTRO_STAMP_T0();
    const bool st_adopted = adopt;
    // ---- F ----
    spec.armed = armed;
    forward2_body<T, M, true, kSrcLds, CIRC, kSoft, false>(A, pdg, pd, o, mode, 0, 1, smem_raw, fh, active_flag, sCand, ff,
                                                    kWave4 ? &spec : nullptr, alpha_tab,
                                                    FwdSync<kSoft>{sync_words, loops * kFwdSeqStride}, sCost, fh,
                                                    sync_words + kSyEAhead0, persistent ? (kWave4 ? 3 : 2) : 0, loops + 1, b);
    ++loops;
    if (wave == 0 && st_adopted) ALTRO_STAMP_ADD(11, st_f);
    const long long st_x = ALTRO_STAMP_T0();
    // ---- E AHEAD.  Phase 3 runs on the cost wave alone; the other waves left the forward pass behind barrier V and
    //      would only wait for it.  If the line search rejected every trial, the trajectory and -- unless phase 3 ends
    //      the inner solve -- the multipliers are what they were, so the next iteration's expansions can be computed
    //      NOW, from the same LDS block, by the waves that are free: the same work the next iteration would do at its
    //      top (nothing is skipped), 6 000 cycles earlier.  If phase 3 does end the inner solve (dual / penalty update)
    //      or the instance, the result is discarded and E runs again as usual. ----
    //      Two things keep the state exactly the reference's: the constraint values c_ that an expansion step stores
    //      (ilqr.hpp:675 -> al_cost.hpp:264-274) go to an LDS scratch and are committed only when the expansions are the
    //      next iteration's -- phase 3's dual update reads the c_ the line search left (quirk Q6) --, and the cost wave
    //      holds its dual / penalty sweeps until these waves are through (forward_phase3), so that the records left in
    //      memory when the instance finishes are this iteration's own.
    if (persistent && wave != 1 && reinterpret_cast<const int*>(xch)[1] == 0) {
      constexpr int kAhead = kWave4 ? 3 : 2;                     // waves 0, 2 (, 3)
      const int wi = wave == 0 ? 0 : wave - 1;                   // 0, 1 (, 2)
      const CtxL<T> CA(A, b, sPool, sm + L.nX + L.nU + L.nKD + 2 * L.rowsP(), sm + L.nX + L.nU + L.nKD,
                       sm + L.nX + L.nU + L.nKD + L.rowsP(), sCvalAhead);
      expansion_from_lds<T, M>(A, pd, CA, sm, sm + L.nX, sCost, b, wi * kBlock + lane, kAhead * kBlock);
      if (lane == 0) FwdSync<true>::post(sync_words + kSyEAhead0 + wi, loops);  // (loops: already this iteration's number + 1)
      if (wave == 0 && st_adopted) ALTRO_STAMP_ADD(21, st_x);
    }
    // (phase 2's stores were drained by barrier V; what is in flight now are the scalars of phase 3, which only a
    //  backward pass of the next iteration would read from global memory: see the end of the loop)
    if (SPEC) lds_barrier(); else __syncthreads();
    if (wave == 0 && st_adopted) ALTRO_STAMP_ADD(12, st_x);
#ifdef ALTRO_STAMPS
    if (st_adopted) {
      ++spec_iters;
      if (wave == 0) ALTRO_STAMP_ADD(14, st_it);
    }
#endif
    if (!persistent || *active_flag == 0) break;
    // (the expansions ahead are the next iteration's iff the step was rejected and the inner solve goes on: then their
    //  constraint values become c_, as the expansion step at the top of the loop would have stored them)
    e_done = ff[0] != 0.0 && ff[3] == 0.0;
    if (e_done)
      for (int r = tid; r < pd->total_rows; r += kThreads) A.cval[(unsigned)r * Bp + (unsigned)b] = sCvalAhead[r];
    // the speculation holds if the line search rejected every trial, the inner solve goes on (no dual / penalty
    // update: ff[3]) and phase 3 set exactly the regularisation the speculative pass assumed
    if (kWave4) adopt = armed && fh2[7] != 0.0 && ff[0] != 0.0 && ff[3] == 0.0 && ff[1] == fh2[8] && ff[2] == fh2[9];
    if (o.fast_forward_stalls) {
      // OPT-IN, off by default.  A rejected line search leaves the trajectory, the multipliers and -- once
      // the regularisation has settled into its increase/decrease cycle -- the whole state of the instance
      // unchanged, so the next iteration recomputes exactly the same rejection: the reference (and the default
      // path here) repeats it until max_iterations_inner.  Two consecutive rejections with bit-identical
      // regularisation state prove the fixed point; the identical iterations in between are then only
      // counted (and logged), and the last one is executed normally so that every status decision is taken
      // by the usual code.
      const bool rej = ff[0] != 0.0 && ff[3] == 0.0;
      const bool stalled = rej && prev_rej && ff[1] == prev_rho && ff[2] == prev_drho;
      prev_rej = rej;
      prev_rho = ff[1];
      prev_drho = ff[2];
      if (stalled && tid == 0) {
        const int it_in = A.it_inner[b], it_tot = A.it_total[b];
        int k = o.max_iterations_inner - 1 - it_in;
        const int k2 = o.max_iterations_total - 1 - it_tot;
        k = k < k2 ? k : k2;
        if (k > 0) {
          A.it_inner[b] = it_in + k;
          A.it_total[b] = it_tot + k;
          for (int j = 0; j < k; ++j) hist_push(A, b);
          skipped += k;  // (kept apart from `loops`, which must stay wave-uniform)
        }
        // the cost wave of the next iteration reads these counters from global memory (load_inst_pre): another wave, and
        // the barrier below may be LDS-only -- drain this wave's stores first (with the expansions computed ahead there
        // is no expansion phase in between any more to hide the race)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    // ---- segments of rejection streaks (DevArrays::seg_*) that the batched sweeps handed over: a column with a successor
    //      verifies the successor's assumptions at its segment's end and retires, or cancels the chain; a cancelled column
    //      leaves.  (The batched kernels do this inside forward_phase3; here it sits beside the twins' bookkeeping: compiled
    //      into phase 3 it cost this kernel's register allocation 560 B of scratch per lane.)  The iteration that just ended
    //      is complete either way; a column that leaves does so through the loop's normal exit. ----
    bool seg_leave = false;
    if (SEG && A.seg_end) {
      if (tid == 0) {
        double leave = 0.0;
        const int flag = A.seg_flag[b];
        const int nxt = A.seg_next[b];
        const bool rc = ff[0] != 0.0 && ff[3] == 0.0;  // every trial rejected, and the inner solve goes on
        if (flag & kSegCancelled) {
          leave = 1.0;  // a predecessor did not arrive where this column assumed it would: its work is void
        } else if (nxt >= 0) {
          bool cancel = !rc;
          if (rc && (int)ff[6] == A.seg_end[b]) {
            const bool same = A.seg_tot0[nxt] == (int)ff[7] && tw_bits(A.seg_rho0[nxt]) == tw_bits(ff[1]) &&
                              tw_bits(A.seg_drho0[nxt]) == tw_bits(ff[2]) &&
                              (__hip_atomic_load(A.seg_flag + nxt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kSegCancelled) == 0;
            if (same) {
              A.seg_flag[b] = flag | kSegRetired;  // (the next column owns the instance from here: k_seg_fixup follows the chain)
              leave = 1.0;
            } else {
              cancel = true;
            }
          }
          if (cancel) {
            int sgd = nxt;
            for (int guard = 0; sgd >= 0 && guard < 256; ++guard) {
              atomicOr(A.seg_flag + sgd, kSegCancelled);
              sgd = A.seg_next[sgd];
            }
            A.seg_next[b] = -1;
            A.seg_end[b] = kSegNoEnd;
          }
        }
        if (leave != 0.0) A.phase[b] = 0;
        ff[12] = leave;
      }
      if (SPEC && adopt) lds_barrier(); else __syncthreads();
      seg_leave = ff[12] != 0.0;
      if (seg_leave) break;
    }
    // ---- twin workgroups (TwinCtl).  Primary: publish the state this iteration leaves while a streak is on, answer a
    //      claim when the iteration it names is reached.  Twin: a look at the verdict now and then. ----
    if (box) {
      if (!is_twin) {
        const bool rc = ff[0] != 0.0 && ff[3] == 0.0;  // every trial rejected, and the inner solve goes on
        tw_streak = rc ? tw_streak + 1 : 0;
        if (!rc) tw_break_total = (int)ff[7];
        if (tid == 0) {
          double act = 0.0;
          if (!tw_closed && (tw_streak >= 2 || tw_ver != 0)) {
            const int it_in = (int)ff[6], it_tot = (int)ff[7];  // the counters entering the next iteration
            if (tw_claim == 0) {
              // (a column whose streak the batched sweeps have split into segments ends at its segment's end: no twin)
              if (rc && tw_streak >= 2 && (!SEG || !A.seg_end || A.seg_next[b] < 0)) {
                ++tw_ver;
                unsigned long long* buf = box + kTwSnap + 4 * (tw_ver & 1);
                tw_store(buf, ((unsigned long long)(unsigned)it_in << 32) | (unsigned long long)(unsigned)it_tot);
                tw_store(buf + 1, tw_bits(ff[1]));
                tw_store(buf + 2, tw_bits(ff[2]));
                tw_store(buf + 3, (unsigned long long)(unsigned)loops);
                // once per streak: whatever the last accepted step and the first rejected iteration stored (trajectory,
                // multipliers, cost_prev = cost_cur) leaves this XCD's L2 before a twin is told that it may read it
                if (tw_streak == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                tw_order();
                tw_store(box + kTwSeq, tw_ver);
                if (!tw_shown) {
                  tw_order();
                  __hip_atomic_fetch_or(tw.state + slot, kTwsSnap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  tw_shown = true;
                }
                if (tw_ver == 1) {
                  stamp(kTsPSnap);
                  stamp(kTsPLoopsAtSnap, loops);
                }
              }
              // (a streak that broke: the snapshot on display describes a state that no longer exists -- a late twin would
              //  claim on it and be refused; hide it until the next streak publishes)
              if (!rc && tw_shown) {
                __hip_atomic_fetch_and(tw.state + slot, ~kTwsSnap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                tw_store(box + kTwSeq, 0ull);
                tw_shown = false;
              }
              tw_claim = tw_load(box + kTwClaim);
              if (tw_claim != 0) stamp(kTsPClaim);
              if (tw_claim != 0) {  // (its other words were stored before it; from now on nothing is published or polled)
                tw_claim_rho = tw_load(box + kTwClaimRho);
                tw_claim_drho = tw_load(box + kTwClaimDrho);
                tw_claim_snap = (int)tw_load(box + kTwClaimSnap);
              }
            }
            if (tw_claim != 0) {
              const int s_in = (int)(tw_claim >> 32), s_tot = (int)(tw_claim & 0xffffffffull);
              bool refuse = !rc || it_in > s_in;
              int why = !rc ? 1 : (it_in > s_in ? 2 : 0);
              if (!refuse && it_in == s_in) {
                // the twin's assumptions about the state entering this iteration, bit for bit -- and no iteration since
                // its snapshot that was anything but a rejected one
                const bool same = it_tot == s_tot && tw_bits(ff[1]) == tw_claim_rho && tw_bits(ff[2]) == tw_claim_drho &&
                                  tw_break_total < tw_claim_snap;
                if (same) act = 2.0; else refuse = true;
                why = it_tot != s_tot ? 3 : (tw_bits(ff[1]) != tw_claim_rho ? 4 : (tw_bits(ff[2]) != tw_claim_drho ? 5 : 6));
              }
              if (refuse) {
                tw_store(box + kTwWhy, (unsigned long long)why | ((unsigned long long)(unsigned)it_in << 8) | ((unsigned long long)(unsigned)s_in << 24));
                tw_cas(box + kTwHand, 0ull, kTwRefused);
                __hip_atomic_fetch_or(tw.state + slot, kTwsClosed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                tw_closed = true;
              }
            }
          }
          ff[13] = act;
        }
      } else if (tid == 0) {
        if (loops == 1) stamp(kTsTFirst);
        ff[13] = ((loops & 7) == 0 && tw_load(box + kTwHand) >= kTwRefused) ? 1.0 : 0.0;
      }
    }
    // (waves of a workgroup share the CU's vector L1, which stores write through and keep coherent,
    // so what this iteration wrote -- trajectory, multipliers, records -- is what the next one reads)
    // everyone has read the flag before phase 3 of the next iteration rewrites it; without a speculated backward pass
    // the next iteration reads this one's scalars from global memory: drain the stores
    if (SPEC && adopt) lds_barrier(); else __syncthreads();
    if (box) {
      const double act = ff[13];
      if (act == 1.0) return;  // twin: the primary refused (or finished) -- nothing of the clone is visible outside its column
      if (act == 2.0) {
        // HAND-OVER.  Everything this workgroup has stored leaves its L2 before the twin -- which will copy its column
        // over the instance's -- is told so: the two workgroups may sit on different XCDs, whose L2s write back on their own.
        __syncthreads();
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          tw_store(box + kTwHandLoops, (unsigned long long)(unsigned)loops);
          tw_order();
          ff[13] = tw_cas(box + kTwHand, 0ull, kTwOk) ? 3.0 : 0.0;  // (lost against a twin that gave up: go on alone)
          __hip_atomic_fetch_or(tw.state + slot, kTwsClosed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tw_closed = true;
          stamp(kTsPHand);
        }
        __syncthreads();
        if (ff[13] == 3.0) {
          report(loops + skipped, loops);
          return;  // the twin owns the instance: its copy-back is the instance's state
        }
      }
    }
  }
  // ---- a twin commits: the primary's verdict, then the shadow column over the instance's own ----
  if (is_twin) {
    if (tid == 0) {
      stamp(kTsTDone);
      stamp(kTsTLoops, loops);
      unsigned long long h = 0;
      for (int tries = 0; tries < kTwinHandPolls && (h = tw_load(box + kTwHand)) == 0; ++tries) __builtin_amdgcn_s_sleep(32);
      if (h == 0) h = tw_cas(box + kTwHand, 0ull, kTwRevoked) ? kTwRevoked : tw_load(box + kTwHand);
      if (h == kTwOk) {
        ff[14] = (double)(unsigned)tw_load(box + kTwHandLoops);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        stamp(kTsTVerdict);
      }
      ff[13] = h == kTwOk ? 1.0 : 0.0;
    }
    __syncthreads();
    if (ff[13] == 0.0) return;
    tw_loops0 = (int)ff[14];
    if (sweeps_out && tid == 0) atomicAdd(sweeps_out + 4, 1);  // hand-overs of this launch
    {
      using R_ = Rec<T, M::n, M::m>;
      using RS_ = rec_scalar_t<T, M>;
      using RR_ = Rec<RS_, M::n, M::m>;
      const int bT = b;
      for (int i = tid; i < (N + 1) * R_::nP; i += kThreads) {
        const int k = i / R_::nP, e = i - k * R_::nP;
        A.X[((size_t)(unsigned)k * Bp + (unsigned)b_real) * R_::nP + e] = A.X[((size_t)(unsigned)k * Bp + (unsigned)bT) * R_::nP + e];
      }
      for (int i = tid; i < N * R_::mP; i += kThreads) {
        const int k = i / R_::mP, e = i - k * R_::mP;
        A.U[((size_t)(unsigned)k * Bp + (unsigned)b_real) * R_::mP + e] = A.U[((size_t)(unsigned)k * Bp + (unsigned)bT) * R_::mP + e];
      }
      RS_* const E = (RS_*)A.EXP;
      for (int i = tid; i < (N + 1) * RR_::EP; i += kThreads) {
        const int k = i / RR_::EP, e = i - k * RR_::EP;
        E[((size_t)(unsigned)k * Bp + (unsigned)b_real) * RR_::EP + e] = E[((size_t)(unsigned)k * Bp + (unsigned)bT) * RR_::EP + e];
      }
      auto column = [&](T* arr, int rows) __attribute__((always_inline)) {
        for (int r = tid; r < rows; r += kThreads) arr[(unsigned)r * Bp + (unsigned)b_real] = arr[(unsigned)r * Bp + (unsigned)bT];
      };
      column(A.costs, N + 1);
      column(A.lam, pd->total_rows);
      column(A.pen, pd->total_rows);
      column(A.cval, pd->total_rows);
      if (tid == 0) {
        const double c0 = A.rho_reg[bT], c1 = A.drho[bT], c2 = A.dV0[bT], c3 = A.dV1[bT], c4 = A.J0[bT], c5 = A.initial_cost[bT],
                     c6 = A.cost_cur[bT], c7 = A.cost_prev[bT], c8 = A.dJ[bT], c9 = A.grad[bT], c10 = A.viol[bT], c11 = A.penmax[bT],
                     c12 = A.alpha[bT], c13 = A.z[bT], c14 = A.reg_log[bT];
        const int i0 = A.status[bT], i1 = A.status_al[bT], i2 = A.it_inner[bT], i3 = A.it_outer[bT], i4 = A.it_total[bT],
                  i5 = A.phase[bT], i6 = A.need_init_cost[bT];
        A.rho_reg[b_real] = c0; A.drho[b_real] = c1; A.dV0[b_real] = c2; A.dV1[b_real] = c3; A.J0[b_real] = c4;
        A.initial_cost[b_real] = c5; A.cost_cur[b_real] = c6; A.cost_prev[b_real] = c7; A.dJ[b_real] = c8; A.grad[b_real] = c9;
        A.viol[b_real] = c10; A.penmax[b_real] = c11; A.alpha[b_real] = c12; A.z[b_real] = c13; A.reg_log[b_real] = c14;
        A.status[b_real] = i0; A.status_al[b_real] = i1; A.it_inner[b_real] = i2; A.it_outer[b_real] = i3; A.it_total[b_real] = i4;
        A.phase[b_real] = i5; A.need_init_cost[b_real] = i6;
      }
    }
    b = b_real;  // (the gains below go to the instance's own records)
    if (tid == 0) stamp(kTsTCommit);
  } else if (box && tid == 0 && !tw_closed) {
    tw_cas(box + kTwHand, 0ull, kTwRefused);  // the instance is finished: a twin still waiting for a streak may leave
    __hip_atomic_fetch_or(tw.state + slot, kTwsClosed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stamp(kTsPEnd);
  }
#ifdef ALTRO_STAMPS
  __syncthreads();
  if (tid == 0 && spec_iters > 60 && loops > 100) {  // (the stragglers: a few dozen lines)
    const double c = 1.0 / spec_iters;
    const double ca = 1.0 / loops;  // the per-wave stamps run in every iteration
    printf("STAMPS (instance %d): %d iterations, %d adopted (speculated); per adopted iteration, shader-clock cycles:\n"
           "  E %.0f | take-over %.0f | forward pass (wave 0) %.0f | barrier behind F (wave 0 waits) %.0f | whole iteration %.0f\n"
           "  rollout wave: knot loop %.0f, wait A+S %.0f, phase 2 %.0f\n"
           "  cost wave: knot loop + wait A %.0f, selection %.0f, phase 2 + wait V %.0f, phase 3 %.0f\n"
           "  auxiliary wave: knot loop %.0f | fourth wave: speculative backward pass %.0f (only passes of armed iterations)\n"
           "  take-over: copy + hand-over (wave 0) %.0f, running cost (wave 1) %.0f | E alone, waves 0..3: %.0f %.0f %.0f %.0f | E ahead (wave 0) %.0f\n",
           b, loops, spec_iters, c * g_stamp_acc[9], c * g_stamp_acc[10], c * g_stamp_acc[11], c * g_stamp_acc[12], c * g_stamp_acc[14],
           ca * g_stamp_acc[0], ca * g_stamp_acc[1], ca * g_stamp_acc[2], ca * g_stamp_acc[3], ca * g_stamp_acc[4], ca * g_stamp_acc[5],
           ca * g_stamp_acc[6], ca * g_stamp_acc[7], c * g_stamp_acc[8], c * g_stamp_acc[16], c * g_stamp_acc[15],
           c * g_stamp_acc[17], c * g_stamp_acc[18], c * g_stamp_acc[19], c * g_stamp_acc[20], c * g_stamp_acc[21]);
  }
#endif
  // the gains for the getters (nothing inside the sweep reads them from global memory)
  for (int i = tid; i < N * R::KP; i += kThreads) {
    const int k = i / R::KP, e = i - k * R::KP;
    using RS = rec_scalar_t<T, M>;
    if (e < (Rec<RS, M::n, M::m>::KP)) RECP((RS*)A.KD, k, (Rec<RS, M::n, M::m>::KP))[e] = (RS)sKDf[i];
  }
  report(tw_loops0 + loops + skipped, loops);
}


// The last valid column of every chain of segments (DevArrays::seg_*) over the instance's own: one workgroup per instance,
// follows seg_next while the column it stands on has retired (= its successor started from exactly the state it arrived
// with), then copies that column's trajectory, multipliers, constraint values, records and solver state.
template <class T, class M>
__global__ __launch_bounds__(kBlock) void k_seg_fixup(DevArrays<T> A, const ProblemDesc* __restrict__ pd) {
  using R = Rec<T, M::n, M::m>;
  using RS = rec_scalar_t<T, M>;
  using RR = Rec<RS, M::n, M::m>;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= A.B) return;
  int cur = b;
  for (int guard = 0; guard < 4096; ++guard) {
    const int nx = A.seg_next[cur];
    if (!(A.seg_flag[cur] & kSegRetired) || nx < 0 || (A.seg_flag[nx] & kSegCancelled)) break;
    cur = nx;
  }
  if (cur == b) return;
  const unsigned Bp = A.Bp, src = (unsigned)cur, dst = (unsigned)b;
  const int N = A.N;
  for (int i = tid; i < (N + 1) * R::nP; i += kBlock) {
    const int k = i / R::nP, e = i - k * R::nP;
    A.X[((size_t)(unsigned)k * Bp + dst) * R::nP + e] = A.X[((size_t)(unsigned)k * Bp + src) * R::nP + e];
  }
  for (int i = tid; i < N * R::mP; i += kBlock) {
    const int k = i / R::mP, e = i - k * R::mP;
    A.U[((size_t)(unsigned)k * Bp + dst) * R::mP + e] = A.U[((size_t)(unsigned)k * Bp + src) * R::mP + e];
  }
  RS* const E = (RS*)A.EXP;
  for (int i = tid; i < (N + 1) * RR::EP; i += kBlock) {
    const int k = i / RR::EP, e = i - k * RR::EP;
    E[((size_t)(unsigned)k * Bp + dst) * RR::EP + e] = E[((size_t)(unsigned)k * Bp + src) * RR::EP + e];
  }
  RS* const G = (RS*)A.KD;
  for (int i = tid; i < N * RR::KP; i += kBlock) {
    const int k = i / RR::KP, e = i - k * RR::KP;
    G[((size_t)(unsigned)k * Bp + dst) * RR::KP + e] = G[((size_t)(unsigned)k * Bp + src) * RR::KP + e];
  }
  for (int r = tid; r <= N; r += kBlock) A.costs[(unsigned)r * Bp + dst] = A.costs[(unsigned)r * Bp + src];
  for (int r = tid; r < pd->total_rows; r += kBlock) {
    A.lam[(unsigned)r * Bp + dst] = A.lam[(unsigned)r * Bp + src];
    A.pen[(unsigned)r * Bp + dst] = A.pen[(unsigned)r * Bp + src];
    A.cval[(unsigned)r * Bp + dst] = A.cval[(unsigned)r * Bp + src];
  }
  if (tid == 0) {
    A.rho_reg[dst] = A.rho_reg[src]; A.drho[dst] = A.drho[src]; A.dV0[dst] = A.dV0[src]; A.dV1[dst] = A.dV1[src]; A.J0[dst] = A.J0[src];
    A.initial_cost[dst] = A.initial_cost[src]; A.cost_cur[dst] = A.cost_cur[src]; A.cost_prev[dst] = A.cost_prev[src];
    A.dJ[dst] = A.dJ[src]; A.grad[dst] = A.grad[src]; A.viol[dst] = A.viol[src]; A.penmax[dst] = A.penmax[src];
    A.alpha[dst] = A.alpha[src]; A.z[dst] = A.z[src]; A.reg_log[dst] = A.reg_log[src];
    A.status[dst] = A.status[src]; A.status_al[dst] = A.status_al[src]; A.it_inner[dst] = A.it_inner[src];
    A.it_outer[dst] = A.it_outer[src]; A.it_total[dst] = A.it_total[src]; A.phase[dst] = A.phase[src];
    A.need_init_cost[dst] = A.need_init_cost[src];
  }
}

// -------------------------------------------------------------------------------------------------
// THE DEVICE-SIDE SWEEP LOOP (round 6): the bulk phase of a batched solve without the host.
//
// What every instance runs is iLQR::Solve's loop (altro/ilqr/ilqr.hpp:300-313: UpdateExpansions, BackwardPass, ForwardPass,
// UpdateConvergenceStatistics until IsDone) inside the AL loop (al_solver.hpp:313-401) -- nothing in it needs the host, and
// nothing in it couples two instances.  Rounds 1 - 5 ran it as SWEEPS: three launches per iteration and chain of sweeps
// (k_expansions, k_backward_mfma, k_forward2), the host one sweep ahead, polling pinned counters to size the next grids.
// Here ONE launch of persistent workgroups does the same work: a workgroup (rollout / cost / auxiliary wave, the block of
// k_forward2) holds up to three instances in its SLOTS and runs E -> B -> F for them, iteration after iteration, with the
// bodies of the three kernels as they are -- expansion_body over the workgroup's threads, backward_mfma_body on wave 0 (three
// of the four 4 x 4 x 4 blocks carry an instance), forward2_body on all three waves -- separated by workgroup barriers
// instead of kernel boundaries: what a phase writes to global memory its successor reads through the CU's own L1, which the
// waves of a workgroup share (same argument as k_sweep_fused).  A slot whose instance has finished (the state machine of
// forward_phase3 clears DevArrays::phase) is refilled from a queue of not yet started instances -- one cursor per XCD over a
// contiguous eighth of the batch, so that the workgroups of an XCD work on neighbouring columns of the instance-minor
// arrays (see xcd_block); a workgroup whose range has run dry steals from the others.  No sweep boundary, no list rebuild,
// no counter for the host: instances advance at their own pace.
// HAND-OVER: when lc.handover or fewer instances of the batch are unfinished (started or not), every workgroup appends what
// it holds (and what is left in the queues) to the tail list and leaves; k_sweep_fused -- enqueued behind this launch by a
// host that never looked -- takes that list, one workgroup per straggler, twins and all.  Same device code on the same
// inputs in the same per-instance order: results are bit-identical to the host-paced sweeps
// (tests/test_fused_gpu.py::test_launch_variants_are_bit_identical, ALTRO_HIP_SWEEP_LOOP=0 restores the sweeps).
// -------------------------------------------------------------------------------------------------
enum LoopWord {
  kLwFinished = 0,  // instances that have left the solve inside this launch
  kLwTail = 1,      // length of the tail list
  kLwUnits = 2,     // (instance, iteration) units this launch ran
  kLwMaxLoops = 3,  // most iterations one workgroup ran
  kLwGroups = 4,    // workgroups that ran at least one iteration
  kLwTicks = 8,     // [4] 100 MHz ticks all workgroups spent in: slot bookkeeping, E, B, F (diagnostics: ALTRO_HIP_LOOP_LOG)
  kLwCursor = 16,   // [8] instances handed out of each XCD's range
  kLwWords = 24
};
struct LoopCtl {
  int* win;        // [gridDim.x][4]: the instances in the workgroup's slots this iteration (-1: empty; [3] is always empty)
  int* ctl;        // [kLwWords], zeroed by the host before the launch
  int* tail_list;  // [handover + slots of the launch]: what is handed to the persistent tail kernel
  int handover;    // unfinished instances of the batch at or below which the workgroups hand over and leave
  int per_xcd;     // instances per XCD range (a multiple of 16: the instances of one 128-byte line of the row arrays)
};
constexpr int kLoopXcds = 8;
#ifndef ALTRO_LOOP_BWD_AHEAD
#define ALTRO_LOOP_BWD_AHEAD 3
#endif
// prefetch depth of the backward pass inside the loop kernel: the stand-alone kernel's six knots in two ping-pong blocks are
// 120 of its 256 registers; here the workgroup shares its SIMDs with a second one (two waves per SIMD, 256 registers each)
constexpr int kLoopBwdAhead = ALTRO_LOOP_BWD_AHEAD;
#ifndef ALTRO_LOOP_MIN_BLOCKS
#define ALTRO_LOOP_MIN_BLOCKS 2
#endif
constexpr int kLoopMinBlocks = ALTRO_LOOP_MIN_BLOCKS;  // workgroups per CU the loop kernel is compiled for (registers: 512 / ceil(3 * blocks / 4) per lane)
// up to `want` not yet started instances for a workgroup of XCD `xcd`: own range first, then the others' (thread 0 only)
ALTRO_DEV int loop_pull(const LoopCtl& lc, int B, int xcd, int want, int* out) {
  int got = 0;
  for (int q = 0; q < kLoopXcds && got < want; ++q) {
    const int x = (xcd + q) % kLoopXcds;
    const int lo = x * lc.per_xcd, len = (B - lo < lc.per_xcd ? B - lo : lc.per_xcd);
    if (len <= 0) continue;
    int* cur = lc.ctl + kLwCursor + x;
    if (__hip_atomic_load(cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= len) continue;
    const int need = want - got;
    const int at = atomicAdd(cur, need);
    for (int j = 0; j < need && at + j < len; ++j) out[got++] = lo + at + j;
  }
  return got;
}
// The two big phases as functions of their own (A/B builds, -DALTRO_LOOP_NOINLINE_B / _F): the register allocator then
// treats the recursion's prefetch blocks and the forward pass's knot loops separately instead of spilling one inside the other
#ifdef ALTRO_LOOP_NOINLINE_B
#define ALTRO_LOOP_B_ATTR __device__ __attribute__((noinline))
#else
#define ALTRO_LOOP_B_ATTR ALTRO_DEV
#endif
#ifdef ALTRO_LOOP_NOINLINE_F
#define ALTRO_LOOP_F_ATTR __device__ __attribute__((noinline))
#else
#define ALTRO_LOOP_F_ATTR ALTRO_DEV
#endif
template <class T, class M>
ALTRO_LOOP_B_ATTR void loop_backward(const DevArrays<T>& Aw, const DevOpts& o, int lane, double* sKD) {
  backward_mfma_body<T, M, false, false, false, kLoopBwdAhead>(Aw, o, 0, lane, 0, sKD, nullptr, 0, nullptr);
}
template <class T, class M, int SRC>
ALTRO_LOOP_F_ATTR void loop_forward(const DevArrays<T>& Aw, const ProblemDesc* __restrict__ pdg, const ProblemDesc* pd, const DevOpts& o,
                                    int mode, int per_wave, unsigned char* smem_raw) {
  forward2_body<T, M, false, SRC>(Aw, pdg, pd, o, mode, 0, per_wave, smem_raw, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                  FwdSync<false>{nullptr, 0}, nullptr, nullptr, nullptr, 0, 0, -1, 0);
}
template <class T, class M, int SRC>
__global__ __launch_bounds__(kFwdWaves * kBlock, kLoopMinBlocks) void k_sweep_loop(DevArrays<T> A, const ProblemDesc* __restrict__ pdg,
                                                                        const ProblemDesc pd_arg, DevOpts o, int mode, LoopCtl lc) {
  constexpr int PW = kBlock / kLineSearchLanes;  // slots of a workgroup (instances per wave of the forward pass)
  static_assert(PW <= 3, "a window holds four entries, the fourth stays empty for the backward pass's fourth block");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __shared__ int s_slot[4];
  __shared__ int s_go;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int N = A.N;
  int* const win = lc.win + (size_t)blockIdx.x * 4;
  // what the bodies of the batched kernels see: a list of four entries -- this workgroup's window -- and nobody to append to
  DevArrays<T> Aw = A;
  Aw.act_list = win;
  Aw.act_count = nullptr;
  Aw.act_count_const = 4;
  Aw.next_list = nullptr;
  Aw.next_count = nullptr;
  Aw.host_count = nullptr;
  Aw.seg_end = nullptr;
  if (tid < 4) s_slot[tid] = -1;
  int loops = 0, units = 0;
  int ticks[4] = {0, 0, 0, 0};
  long long t_mark = wall_clock64();
  auto mark = [&](int which) __attribute__((always_inline)) {
    if (tid == 0) {
      const long long now = wall_clock64();
      ticks[which] += (int)(now - t_mark);
      t_mark = now;
    }
  };
  __syncthreads();
  for (;;) {
    if (tid == 0) {
      // ---- slots: drop what has finished, hand over or refill ----
      int fin = 0, held = 0;
      for (int g = 0; g < PW; ++g) {
        const int b = s_slot[g];
        if (b < 0) continue;
        if (A.phase[b] != 1) {
          s_slot[g] = -1;
          ++fin;
        } else {
          ++held;
        }
      }
      const int done = fin ? atomicAdd(lc.ctl + kLwFinished, fin) + fin
                           : __hip_atomic_load(lc.ctl + kLwFinished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int xcd = (int)(blockIdx.x % kLoopXcds);
      int go = 0;
      if (A.B - done <= lc.handover) {
        // the tail: whatever this workgroup holds, and whatever nobody has started yet, goes to the persistent kernel
        for (int g = 0; g < PW; ++g) {
          if (s_slot[g] >= 0) lc.tail_list[atomicAdd(lc.ctl + kLwTail, 1)] = s_slot[g];
          s_slot[g] = -1;
        }
        int b1;
        while (loop_pull(lc, A.B, xcd, 1, &b1) == 1) lc.tail_list[atomicAdd(lc.ctl + kLwTail, 1)] = b1;
      } else {
        if (held < PW) {
          int fresh[PW];
          const int got = loop_pull(lc, A.B, xcd, PW - held, fresh);
          for (int g = 0, j = 0; g < PW && j < got; ++g)
            if (s_slot[g] < 0) s_slot[g] = fresh[j++];
          held += got;
        }
        go = held > 0 ? 1 : 0;
        units += held;
      }
      for (int g = 0; g < 4; ++g) win[g] = s_slot[g];
      s_go = go;
    }
    __syncthreads();  // (s_waitcnt vmcnt(0) in front of the barrier: the window is in memory)
    if (!s_go) break;
    ++loops;
    mark(0);
    // ---- E: iLQR::UpdateExpansions (ilqr.hpp:670-677) of the slots' instances, one (instance, knot) per thread and round ----
    for (int u = tid; u < PW * (N + 1); u += kFwdWaves * kBlock) {
      const int g = u / (N + 1), k = u - g * (N + 1);
      const int b = s_slot[g];
      if (b >= 0) expansion_body<T, M>(A, pdg, b, k);
    }
    __syncthreads();
    mark(1);
    // ---- B: iLQR::BackwardPass (ilqr.hpp:385-445) on wave 0, one 4 x 4 x 4 block per slot ----
    if (wave == 0) loop_backward<T, M>(Aw, o, lane, reinterpret_cast<double*>(smem_raw));
    __syncthreads();
    mark(2);
    // ---- F: iLQR::ForwardPass + the state machine (ilqr.hpp:512-619, al_solver.hpp:313-401) on the three waves ----
    loop_forward<T, M, SRC>(Aw, pdg, &pd_arg, o, mode, PW, smem_raw);
    __syncthreads();
    mark(3);
  }
  if (tid == 0 && loops > 0) {
    for (int i = 0; i < 4; ++i) atomicAdd(lc.ctl + kLwTicks + i, ticks[i]);
    atomicAdd(lc.ctl + kLwUnits, units);
    atomicMax(lc.ctl + kLwMaxLoops, loops);
    atomicAdd(lc.ctl + kLwGroups, 1);
  }
}

// Keeps one wavefront busy for `ticks` of the constant 100 MHz clock: the engine checks with it that the streams of its
// chains of sweeps run side by side (streams that share a hardware queue run one after the other).
template <int kDummy>
__global__ __launch_bounds__(kBlock) void k_spin(long long ticks, long long* stamps) {
  const long long t0 = wall_clock64();
  long long t = t0;
  while (t - t0 < ticks) t = wall_clock64();
  if (stamps && threadIdx.x == 0) {  // start and end on the device's constant clock
    stamps[0] = t0;
    stamps[1] = t;
  }
}

// Concatenates the active lists the chains of batched sweeps leave behind into the list of the persistent kernel.
struct ChainLists {
  const int* list[kMaxSweepChains];
  const int* count[kMaxSweepChains];
  int n;
};
template <int kDummy>
__global__ __launch_bounds__(256) void k_merge_lists(ChainLists in, int* __restrict__ out_list, int* __restrict__ out_count) {
  int off = 0;
  for (int q = 0; q < in.n; ++q) {
    const int c = *in.count[q];
    for (int i = threadIdx.x; i < c; i += 256) out_list[off + i] = in.list[q][i];
    off += c;
  }
  if (threadIdx.x == 0) *out_count = off;
}

// The two layout conversions of the host boundary, on the device so that the host only copies contiguous buffers:
// device records [knots][Bp][EP] (fields off .. off+E) -> the caller's rows [B][knots][E] of doubles, and back
// (padding elements and padding instances zeroed; `per_instance` = 0: one [knots][E] block shared by the batch).
template <class E_>
__global__ __launch_bounds__(kBlock) void k_rec_to_rows(const E_* __restrict__ dev, double* __restrict__ out, int knots, int EP,
                                                        int off, int E, int B, int Bp) {
  const int b = blockIdx.x * kBlock + threadIdx.x, k = blockIdx.y;
  if (b >= B) return;
  const E_* src = dev + ((size_t)k * Bp + b) * EP + off;
  double* dst = out + ((size_t)b * knots + k) * E;
  for (int e = 0; e < E; ++e) dst[e] = (double)src[e];
}
template <class T>
__global__ __launch_bounds__(kBlock) void k_rows_to_rec(const double* __restrict__ src, T* __restrict__ dev, int knots, int EP,
                                                        int E, int B, int Bp, int per_instance) {
  const int b = blockIdx.x * kBlock + threadIdx.x, k = blockIdx.y;
  if (b >= Bp) return;
  T* dst = dev + ((size_t)k * Bp + b) * EP;
  const bool live = src != nullptr && b < B;
  const double* s0 = live ? src + ((per_instance ? (size_t)b * knots : 0) + k) * E : nullptr;
  for (int e = 0; e < EP; ++e) dst[e] = (live && e < E) ? T(s0[e]) : T(0);
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
