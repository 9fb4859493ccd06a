// altro_device.hpp — per-knot-point device math of the batched AL-iLQR solver (gfx950).
//
// Data layout in HBM: per-knot RECORDS, batch-minor at record granularity -- element e of knot k of
// instance b lives at arr[(k*Bp + b)*EP + e] (see struct Rec).  Constraint rows and per-instance
// scalars are plain struct-of-arrays [row][b].  All per-knot matrices are tiny (n<=12, m<=4) and
// live in VGPRs as fully unrolled arrays; nothing here touches scratch memory.
//
// Every function cites the reference code it replaces (paths relative to the reference root).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "altro_common.hpp"

namespace altro_hip {

#define ALTRO_DEV __device__ __forceinline__

// most chains of batched sweeps an engine runs side by side (Engine::Chain; four unless ALTRO_HIP_CHAINS asks for more)
constexpr int kMaxSweepChains = 8;

// -------------------------------------------------------------------------------------------------
// User-defined cost / constraint functors.  Only a user-model plugin (the translation
// unit altro_register_model_source generates) defines them -- its source is included BEFORE this header, in namespace
// altro_user -- and announces them with
//     #define ALTRO_USER_COSTS        CostA, CostB, ...      (or ALTRO_USER_COST X for a single one)
//     #define ALTRO_USER_CONSTRAINTS  ConA, ConB, ...        (or ALTRO_USER_CONSTRAINT X)
// Every knot of the reference's Problem may carry any CostFunction / Constraint subclass (problem.hpp:113-202); here a
// cost group / constraint descriptor carries the INDEX of its type in the list, and the device code dispatches on it
// (UserDispatch: an if-chain over the list, each arm instantiated for its own nparams / p).  Everywhere else the lists
// are empty and `if constexpr (kHasUser...)` removes every trace: the built-in engines compile exactly as without them.
// -------------------------------------------------------------------------------------------------
template <class... Fs>
struct UserTypeList {
  static constexpr int size = (int)sizeof...(Fs);
};
template <class F>
struct UserTag {
  using type = F;
};
template <class L>
struct UserDispatch;
template <>
struct UserDispatch<UserTypeList<>> {
  template <class Fn>
  __host__ __device__ __forceinline__ static void call(int, Fn&&) {}
};
template <class F0, class... Fs>
struct UserDispatch<UserTypeList<F0, Fs...>> {
  // fn(UserTag<F>{}) for the idx-th type of the list (nothing for an index out of range)
  template <class Fn>
  __host__ __device__ __forceinline__ static void call(int idx, Fn&& fn) {
    if (idx == 0) fn(UserTag<F0>{});
    else UserDispatch<UserTypeList<Fs...>>::call(idx - 1, fn);
  }
};
#if defined(ALTRO_USER_COSTS) || defined(ALTRO_USER_COST) || defined(ALTRO_USER_CONSTRAINTS) || defined(ALTRO_USER_CONSTRAINT)
namespace user_lists_ {
using namespace ::altro_user;  // (the names in the macros are the user's)
#if defined(ALTRO_USER_COSTS)
using Costs = UserTypeList<ALTRO_USER_COSTS>;
#elif defined(ALTRO_USER_COST)
using Costs = UserTypeList<ALTRO_USER_COST>;
#else
using Costs = UserTypeList<>;
#endif
#if defined(ALTRO_USER_CONSTRAINTS)
using Cons = UserTypeList<ALTRO_USER_CONSTRAINTS>;
#elif defined(ALTRO_USER_CONSTRAINT)
using Cons = UserTypeList<ALTRO_USER_CONSTRAINT>;
#else
using Cons = UserTypeList<>;
#endif
}  // namespace user_lists_
using UserCostList = user_lists_::Costs;
using UserConList = user_lists_::Cons;
#else
using UserCostList = UserTypeList<>;
using UserConList = UserTypeList<>;
#endif
constexpr bool kHasUserCost = UserCostList::size > 0;
constexpr bool kHasUserCon = UserConList::size > 0;
// what the host needs to know of the idx-th user type (parameter count, rows, cone); -1 for an index out of range
inline int UserCostParams(int idx) {
  int np = -1;
  UserDispatch<UserCostList>::call(idx, [&](auto tag) { np = decltype(tag)::type::nparams; });
  return np;
}
inline void UserConInfo(int idx, int* nparams, int* p, int* equality) {
  *nparams = *p = *equality = -1;
  UserDispatch<UserConList>::call(idx, [&](auto tag) {
    using F = typename decltype(tag)::type;
    *nparams = F::nparams;
    *p = F::p;
    *equality = F::equality ? 1 : 0;
  });
}

// -------------------------------------------------------------------------------------------------
// Device array bundle (passed to kernels by value)
// -------------------------------------------------------------------------------------------------
// Record layout.  Per-knot data of one instance is a contiguous, 16-byte aligned RECORD; records of
// the 64 instances a wavefront owns are adjacent: arr[(k*Bp + b)*EP + e].  A lane therefore reads
// its record with a few 16-byte vector loads whose only varying address part is a scalar (k) --
// no per-element address arithmetic on the serial critical path -- and a wave touches one
// contiguous 64*EP*sizeof(T) block (fully coalesced).
template <class T, int n, int m>
struct Rec {
  static constexpr int V = 16 / (int)sizeof(T);  // elements per 16-byte vector
  static constexpr int pad(int e) { return (e + V - 1) / V * V; }
  static constexpr int nP = pad(n), mP = pad(m);  // X / U records
  // expansion record: [A|B] | lxx | lxu | luu | lx | lu
  static constexpr int oAB = 0, oLxx = n * (n + m), oLxu = oLxx + n * n, oLuu = oLxu + n * m,
                       oLx = oLuu + m * m, oLu = oLx + n, eE = oLu + m, EP = pad(eE);
  // gain record: K (m x n, column-major) | d
  static constexpr int oK = 0, oD = m * n, KP = pad(m * n + m);
  // cost-to-go record: P | p
  static constexpr int oP = 0, op = n * n, CP = pad(n * n + n);
};
template <class T>
struct VecOf;
template <>
struct VecOf<double> {
  using type = double2;
};
template <>
struct VecOf<float> {
  using type = float4;
};
// 16-byte vector copies between a record (global / LDS) and registers; EP is a padded length.
template <class T, int EP>
__device__ __forceinline__ void load_rec(const T* p, T* out) {
  using V = typename VecOf<T>::type;
  constexpr int VN = 16 / (int)sizeof(T);
  const V* pv = reinterpret_cast<const V*>(p);
#pragma unroll
  for (int i = 0; i < EP / VN; ++i) {
    const V v = pv[i];
    const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
    for (int j = 0; j < VN; ++j) out[i * VN + j] = e[j];
  }
}
template <class T, int EP>
__device__ __forceinline__ void store_rec(T* p, const T* in) {
  using V = typename VecOf<T>::type;
  constexpr int VN = 16 / (int)sizeof(T);
  V* pv = reinterpret_cast<V*>(p);
#pragma unroll
  for (int i = 0; i < EP / VN; ++i) {
    V v;
    T* e = reinterpret_cast<T*>(&v);
#pragma unroll
    for (int j = 0; j < VN; ++j) e[j] = in[i * VN + j];
    pv[i] = v;
  }
}

// -------------------------------------------------------------------------------------------------
// Record storage type.  An engine is instantiated for (T, M): T is the type of the solver state and of
// ALL arithmetic (trajectory, multipliers, costs, Riccati recursion, rollouts); M is the model, optionally
// wrapped in WithRec32<>, which keeps the two BULK per-knot records -- the expansion record and the gain
// record, 94 of the 127 elements a knot moves per iteration for the unicycle -- in fp32 in HBM.  That is the
// ALTRO_F32 dtype of the C-ABI: fp32 where the bytes are, fp64 where differences of nearly equal numbers
// decide the schedule (dJ < 1e-4 on J ~ 1e2, z = (J0 - J) / expected, c of an active constraint).  An
// all-fp32 solver loses 10-15 points of solved fraction on the obstacle problems and cannot meet the stated
// 1e-3 state tolerance on the 12-state model; this one reproduces the fp64 schedule (scripts/cpu_fp32_study.py).
// -------------------------------------------------------------------------------------------------
template <class M>
struct WithRec32 : M {
  using rec_t = float;
};
template <class T, class M, class = void>
struct RecScalar {
  using type = T;
};
template <class T, class M>
struct RecScalar<T, M, std::void_t<typename M::rec_t>> {
  using type = typename M::rec_t;
};
template <class T, class M>
using rec_scalar_t = typename RecScalar<T, M>::type;

// A record stored as RS in memory <-> T registers (Rec<T> register layout; the element offsets of Rec<T>
// and Rec<RS> coincide, only the padding at the end differs).  E = number of meaningful elements.
template <class T, class RS, int EPT, int EPS, int E>
__device__ __forceinline__ void load_rec_as(const RS* p, T* out) {
  if constexpr (std::is_same<T, RS>::value) {
    load_rec<T, EPT>(p, out);
  } else {
    RS tmp[EPS];
    load_rec<RS, EPS>(p, tmp);
#pragma unroll
    for (int e = 0; e < EPT; ++e) out[e] = e < E ? (T)tmp[e < E ? e : 0] : T(0);
  }
}
template <class T, class RS, int EPT, int EPS, int E>
__device__ __forceinline__ void store_rec_as(RS* p, const T* in) {
  if constexpr (std::is_same<T, RS>::value) {
    store_rec<T, EPT>(p, in);
  } else {
    RS tmp[EPS];
#pragma unroll
    for (int e = 0; e < EPS; ++e) tmp[e] = e < E ? (RS)in[e < E ? e : 0] : RS(0);
    store_rec<RS, EPS>(p, tmp);
  }
}

// -------------------------------------------------------------------------------------------------
// Device array bundle (passed to kernels by value)
// -------------------------------------------------------------------------------------------------
template <class T>
struct DevArrays {
  int B, Bp, N;
  // trajectory (Z_) records X[k][b][nP], U[k][b][mP]; initial state x0[b][nP]
  T *x0, *X, *U;
  // expansion records EXP[k][b][EP] (dynamics Jacobian + cost expansion) and gain records KD[k][b][KP],
  // stored as rec_scalar_t<T, M> (T, or float under WithRec32<M>): typed by the kernels, which know M
  void *EXP, *KD;
  // per-knot cost costs[k][b]; cost-to-go records CTG[k][b][CP] (written only on request)
  T *costs, *CTG;
  // line-search candidates, instance-major [b][k][trial][x|u] (Zbar_ of every speculative trial)
  T* trial;
  // constraint rows [row][b]: duals, penalties, stored constraint values (c_)
  T *lam, *pen, *cval;
  // parameters
  const T *pool, *ipool;
  const int *knot_class, *knot_rowbase;
  const double* phi;  // penalty scaling per (class, constraint): phi[cls*kMaxConPerKnot + c]
  // per-instance solver state [field][b].  ALWAYS fp64, also when T = float: costs, expected
  // decrease, regularisation and the convergence statistics are differences of nearly equal
  // numbers (dJ < 1e-4 on J ~ 1e2; z = (J0 - J) / expected) that fp32 cannot resolve.
  double *rho_reg, *drho, *dV0, *dV1, *J0, *initial_cost, *cost_cur, *cost_prev, *dJ, *grad, *viol,
      *penmax, *alpha, *z, *reg_log;
  int *status, *status_al, *it_inner, *it_outer, *it_total, *phase, *need_init_cost;
  // active-instance compaction: this launch handles instances act_list[0 .. *act_count) (nullptr:
  // identity list of act_count_const entries); instances that keep iterating are appended to
  // next_list / next_count by the forward kernel's state machine
  const int* act_list;
  const int* act_count;
  int act_count_const;
  int* next_list;
  int* next_count;
  // host-visible (pinned, mapped) word that receives this launch's instance count: the host sizes the
  // next grid from it without a device-to-host copy in the stream (nullptr: not published)
  int* host_count;
  // optional per-iteration history [field][cap][Bp]
  double* hist;
  int* hist_len;
  int hist_cap;
  int record_ctg;
  // chains of sweeps (the engine runs the batched sweeps of a large batch as up to four independent chains, one stream
  // each): the instance range of this launch's chain for the dense mode of k_expansions (chain_hi = 0: the whole
  // batch), and for the persistent kernel the number of batched sweeps each chain ran before it (chain_size = 0: [0])
  int chain_lo, chain_hi;
  int chain_size;
  int chain_base[kMaxSweepChains];
  // per-knot steps h[k] (k < N) and times t[k] (k <= N), 32-bit floats like the reference's KnotPoint (knotpoint.hpp:
  // 179-180): nullptr for a uniform step and a time-invariant model (the hot kernels then use ProblemDesc::hstep);
  // set by altro_set_steps / altro_set_times or by a time-varying user model -- see Engine::knot_times_
  const float *hk, *tk;
  int xcd_remap;          // workgroup order of k_forward2 / k_backward_mfma: neighbouring instances on one XCD (xcd_block)
  int cand_front;         // k_forward2: leading line-search trials that own a candidate slot (CandLayout, altro_kernels.hpp)
  const int* knot_model;  // model of the source's ALTRO_USER_MODELS list that knot k uses (Problem::SetDynamics(model, k)); null: 0
  // SEGMENTS OF A REJECTION STREAK in the batched sweeps (round 5; forward_phase3, altro_kernels.hpp).  An instance whose
  // line search rejects every trial changes nothing but its regularisation (by a rule known in advance) and its counters, so
  // the state entering iteration j + L of such a streak is known at iteration j.  The state machine then SPLITS what is left
  // of the inner solve into seg_parts segments: the instance keeps the first, the others start in SHADOW COLUMNS -- clones of
  // the instance with counters and regularisation advanced -- that join the active list as instances of their own, so that
  // one sweep advances the streak by seg_parts iterations.  A column that reaches the end of its segment compares what it
  // holds with what its successor assumed (bit for bit) and retires; a mismatch, or anything but a rejected iteration on the
  // way, cancels the successors and the column goes on alone.  k_seg_fixup copies the last valid column of every chain back
  // over the instance's own.  Every iteration is computed with the inputs the sequential order gives it: same bits.
  // All arrays [Bp]; nullptr = feature off (ALTRO_HIP_SEGMENTS=0, small batches, a recorded history).
  int *seg_end;     // it_inner at which this column's segment ends (INT_MAX: it owns the rest)
  int *seg_next;    // shadow column of the next segment (-1: none)
  int *seg_flag;    // kSegCancelled | kSegRetired
  int *seg_streak;  // consecutive rejected iterations (inner solve going on) of this column
  int *seg_tot0;    // it_total / regularisation this column ASSUMED when it started: what its predecessor must arrive with
  double *seg_rho0, *seg_drho0;
  int* seg_cursor;     // next free column of this launch's slice of shadow columns (device counter of the chain)
  int seg_lo, seg_hi;  // that slice
  int seg_parts;       // segments a streak is split into (1: this sweep does not split -- the host sized the next sweep's grids
                       // for the count it knows, and only every kSegSplitEvery-th sweep may outgrow that)
};
constexpr int kSegSplitEvery = 4;
constexpr int kSegCancelled = 1, kSegRetired = 2;
constexpr int kSegNoEnd = 0x7fffffff;
constexpr int kSegMinRemaining = 24;  // iterations left in an inner solve below which a streak is not split
// step and time of knot k (k wave-uniform where it matters: scalar loads)
template <class T>
ALTRO_DEV T step_of(const DevArrays<T>& A, const ProblemDesc* pd, int k) {
  return A.hk ? T(A.hk[k]) : T(pd->hstep);
}
template <class T>
ALTRO_DEV float time_of(const DevArrays<T>& A, int k) {
  return A.tk ? A.tk[k] : 0.0f;
}
template <class T>
ALTRO_DEV int model_of(const DevArrays<T>& A, int k) {
  return A.knot_model ? A.knot_model[k] : 0;
}

template <class T>
ALTRO_DEV void sincos_(T x, T* s, T* c);
// fp64 sin/cos for the model dynamics.  ocml's sincos carries a double-double argument reduction
// (~250 instructions); the rollout evaluates 3 of them per knot on the serial critical path, so this
// is a short Cody-Waite reduction (exact for |x| < 1e5 thanks to FMA) + the fdlibm minimax kernels
// on [-pi/4, pi/4]: <= 1.5 ulp from the correctly rounded value, which is also what glibc (the CPU
// oracle) delivers, so results agree to ~1e-16.  Larger arguments take the ocml path.
// Keeps a value in its register across this point: the compiler can neither sink the computation that produced
// it into a conditional region nor replace a select on it by a branch.  On a lone wavefront a divergent region
// (s_and_saveexec + s_cbranch_execz + s_or exec) costs ~50 cycles even when every lane takes the same side.
ALTRO_DEV void pin(double& x) { asm volatile("" : "+v"(x)); }
ALTRO_DEV void pin(float& x) { asm volatile("" : "+v"(x)); }
ALTRO_DEV void pin(int& x) { asm volatile("" : "+v"(x)); }
// The same for a wave-uniform integer, kept in a SCALAR register: under scalar-register pressure the compiler
// re-loads values that came from the kernel arguments (s_load_dword + s_waitcnt: a memory round trip, exposed on
// a serial chain) instead of spilling them; a pinned value can only be spilled to a vector lane.
ALTRO_DEV void pin_s(int& x) { asm volatile("" : "+s"(x)); }

// d = a * b + c as the three-address v_fma_f64.  For a polynomial step p <- z * p + K with K a loop-invariant
// coefficient register the compiler otherwise picks the two-address v_fmac_f64 and pays a v_mov_b64 copy of K per
// step; it also runs the sine and the cosine polynomial one after the other, each step waiting ~12 cycles for the
// previous one.  fdlibm_sincos_kernels() below issues the two chains interleaved, step by step.
ALTRO_DEV double fma3(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// __kernel_sin / __kernel_cos of fdlibm on |r| <= pi/4 (r*r = z): the same operations as the plain C version in the
// same order per chain, hence the same bits.
ALTRO_DEV void fdlibm_sincos_kernels(double r, double* s, double* c) {
  const double z = r * r;
  double ps = fma3(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  double pc = fma3(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  ps = fma3(z, ps, 2.75573137070700676789e-06);
  pc = fma3(z, pc, -2.75573143513906633035e-07);
  ps = fma3(z, ps, -1.98412698298579493134e-04);
  pc = fma3(z, pc, 2.48015872894767294178e-05);
  ps = fma3(z, ps, 8.33333333332248946124e-03);
  pc = fma3(z, pc, -1.38888888888741095749e-03);
  ps = fma3(z, ps, -1.66666666666666324348e-01);
  pc = fma3(z, pc, 4.16666666666666019037e-02);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  *s = fma(r * z, ps, r);
  *c = w + (((1.0 - w) - hz) + z * (z * pc));
}

// The rare path of sincos_ (|x| >= 1e5: a line-search trial that blew up): ocml's sincos with its Payne-Hanek argument
// reduction, ~1 200 instructions and ~160 live registers per inlined copy (the rollout wave alone inlines six).  Round 5
// measured the out-of-line alternative at compile level (-DALTRO_SINCOS_BIG_NOINLINE, profiles/r05_spill_map.txt): the
// persistent kernel's vgpr_spill_count falls from 210 - 218 to 4 - 8 -- the spilled values live in these cold copies, not
// on the knot-to-knot chains -- but a call inside the kernel costs 40 B of scratch per lane and moves ~15 more SGPR
// spill reloads (v_readlane) per knot INTO the MFMA recursion.  Inline stays the default.
#ifdef ALTRO_SINCOS_BIG_NOINLINE
__device__ __attribute__((noinline)) inline void sincos_big(double x, double* s, double* c) { sincos(x, s, c); }
#else
ALTRO_DEV void sincos_big(double x, double* s, double* c) { sincos(x, s, c); }
#endif
template <>
ALTRO_DEV void sincos_<double>(double x, double* s, double* c) {
  // Arguments of 1e5 and beyond (a line-search trial that blew up) take ocml's path with its full argument
  // reduction.  The test is WAVE-UNIFORM (a ballot, i.e. a scalar branch): the fast path below is computed by every
  // lane, and only when some lane of the wave is out of range do those lanes recompute -- per lane the same values
  // as a divergent if / else, without a divergent region on the rollout's serial chain.
  const bool big = !(fabs(x) < 1.0e5);
  const double kq = rint(x * 6.36619772367581382433e-01);  // x * 2/pi
  // pi/2 split: hi has 53 bits, lo the next 53
  double r = fma(-kq, 1.57079632679489655800e+00, x);
  r = fma(-kq, 6.12323399573676603587e-17, r);
  double sr, cr;
  fdlibm_sincos_kernels(r, &sr, &cr);
  const int q = big ? 0 : ((int)kq & 3);
  const double s0 = (q & 1) ? cr : sr;
  const double c0 = (q & 1) ? sr : cr;
  double so = (q & 2) ? -s0 : s0;
  double co = ((q + 1) & 2) ? -c0 : c0;
  if (__builtin_expect(__ballot(big) != 0ull, 0)) {
    if (big) sincos_big(x, &so, &co);
  }
  *s = so;
  *c = co;
}
template <>
ALTRO_DEV void sincos_<float>(float x, float* s, float* c) {
  sincosf(x, s, c);
}
// sin/cos of a small argument (|x| < pi/4): the fdlibm kernels without range reduction.
template <class T>
ALTRO_DEV void sincos_small(T x, T* s, T* c);
template <>
ALTRO_DEV void sincos_small<double>(double r, double* s, double* c) {
  fdlibm_sincos_kernels(r, s, c);
}
template <>
ALTRO_DEV void sincos_small<float>(float r, float* s, float* c) {
  sincosf(r, s, c);
}
// x / 6: reciprocal multiply + one FMA correction step (Markstein).  With the correctly rounded
// reciprocal this returns the correctly rounded quotient, i.e. the same bits as the division the
// reference performs, in 3 instructions instead of the ~11 of an fp64 division.
template <class T>
ALTRO_DEV T div6(T a) {
  const T y = T(1) / T(6);
  const T q = a * y;
  const T r = fma(T(-6), q, a);
  return fma(r, y, q);
}

template <class T>
ALTRO_DEV T sqrt_(T x);
template <>
ALTRO_DEV double sqrt_<double>(double x) {
  return sqrt(x);
}
template <>
ALTRO_DEV float sqrt_<float>(float x) {
  return sqrtf(x);
}
// |x| as the free source modifier of the consuming instruction (a compare-and-select costs four instructions and
// a VCC hazard on a lone wave's chain); differs from `x < 0 ? -x : x` only in the sign of a zero result
ALTRO_DEV double abs_(double x) { return __builtin_fabs(x); }
ALTRO_DEV float abs_(float x) { return __builtin_fabsf(x); }
template <class T>
ALTRO_DEV T min_(T a, T b) {
  return b < a ? b : a;
}
template <class T>
ALTRO_DEV T max_(T a, T b) {
  return a < b ? b : a;
}

// -------------------------------------------------------------------------------------------------
// Continuous-time models (closed registry).  jac = [A|B], n x (n+m) column-major, fully written.
// -------------------------------------------------------------------------------------------------
struct UnicycleM {  // examples/unicycle.cpp:12-33
  static constexpr int n = 3, m = 2;
  static constexpr bool kHasFusedRk4 = true;
  static constexpr bool kHasCarriedTrig = true;
  static constexpr bool kHasFusedJacobian = true;
  // RK4 step with the duplicated work of the generic formula removed.  theta' = omega is constant
  // over the step, so the stage angles are theta, theta + (omega*0.5)*h (stages 2 AND 3: the two
  // expressions are the same floating-point computation) and theta + omega*h: 3 sincos instead of
  // 4.  Every other operation is performed exactly as rk4_step_generic does, in the same order.
  template <class T>
  static ALTRO_DEV void rk4_fused(const T* x, const T* u, T hh, T* xn) {
    T s1, c1;
    sincos_(x[2], &s1, &c1);
    rk4_fused_sc(x, u, hh, xn, s1, c1);
  }
  // The same step with sin / cos of theta supplied by the caller and sin / cos of the stage-4 angle
  // theta + omega*h handed back.  theta_{k+1} is that angle (up to the rounding of the RK4 combination,
  // ~1e-17 rad), so a rollout may carry (s4, c4) into the next step instead of evaluating a full sincos
  // on its serial chain; it re-evaluates every kTrigResync knots, which bounds the drift of the
  // rotations to ~3e-15.
  static constexpr int kTrigResync = 8;
  // EXACT = false (the line-search rollouts of the forward pass): the RK4 combination is evaluated in
  // factored form, x + (v h / 6)(c1 + 4 c2 + c4), theta + h w -- 8 operations instead of 33 on the
  // serial chain of the rollout wave; it differs from the reference's operation order by a few ulp of
  // the INCREMENT (~1e-17 absolute per step), far inside the stated tolerance.
  template <class T, bool EXACT = true>
  static ALTRO_DEV void rk4_fused_sc(const T* x, const T* u, T hh, T* xn, T& s1, T& c1) {
    const T v = u[0], w = u[1];
    T s2, c2, s4, c4;
    // stage angles theta + d2 and theta + d4 with d2 = (w*0.5)*h, d4 = w*h: |d| < pi/4 in any sane
    // rollout, so sin/cos of the stage angle come from the angle-addition formulas with the fdlibm
    // kernels evaluated directly on d (no range reduction, no quadrant selects): ~25 instructions
    // instead of ~57, accurate to ~2 ulp.  Larger steps take the full sincos.
    const T d2 = w * T(0.5) * hh, d4 = w * hh;
    {
      T sd, cd;
      sincos_small(d2, &sd, &cd);
      s2 = s1 * cd + c1 * sd;
      c2 = c1 * cd - s1 * sd;
      // d4 = 2 d2 exactly (the factor 0.5 is exact): double-angle formulas instead of a second kernel
      const T sdd = T(2) * sd * cd, cdd = fma(T(-2) * sd, sd, T(1));
      s4 = s1 * cdd + c1 * sdd;
      c4 = c1 * cdd - s1 * sdd;
    }
    // larger steps: the full sincos.  Wave-uniform test (see sincos_): the lanes in range keep the values above,
    // exactly what a per-lane if / else selects, and the common case has no divergent region on the chain.
    const bool wide = !(abs_(d4) < T(0.78));
    if (__builtin_expect(__ballot(wide) != 0ull, 0)) {
      if (wide) {
        sincos_(x[2] + d2, &s2, &c2);
        sincos_(x[2] + d4, &s4, &c4);
      }
    }
    if (EXACT) {
      const T k1x = v * c1, k1y = v * s1;
      const T k2x = v * c2, k2y = v * s2;  // k3 == k2: same stage angle
      const T k4x = v * c4, k4y = v * s4;
      xn[0] = x[0] + div6(hh * (k1x + 2 * k2x + 2 * k2x + k4x));
      xn[1] = x[1] + div6(hh * (k1y + 2 * k2y + 2 * k2y + k4y));
      xn[2] = x[2] + div6(hh * (w + 2 * w + 2 * w + w));
    } else {
      const T vh6 = v * (hh * T(1.0 / 6.0));
      xn[0] = fma(vh6, fma(T(4), c2, c1) + c4, x[0]);
      xn[1] = fma(vh6, fma(T(4), s2, s1) + s4, x[1]);
      xn[2] = fma(hh, w, x[2]);
    }
    s1 = s4;
    c1 = c4;
  }
  // RungeKutta4::Jacobian (integration.hpp:132-169) with the structural zeros of the unicycle removed.
  // The continuous Jacobians only have A[0][2] = -v sin, A[1][2] = v cos, B[0][0] = cos, B[1][0] = sin,
  // B[2][1] = 1 and depend on theta alone, whose stage values are theta, theta + w h / 2 (twice) and
  // theta + w h: 3 sincos instead of 7, ~40 flops instead of ~700, 7 exact divisions by 6 instead of 15
  // IEEE ones.  Every surviving operation is performed exactly as rk4_jacobian does, in the same order
  // (the dropped ones multiply or add exact zeros), so the result is the same bits.
  template <class T>
  static ALTRO_DEV void rk4_jac_fused(const T* x, const T* u, T hh, T* J) {
    const T v = u[0], w = u[1];
    const T th[4] = {x[2], x[2] + T(0.5) * w * hh, x[2] + T(0.5) * w * hh, x[2] + w * hh};
    T sn[4], cs[4];
    sincos_(th[0], &sn[0], &cs[0]);
    sincos_(th[1], &sn[1], &cs[1]);
    sn[2] = sn[1];
    cs[2] = cs[1];
    sincos_(th[3], &sn[3], &cs[3]);
    T sA02 = T(0), sA12 = T(0), sB00 = T(0), sB10 = T(0), sB01 = T(0), sB11 = T(0), sB21 = T(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const T a = -v * sn[s], bq = v * cs[s];  // A_s[0][2], A_s[1][2]
      const T coef = (s == 3) ? T(1) : T(0.5);
      const T wgt = (s == 0 || s == 3) ? T(1) : T(2);
      // dA_s = A_s (I + c dA_{s-1}) h: row 2 of dA is zero, so the bracket's [2][2] entry is exactly 1
      const T nA02 = a * hh, nA12 = bq * hh;
      // dB_s = B_s h + c A_s dB_{s-1} h: dB_{s-1}[2][1] = h exactly, dB_{s-1}[2][0] = 0
      const T nB00 = cs[s] * hh, nB10 = sn[s] * hh;
      const T nB01 = (s == 0) ? T(0) : coef * (a * hh) * hh;
      const T nB11 = (s == 0) ? T(0) : coef * (bq * hh) * hh;
      if (s == 0) {
        sA02 = nA02;
        sA12 = nA12;
        sB00 = nB00;
        sB10 = nB10;
        sB01 = nB01;
        sB11 = nB11;
        sB21 = hh;
      } else {
        sA02 = sA02 + wgt * nA02;
        sA12 = sA12 + wgt * nA12;
        sB00 = sB00 + wgt * nB00;
        sB10 = sB10 + wgt * nB10;
        sB01 = sB01 + wgt * nB01;
        sB11 = sB11 + wgt * nB11;
        sB21 = sB21 + wgt * hh;
      }
    }
#pragma unroll
    for (int i = 0; i < n * (n + m); ++i) J[i] = T(0);
    J[0 + 0 * n] = T(1);
    J[1 + 1 * n] = T(1);
    J[2 + 2 * n] = T(1);
    J[0 + 2 * n] = div6(sA02);
    J[1 + 2 * n] = div6(sA12);
    J[n * n + 0 + 0 * n] = div6(sB00);
    J[n * n + 1 + 0 * n] = div6(sB10);
    J[n * n + 0 + 1 * n] = div6(sB01);
    J[n * n + 1 + 1 * n] = div6(sB11);
    J[n * n + 2 + 1 * n] = div6(sB21);
  }
  template <class T>
  static ALTRO_DEV void f(const T* x, const T* u, T* xd) {
    T s, c;
    sincos_(x[2], &s, &c);
    xd[0] = u[0] * c;
    xd[1] = u[0] * s;
    xd[2] = u[1];
  }
  template <class T>
  static ALTRO_DEV void jac(const T* x, const T* u, T* J) {
#pragma unroll
    for (int i = 0; i < n * (n + m); ++i) J[i] = T(0);
    T s, c;
    sincos_(x[2], &s, &c);
    J[0 + 2 * n] = -u[0] * s;
    J[0 + 3 * n] = c;
    J[1 + 2 * n] = u[0] * c;
    J[1 + 3 * n] = s;
    J[2 + 4 * n] = T(1);
  }
};

template <int DOF>
struct TripleIntegratorM {  // examples/triple_integrator.cpp:9-33
  static constexpr int n = 3 * DOF, m = DOF;
  static constexpr bool kHasFusedRk4 = false;
  static constexpr bool kHasCarriedTrig = false;
  static constexpr bool kHasFusedJacobian = false;
  template <class T>
  static ALTRO_DEV void f(const T* x, const T* u, T* xd) {
#pragma unroll
    for (int i = 0; i < DOF; ++i) {
      xd[i] = x[i + DOF];
      xd[i + DOF] = x[i + 2 * DOF];
      xd[i + 2 * DOF] = u[i];
    }
  }
  template <class T>
  static ALTRO_DEV void jac(const T*, const T*, T* J) {
#pragma unroll
    for (int i = 0; i < n * (n + m); ++i) J[i] = T(0);
#pragma unroll
    for (int i = 0; i < DOF; ++i) {
      J[i + (i + DOF) * n] = T(1);
      J[(i + DOF) + (i + 2 * DOF) * n] = T(1);
      J[(i + 2 * DOF) + (i + 3 * DOF) * n] = T(1);
    }
  }
};

// Build-defined 12-state / 4-control model of BASELINE config 5 (no reference counterpart):
// x = (p, phi, v, w), u = (a, tau);  p' = v, phi' = w, v' = ((g+a) phi_y, -(g+a) phi_x, a), w' = tau.
struct Quadrotor12M {
  static constexpr int n = 12, m = 4;
  static constexpr bool kHasFusedRk4 = false;
  static constexpr bool kHasCarriedTrig = false;
  static constexpr bool kHasFusedJacobian = false;
  template <class T>
  static ALTRO_DEV void f(const T* x, const T* u, T* xd) {
    const T g = T(9.81);
    T a = u[0];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      xd[i] = x[6 + i];
      xd[3 + i] = x[9 + i];
      xd[9 + i] = u[1 + i];
    }
    xd[6] = (g + a) * x[4];
    xd[7] = -(g + a) * x[3];
    xd[8] = a;
  }
  template <class T>
  static ALTRO_DEV void jac(const T* x, const T* u, T* J) {
    const T g = T(9.81);
#pragma unroll
    for (int i = 0; i < n * (n + m); ++i) J[i] = T(0);
    T a = u[0];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      J[i + (6 + i) * n] = T(1);
      J[(3 + i) + (9 + i) * n] = T(1);
      J[(9 + i) + (n + 1 + i) * n] = T(1);
    }
    J[6 + 4 * n] = g + a;
    J[7 + 3 * n] = -(g + a);
    J[6 + n * n] = x[4];
    J[7 + n * n] = -x[3];
    J[8 + n * n] = T(1);
  }
};

// -------------------------------------------------------------------------------------------------
// RK4 (altro/problem/integration.hpp:123-169); h is a 32-bit float promoted to T (quirk Q1)
// -------------------------------------------------------------------------------------------------
// Time-varying dynamics (ContinuousDynamics::Evaluate(x, u, t, xdot), altro/problem/dynamics.hpp:59-95): a model that
// declares `static constexpr bool time_varying = true` takes the time as a 32-bit float, f(x, u, t, xdot) and
// jac(x, u, t, J); every other model keeps the three-argument form and never sees t.
template <class M, class = void>
struct model_time_varying : std::false_type {};
template <class M>
struct model_time_varying<M, std::void_t<decltype(M::time_varying)>> : std::integral_constant<bool, M::time_varying> {};
template <class T, class M>
ALTRO_DEV void model_f(const T* x, const T* u, float t, T* xd) {
  if constexpr (model_time_varying<M>::value) M::f(x, u, t, xd);
  else M::f(x, u, xd);
}
template <class T, class M>
ALTRO_DEV void model_jac(const T* x, const T* u, float t, T* J) {
  if constexpr (model_time_varying<M>::value) M::jac(x, u, t, J);
  else M::jac(x, u, J);
}
// stage times of RungeKutta4::Integrate (integration.hpp:126-129): `t + 0.5 * h` is evaluated in double (float
// operands, double literal) and narrowed to the float parameter of Evaluate
template <class T>
ALTRO_DEV float stage_time(float t, T hh, double c) {
  return (float)((double)t + c * (double)hh);
}

// CONTRACTION MODE OF THE GENERIC DYNAMICS CODE.  The library is compiled with -ffp-contract=fast: the optimiser may fuse
// a multiply and an add across statements, and WHETHER it does depends on what the function was inlined into -- so the
// same generic RK4 code could round differently in k_expansions and in the persistent kernel's expansion phase.  The
// hand-fused unicycle paths pin every fma; the generic integrators and the user's f / jac / step get the language-level
// rule instead (`contract(on)`: a * b + c fuses only inside ONE expression, wherever the function ends up), so that a
// model that can take both the batched kernels and the persistent kernel (n <= 3, m <= 2) sees one arithmetic.
#pragma clang fp contract(on)
// Generic RK4 step.  Models may provide `rk4_fused` (same arithmetic, shared sub-expressions).
template <class T, class M>
ALTRO_DEV void rk4_step_generic(const T* x, const T* u, T hh, T* xn, float t = 0.0f) {
  constexpr int n = M::n;
  T k1[n], k2[n], k3[n], k4[n], xt[n];
  model_f<T, M>(x, u, t, k1);
#pragma unroll
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k1[i] * T(0.5) * hh;
  model_f<T, M>(xt, u, stage_time(t, hh, 0.5), k2);
#pragma unroll
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k2[i] * T(0.5) * hh;
  model_f<T, M>(xt, u, stage_time(t, hh, 0.5), k3);
#pragma unroll
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k3[i] * hh;
  model_f<T, M>(xt, u, stage_time(t, hh, 1.0), k4);
#pragma unroll
  for (int i = 0; i < n; ++i) xn[i] = x[i] + hh * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) / 6;
}

template <class T, class M>
ALTRO_DEV void rk4_step(const T* x, const T* u, T hh, T* xn, float t = 0.0f) {
  if constexpr (M::kHasFusedRk4) {
    M::rk4_fused(x, u, hh, xn);
  } else {
    rk4_step_generic<T, M>(x, u, hh, xn, t);
  }
}

// RungeKutta4::Jacobian (integration.hpp:132-169).  The reference evaluates the two middle Jacobians at time 0.5 * t
// and the last one at t (not t + h): reproduced for time-varying models (integration.hpp:144-150).
//
// STRUCTURAL ZEROS.  The chain rule through the four stages is three products of n x n by n x (n + m) matrices; the
// continuous Jacobian of most models is sparse (the 12-state model: 14 entries of 192), and entries its jac() never
// writes are the literal 0 after inlining and unrolling.  IEEE arithmetic does not let the compiler drop `0 * x`
// (x might be infinite), so the dense code multiplied by them: 6 900 fp64 FMAs per knot for the 12-state model, most
// of its k_expansions time and 1.1 KB of scratch per lane.  ALTRO_SZ(a) is true when `a` is a COMPILE-TIME zero
// (__builtin_constant_p resolves after inlining, unrolling and constant propagation): those terms are skipped, and the
// zero pattern propagates through the stages (fill-in included).  Entries that are zero only at run time are
// multiplied as before; a model whose Jacobian is dense compiles to the same code as before.  Against the dense
// evaluation the result differs at most in the sign of a zero (and in not turning 0 * inf into NaN).
#define ALTRO_SZ(a) (__builtin_constant_p(a) && (a) == T(0))
template <class T, class M>
ALTRO_DEV void rk4_jacobian(const T* x, const T* u, T hh, T* J, float t = 0.0f) {
  constexpr int n = M::n, m = M::m, nm = n + m;
  if constexpr (M::kHasFusedJacobian) {
    M::rk4_jac_fused(x, u, hh, J);
    return;
  }
  T k1[n], k2[n], k3[n], xt[n];
  T Jc[n * nm];
  T dA[n * n], dB[n * m], sA[n * n], sB[n * m];
  model_f<T, M>(x, u, t, k1);
#pragma unroll
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k1[i] * T(0.5) * hh;
  model_f<T, M>(xt, u, stage_time(t, hh, 0.5), k2);
#pragma unroll
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k2[i] * T(0.5) * hh;
  model_f<T, M>(xt, u, stage_time(t, hh, 0.5), k3);
  // stage 0
  model_jac<T, M>(x, u, t, Jc);
#pragma unroll
  for (int e = 0; e < n * n; ++e) {
    dA[e] = ALTRO_SZ(Jc[e]) ? T(0) : Jc[e] * hh;
    sA[e] = dA[e];
  }
#pragma unroll
  for (int e = 0; e < n * m; ++e) {
    dB[e] = ALTRO_SZ(Jc[n * n + e]) ? T(0) : Jc[n * n + e] * hh;
    sB[e] = dB[e];
  }
  // stages 1..3: dA_s = A_s (I + c dA_{s-1}) h ; dB_s = B_s h + c A_s dB_{s-1} h
#pragma unroll
  for (int s = 1; s < 4; ++s) {
    const T coef = (s == 3) ? T(1) : T(0.5);
    const T wgt = (s == 3) ? T(1) : T(2);
    if (s == 1) {
#pragma unroll
      for (int i = 0; i < n; ++i) xt[i] = x[i] + T(0.5) * k1[i] * hh;
    } else if (s == 2) {
#pragma unroll
      for (int i = 0; i < n; ++i) xt[i] = x[i] + T(0.5) * k2[i] * hh;
    } else {
#pragma unroll
      for (int i = 0; i < n; ++i) xt[i] = x[i] + k3[i] * hh;
    }
    model_jac<T, M>(xt, u, (s == 3) ? t : (float)(0.5 * (double)t), Jc);
    T Mx[n * n], nA[n * n], nB[n * m];
#pragma unroll
    for (int j = 0; j < n; ++j)
#pragma unroll
      for (int i = 0; i < n; ++i) {
        const T di = (i == j ? T(1) : T(0));
        Mx[i + j * n] = ALTRO_SZ(dA[i + j * n]) ? di : di + coef * dA[i + j * n];
      }
#pragma unroll
    for (int j = 0; j < n; ++j)
#pragma unroll
      for (int i = 0; i < n; ++i) {
        T acc = T(0);
#pragma unroll
        for (int l = 0; l < n; ++l)
          if (!(ALTRO_SZ(Jc[i + l * n]) || ALTRO_SZ(Mx[l + j * n]))) acc += Jc[i + l * n] * Mx[l + j * n];
        nA[i + j * n] = ALTRO_SZ(acc) ? T(0) : acc * hh;
      }
#pragma unroll
    for (int j = 0; j < m; ++j)
#pragma unroll
      for (int i = 0; i < n; ++i) {
        T acc = T(0);
#pragma unroll
        for (int l = 0; l < n; ++l)
          if (!(ALTRO_SZ(Jc[i + l * n]) || ALTRO_SZ(dB[l + j * n]))) acc += Jc[i + l * n] * dB[l + j * n];
        const T jb = Jc[n * n + i + j * n];
        if (ALTRO_SZ(acc)) nB[i + j * n] = ALTRO_SZ(jb) ? T(0) : jb * hh;
        else if (ALTRO_SZ(jb)) nB[i + j * n] = coef * acc * hh;
        else nB[i + j * n] = jb * hh + coef * acc * hh;
      }
#pragma unroll
    for (int e = 0; e < n * n; ++e) {
      dA[e] = nA[e];
      if (!ALTRO_SZ(nA[e])) sA[e] = ALTRO_SZ(sA[e]) ? wgt * nA[e] : sA[e] + wgt * nA[e];
    }
#pragma unroll
    for (int e = 0; e < n * m; ++e) {
      dB[e] = nB[e];
      if (!ALTRO_SZ(nB[e])) sB[e] = ALTRO_SZ(sB[e]) ? wgt * nB[e] : sB[e] + wgt * nB[e];
    }
  }
#pragma unroll
  for (int j = 0; j < n; ++j)
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const T di = (i == j ? T(1) : T(0));
      J[i + j * n] = ALTRO_SZ(sA[i + j * n]) ? di : di + sA[i + j * n] / 6;
    }
#pragma unroll
  for (int e = 0; e < n * m; ++e) J[n * n + e] = ALTRO_SZ(sB[e]) ? T(0) : sB[e] / 6;
}
#undef ALTRO_SZ

// -------------------------------------------------------------------------------------------------
// What stands behind Problem::SetDynamics(model, k) (altro/problem/problem.hpp:155-166) for one knot: the reference
// stores a problem::DiscreteDynamics per knot, which is either
//   * DiscretizedModel<Model, RungeKutta4>   (discretized_model.hpp:24-65, integration.hpp:123-169) -- every built-in
//     model and the default for a user model (f, jac);
//   * DiscretizedModel<Model, ExplicitEuler> (integration.hpp:87-104): a user model that declares
//     `static constexpr int integrator = 1` (or a source compiled with -D / #define ALTRO_USER_INTEGRATOR 1);
//   * the caller's own DiscreteDynamics subclass (dynamics.hpp:148-187: Evaluate(x, u, t, h, xnext),
//     Jacobian(x, u, t, h, jac)): a user model that declares `static constexpr bool discrete = true` and defines
//     step(x, u, t, h, xnext) / step_jac(x, u, t, h, J) with 32-bit float t, h;
// and may be a different one on every knot: a source that lists several models of equal dimensions
// (#define ALTRO_USER_MODELS A, B -> struct with `using Models = UserTypeList<A, B>`) is dispatched on the knot's index
// (DevArrays::knot_model, altro_set_knot_models), each arm being any of the three kinds.
// -------------------------------------------------------------------------------------------------
#ifndef ALTRO_USER_INTEGRATOR
#define ALTRO_USER_INTEGRATOR 0
#endif
constexpr int kIntegratorRk4 = 0, kIntegratorEuler = 1;
template <class M, class = void>
struct model_discrete : std::false_type {};
template <class M>
struct model_discrete<M, std::void_t<decltype(M::discrete)>> : std::integral_constant<bool, M::discrete> {};
template <class M, class = void>
struct model_integrator : std::integral_constant<int, ALTRO_USER_INTEGRATOR> {};
template <class M>
struct model_integrator<M, std::void_t<decltype(M::integrator)>> : std::integral_constant<int, M::integrator> {};
template <class M, class = void>
struct model_list {
  using type = UserTypeList<M>;
  static constexpr bool is_list = false;
};
template <class M>
struct model_list<M, std::void_t<typename M::Models>> {
  using type = typename M::Models;
  static constexpr bool is_list = true;
};
// true when knots cannot share one (step, time, model): the engine then runs its general kernels (SetKnotTimes)
template <class L>
struct list_knot_path;
template <class... Ms>
struct list_knot_path<UserTypeList<Ms...>> {
  static constexpr bool value = ((model_time_varying<Ms>::value || model_discrete<Ms>::value) || ...);
};
template <class M>
struct model_knot_path {
  static constexpr bool value = model_list<M>::is_list || list_knot_path<typename model_list<M>::type>::value;
};

// one model S, one step: x_{k+1} = F(x_k, u_k, t_k, h_k)
template <class T, class S>
ALTRO_DEV void single_step(const T* x, const T* u, T hh, T* xn, float t) {
  if constexpr (model_discrete<S>::value) {
    S::step(x, u, t, (float)hh, xn);  // DiscreteDynamics::Evaluate (hh is the knot's 32-bit step, promoted: the narrowing is exact)
  } else if constexpr (model_integrator<S>::value == kIntegratorEuler) {
    // ExplicitEuler::Integrate (integration.hpp:90-94): Evaluate(x, u, t, xnext); xnext = x + xnext * h
    constexpr int n = S::n;
    T xd[n];
    model_f<T, S>(x, u, t, xd);
#pragma unroll
    for (int i = 0; i < n; ++i) xn[i] = x[i] + xd[i] * hh;
  } else {
    rk4_step<T, S>(x, u, hh, xn, t);
  }
}
// ... and its Jacobian [A | B], n x (n + m) column-major
template <class T, class S>
ALTRO_DEV void single_jacobian(const T* x, const T* u, T hh, T* J, float t) {
  if constexpr (model_discrete<S>::value) {
    S::step_jac(x, u, t, (float)hh, J);  // DiscreteDynamics::Jacobian
  } else if constexpr (model_integrator<S>::value == kIntegratorEuler) {
    // ExplicitEuler::Jacobian (integration.hpp:95-101): jac = Identity(n, n + m) + jac * h
    constexpr int n = S::n, m = S::m;
    model_jac<T, S>(x, u, t, J);
#pragma unroll
    for (int j = 0; j < n + m; ++j)
#pragma unroll
      for (int i = 0; i < n; ++i) J[i + j * n] = (i == j ? T(1) : T(0)) + J[i + j * n] * hh;
  } else {
    rk4_jacobian<T, S>(x, u, hh, J, t);
  }
}
// the discrete dynamics of knot k of model M (`which` = the knot's index into M's list of models; 0 without a list)
template <class T, class M>
ALTRO_DEV void discrete_step(const T* x, const T* u, T hh, T* xn, float t = 0.0f, int which = 0) {
  if constexpr (model_list<M>::is_list) {
    UserDispatch<typename model_list<M>::type>::call(which, [&](auto tag) { single_step<T, typename decltype(tag)::type>(x, u, hh, xn, t); });
  } else {
    single_step<T, M>(x, u, hh, xn, t);
  }
}
template <class T, class M>
ALTRO_DEV void discrete_jacobian(const T* x, const T* u, T hh, T* J, float t = 0.0f, int which = 0) {
  if constexpr (model_list<M>::is_list) {
    UserDispatch<typename model_list<M>::type>::call(which, [&](auto tag) { single_jacobian<T, typename decltype(tag)::type>(x, u, hh, J, t); });
  } else {
    single_jacobian<T, M>(x, u, hh, J, t);
  }
}

#pragma clang fp contract(fast)

// -------------------------------------------------------------------------------------------------
// Per-knot problem access
// -------------------------------------------------------------------------------------------------
// Per-knot data access.  Two interchangeable contexts:
//   CtxG  reads duals / penalties / parameters from the global SoA arrays (parallel kernels),
//   CtxL  reads them from the copy the forward pass staged in LDS (serial rollout loop: no VMEM
//         load may sit behind the candidate stores, and every pointer is an LDS pointer so that
//         the compiler emits ds_read, never flat_load).
template <class T>
struct CtxG {
  const DevArrays<T>& A;
  unsigned b;
  ALTRO_DEV CtxG(const DevArrays<T>& A_, int b_) : A(A_), b((unsigned)b_) {}
  ALTRO_DEV T par(int per_instance, int off, int i) const {
    return per_instance ? A.ipool[(unsigned)(off + i) * (unsigned)A.Bp + b] : A.pool[off + i];
  }
  ALTRO_DEV T shared(int off) const { return A.pool[off]; }
  ALTRO_DEV T lam(int r) const { return A.lam[(unsigned)r * (unsigned)A.Bp + b]; }
  ALTRO_DEV T pen(int r) const { return A.pen[(unsigned)r * (unsigned)A.Bp + b]; }
  ALTRO_DEV void store_c(int r, T c) const { A.cval[(unsigned)r * (unsigned)A.Bp + b] = c; }
};
template <class T>
struct CtxL {
  const DevArrays<T>& A;
  unsigned b;
  const T* sPool;  // shared parameters (LDS, one copy per wave)
  const T* sIp;    // per-instance parameter slots (LDS)
  const T* sLam;   // duals (LDS)
  const T* sPen;   // penalties (LDS)
  T* cdst;         // where store_c puts the constraint values: nullptr = the global c_ rows; otherwise an LDS scratch
                   // [row] (the expansions computed AHEAD by the persistent kernel, committed later or dropped)
  ALTRO_DEV CtxL(const DevArrays<T>& A_, int b_, const T* pool_, const T* ip_, const T* lam_, const T* pen_, T* cdst_ = nullptr)
      : A(A_), b((unsigned)b_), sPool(pool_), sIp(ip_), sLam(lam_), sPen(pen_), cdst(cdst_) {}
  ALTRO_DEV T par(int per_instance, int off, int i) const {
    const T* base = per_instance ? sIp : sPool;  // both LDS
    return base[off + i];
  }
  ALTRO_DEV T shared(int off) const { return sPool[off]; }
  ALTRO_DEV T lam(int r) const { return sLam[r]; }
  ALTRO_DEV T pen(int r) const { return sPen[r]; }
  ALTRO_DEV void store_c(int r, T c) const {
    if (cdst) cdst[r] = c;
    else A.cval[(unsigned)r * (unsigned)A.Bp + b] = c;
  }
};

// Dual-cone projection and its (diagonal) Jacobian (altro/constraints/constraint.hpp:70-78 for
// equalities, :103-114 for inequalities; quirk Q5: v == 0 counts as active).
template <class T>
ALTRO_DEV T dual_proj(int type, T v) {
  return type == 0 ? v : min_(T(0), v);
}
template <class T>
ALTRO_DEV T dual_proj_jac(int type, T v) {
  return type == 0 ? T(1) : (v > T(0) ? T(0) : T(1));
}
// CircleConstraint::Evaluate (examples/obstacle_constraints.hpp:99-107): c = r^2 - |p - centre|^2, in ONE operation
// order.  With plain products the contraction the compiler picks for dx*dx + dy*dy - r*r depends on what it can
// hoist out of the surrounding loop, and every evaluation of the constraint (trial cost, stored c_, expansion; the
// persistent kernel and the batched sweeps) must give the same bits.
template <class T>
ALTRO_DEV T circle_value(T dx, T dy, T rr) {
  return -fma(dx, dx, fma(dy, dy, -(rr * rr)));
}
// c - Pi_K(c): |c| for equalities, max(c, 0) for inequalities (constraint_values.hpp:215-220)
template <class T>
ALTRO_DEV T violation(int type, T c) {
  return type == 0 ? abs_(c) : abs_(c - min_(T(0), c));
}

// QuadraticCost::Evaluate (examples/quadratic_cost.cpp:8-11); H == 0 for LQRCost.  When Q / R are
// diagonal the exact-zero off-diagonal products are skipped (they contribute +0.0 to every sum).
// The parameters of a user functor, read through the context (shared pool or per-instance slots)
template <class T, int NP, class Ctx>
ALTRO_DEV void load_user_params(const Ctx& C, int per_instance, int off, T* par) {
#pragma unroll
  for (int i = 0; i < NP; ++i) par[i] = C.par(per_instance, off, i);
}
// CostFunction::Evaluate of the user's cost (costfunction.hpp:52-58)
template <class T, class Ctx>
ALTRO_DEV T user_cost_eval(const Ctx& C, const CostGroupDesc& g, const T* x, const T* u) {
  T J = T(0);
  UserDispatch<UserCostList>::call(g.user - 1, [&](auto tag) {  // (g.user = 1 + index of the type)
    using F = typename decltype(tag)::type;
    constexpr int NP = F::nparams;
    T par[NP > 0 ? NP : 1];
    load_user_params<T, NP>(C, g.u_pi, g.u_off, par);
    J = F::eval(x, u, par);
  });
  return J;
}

template <class T, int n, int m, class Ctx>
ALTRO_DEV T quad_cost(const Ctx& C, const CostGroupDesc& g, const T* x, const T* u) {
  if constexpr (kHasUserCost) {
    if (g.user) return user_cost_eval<T>(C, g, x, u);
  }
  T xQx = T(0), uRu = T(0), qx = T(0), ru = T(0);
#pragma unroll
  for (int i = 0; i < n; ++i) {
    T s = T(0);
    if (g.q_diag) {
      s = C.shared(g.Q_off + i + i * n) * x[i];
    } else {
#pragma unroll
      for (int j = 0; j < n; ++j) s += C.shared(g.Q_off + i + j * n) * x[j];
    }
    xQx += x[i] * s;
    qx += C.par(g.q_pi, g.q_off, i) * x[i];
  }
#pragma unroll
  for (int i = 0; i < m; ++i) {
    T s = T(0);
    if (g.r_diag) {
      s = C.shared(g.R_off + i + i * m) * u[i];
    } else {
#pragma unroll
      for (int j = 0; j < m; ++j) s += C.shared(g.R_off + i + j * m) * u[j];
    }
    uRu += u[i] * s;
    ru += C.par(g.r_pi, g.r_off, i) * u[i];
  }
  return T(0.5) * xQx + T(0.5) * uRu + qx + ru + C.par(g.c_pi, g.c_off, 0);
}

// ConstraintValues::AugLag (constraint_values.hpp:111-119) for the user's constraint: con_->Evaluate, then the
// projected multipliers.  STORE: also c_ and the violation, as for the built-in kinds.
template <class T, bool STORE, class Ctx>
ALTRO_DEV void user_con_auglag(const Ctx& C, const ConDesc& cd, int r0, T rho, const T* x, const T* u, T& a, T& bsum,
                               T& vmax) {
  UserDispatch<UserConList>::call((int)cd.lo_mask, [&](auto tag) {  // (lo_mask of a USER constraint: index of its type)
    using F = typename decltype(tag)::type;
    constexpr int P = F::p, NP = F::nparams;
    T par[NP > 0 ? NP : 1], c[P];
    load_user_params<T, NP>(C, cd.per_instance, cd.param_off, par);
    F::eval(x, u, par, c);
#pragma unroll
    for (int i = 0; i < P; ++i) {
      T lam = C.lam(r0 + i);
      T lp = dual_proj(cd.type, lam - rho * c[i]);
      a += lp * lp;
      bsum += lam * lam;
      if (STORE) {
        C.store_c(r0 + i, c[i]);
        vmax = max_(vmax, violation(cd.type, c[i]));
      }
    }
  });
}

// ALCost::Evaluate (al_cost.hpp:264-274) = quadratic cost + sum of ConstraintValues::AugLag
// (constraint_values.hpp:111-119, quirk Q2: scalar rho = penalty_(0)).
// STORE: also store c_ (the side effect every Evaluate has in the reference, quirk Q6) and return
// the knot's max violation through *viol.
template <class T, int n, int m, bool STORE, class Ctx>
ALTRO_DEV T knot_cost(const Ctx& C, const ProblemDesc* pd, const KnotClass& kc, int rb, const T* x, const T* u,
                      T* viol) {
  T J = quad_cost<T, n, m>(C, pd->grp[kc.cost_group], x, u);
  T vmax = T(0);
  for (int ci = 0; ci < kc.ncon; ++ci) {
    const ConDesc& cd = kc.con[ci];
    const int r0 = rb + cd.row_off;
    const T rho = C.pen(r0);
    T a = T(0), bsum = T(0);
    if (cd.kind == ALTRO_CON_GOAL) {
#pragma unroll
      for (int i = 0; i < n; ++i) {
        T c = x[i] - C.par(cd.per_instance, cd.param_off, i);
        T lam = C.lam(r0 + i);
        T lp = dual_proj(cd.type, lam - rho * c);
        a += lp * lp;
        bsum += lam * lam;
        if (STORE) {
          C.store_c(r0 + i, c);
          vmax = max_(vmax, violation(cd.type, c));
        }
      }
    } else if (cd.kind == ALTRO_CON_CONTROL_BOUND) {
      int r = r0, pi = cd.param_off;
#pragma unroll
      for (int j = 0; j < m; ++j)
        if ((cd.lo_mask >> j) & 1u) {
          T c = C.shared(pi) - u[j];
          T lam = C.lam(r);
          T lp = dual_proj(1, lam - rho * c);
          a += lp * lp;
          bsum += lam * lam;
          if (STORE) {
            C.store_c(r, c);
            vmax = max_(vmax, violation(1, c));
          }
          ++r;
          ++pi;
        }
#pragma unroll
      for (int j = 0; j < m; ++j)
        if ((cd.hi_mask >> j) & 1u) {
          T c = u[j] - C.shared(pi);
          T lam = C.lam(r);
          T lp = dual_proj(1, lam - rho * c);
          a += lp * lp;
          bsum += lam * lam;
          if (STORE) {
            C.store_c(r, c);
            vmax = max_(vmax, violation(1, c));
          }
          ++r;
          ++pi;
        }
    } else if (!kHasUserCon || cd.kind == ALTRO_CON_CIRCLE) {  // examples/obstacle_constraints.hpp:99-107
      for (int i = 0; i < cd.p; ++i) {
        T dx = x[0] - C.par(cd.per_instance, cd.param_off, 3 * i);
        T dy = x[1] - C.par(cd.per_instance, cd.param_off, 3 * i + 1);
        T rr = C.par(cd.per_instance, cd.param_off, 3 * i + 2);
        T c = circle_value(dx, dy, rr);
        T lam = C.lam(r0 + i);
        T lp = dual_proj(1, lam - rho * c);
        a += lp * lp;
        bsum += lam * lam;
        if (STORE) {
          C.store_c(r0 + i, c);
          vmax = max_(vmax, violation(1, c));
        }
      }
    } else {
      user_con_auglag<T, STORE>(C, cd, r0, rho, x, u, a, bsum, vmax);
    }
    T Jc = a - bsum;
    J += Jc / (2 * rho);
  }
  if (STORE && viol) *viol = vmax;
  return J;
}

// Constants of one run of knots, hoisted into VGPRs before the serial rollout loop: diagonal cost
// weights, linear terms, and the finite bounds of the run's first CONTROL_BOUND constraint.
template <class T, int n, int m>
struct RunConsts {
  bool diag;
  T Qd[n], Rd[m], q[n], r[m], c;
  int bnd_ci;  // index of the hoisted bound constraint, -1 if none
  T bnd[2 * m];
};
template <class T, int n, int m, class Ctx>
ALTRO_DEV void load_run_consts(const Ctx& C, const ProblemDesc* pd, const KnotClass& kc, RunConsts<T, n, m>& R) {
  const CostGroupDesc& g = pd->grp[kc.cost_group];
  R.diag = g.q_diag && g.r_diag && !(kHasUserCost && g.user);  // a user cost goes through quad_cost()
#pragma unroll
  for (int i = 0; i < n; ++i) {
    R.Qd[i] = C.shared(g.Q_off + i + i * n);
    R.q[i] = C.par(g.q_pi, g.q_off, i);
  }
#pragma unroll
  for (int i = 0; i < m; ++i) {
    R.Rd[i] = C.shared(g.R_off + i + i * m);
    R.r[i] = C.par(g.r_pi, g.r_off, i);
  }
  R.c = C.par(g.c_pi, g.c_off, 0);
  R.bnd_ci = -1;
#pragma unroll
  for (int j = 0; j < 2 * m; ++j) R.bnd[j] = T(0);
#pragma unroll
  for (int ci = 0; ci < kMaxConPerKnot; ++ci) {
    if (ci < kc.ncon && R.bnd_ci < 0 && kc.con[ci].kind == ALTRO_CON_CONTROL_BOUND) {
      R.bnd_ci = ci;
      const ConDesc& cd = kc.con[ci];
      int pi = cd.param_off;
#pragma unroll
      for (int j = 0; j < m; ++j)
        if ((cd.lo_mask >> j) & 1u) R.bnd[j] = C.shared(pi++);
#pragma unroll
      for (int j = 0; j < m; ++j)
        if ((cd.hi_mask >> j) & 1u) R.bnd[m + j] = C.shared(pi++);
    }
  }
}
// Same value as knot_cost<.., STORE=false> (identical operation order), with the run constants in
// registers and the constraint list unrolled so that class metadata stays in scalar registers.
template <class T, int n, int m, class Ctx>
ALTRO_DEV T knot_cost_fast(const Ctx& C, const ProblemDesc* pd, const KnotClass& kc, const RunConsts<T, n, m>& R,
                           int rb, const T* x, const T* u) {
  T J;
  if (R.diag) {
    T xQx = T(0), uRu = T(0), qx = T(0), ru = T(0);
#pragma unroll
    for (int i = 0; i < n; ++i) {
      xQx += x[i] * (R.Qd[i] * x[i]);
      qx += R.q[i] * x[i];
    }
#pragma unroll
    for (int i = 0; i < m; ++i) {
      uRu += u[i] * (R.Rd[i] * u[i]);
      ru += R.r[i] * u[i];
    }
    J = T(0.5) * xQx + T(0.5) * uRu + qx + ru + R.c;
  } else {
    J = quad_cost<T, n, m>(C, pd->grp[kc.cost_group], x, u);
  }
#pragma unroll
  for (int ci = 0; ci < kMaxConPerKnot; ++ci) {
    if (ci < kc.ncon) {
      const ConDesc& cd = kc.con[ci];
      const int r0 = rb + cd.row_off;
      const T rho = C.pen(r0);
      T a = T(0), bsum = T(0);
      if (cd.kind == ALTRO_CON_GOAL) {
#pragma unroll
        for (int i = 0; i < n; ++i) {
          T c = x[i] - C.par(cd.per_instance, cd.param_off, i);
          T lam = C.lam(r0 + i);
          T lp = dual_proj(cd.type, lam - rho * c);
          a += lp * lp;
          bsum += lam * lam;
        }
      } else if (cd.kind == ALTRO_CON_CONTROL_BOUND) {
        const bool hoisted = (ci == R.bnd_ci);
        int r = r0, pi = cd.param_off;
#pragma unroll
        for (int j = 0; j < m; ++j)
          if ((cd.lo_mask >> j) & 1u) {
            T c = (hoisted ? R.bnd[j] : C.shared(pi)) - u[j];
            T lam = C.lam(r);
            T lp = dual_proj(1, lam - rho * c);
            a += lp * lp;
            bsum += lam * lam;
            ++r;
            ++pi;
          }
#pragma unroll
        for (int j = 0; j < m; ++j)
          if ((cd.hi_mask >> j) & 1u) {
            T c = u[j] - (hoisted ? R.bnd[m + j] : C.shared(pi));
            T lam = C.lam(r);
            T lp = dual_proj(1, lam - rho * c);
            a += lp * lp;
            bsum += lam * lam;
            ++r;
            ++pi;
          }
      } else if (!kHasUserCon || cd.kind == ALTRO_CON_CIRCLE) {
        for (int i = 0; i < cd.p; ++i) {
          T dx = x[0] - C.par(cd.per_instance, cd.param_off, 3 * i);
          T dy = x[1] - C.par(cd.per_instance, cd.param_off, 3 * i + 1);
          T rr = C.par(cd.per_instance, cd.param_off, 3 * i + 2);
          T c = circle_value(dx, dy, rr);
          T lam = C.lam(r0 + i);
          T lp = dual_proj(1, lam - rho * c);
          a += lp * lp;
          bsum += lam * lam;
        }
      } else {
        T vmax_unused = T(0);
        user_con_auglag<T, false>(C, cd, r0, rho, x, u, a, bsum, vmax_unused);
      }
      T Jc = a - bsum;
      J += Jc / (2 * rho);
    }
  }
  return J;
}

// Cost expansion of one knot: QuadraticCost::Gradient/Hessian + ConstraintValues::AugLagGradient /
// AugLagHessian for every constraint (al_cost.hpp:276-308, quadratic_cost.cpp:13-28,
// constraint_values.hpp:131-177).  Also returns the AL cost (ilqr.hpp:675) and stores c_.
template <class T, int n, int m, class Ctx>
ALTRO_DEV T knot_cost_expansion(const Ctx& C, const ProblemDesc* pd, const KnotClass& kc, int rb, const T* x,
                                const T* u, T* gx, T* gu, T* hxx, T* hxu, T* huu) {
  const CostGroupDesc& g = pd->grp[kc.cost_group];
  T J;
  bool user_cost = false;
  if constexpr (kHasUserCost) user_cost = g.user != 0;
  if (user_cost) {
    // CostFunction::Evaluate / Gradient / Hessian of the user's cost (costfunction.hpp:52-73)
    J = T(0);
    UserDispatch<UserCostList>::call(g.user - 1, [&](auto tag) {
      using F = typename decltype(tag)::type;
      constexpr int NP = F::nparams;
      T par[NP > 0 ? NP : 1];
      load_user_params<T, NP>(C, g.u_pi, g.u_off, par);
      J = F::eval(x, u, par);
      F::gradient(x, u, par, gx, gu);
      F::hessian(x, u, par, hxx, hxu, huu);
    });
  } else {
    T xQx = T(0), uRu = T(0), qx = T(0), ru = T(0);
#pragma unroll
    for (int i = 0; i < n; ++i) {
      T s = T(0);
#pragma unroll
      for (int j = 0; j < n; ++j) {
        T q = C.shared(g.Q_off + i + j * n);
        hxx[i + j * n] = q;
        s += q * x[j];
      }
      T qi = C.par(g.q_pi, g.q_off, i);
      gx[i] = s + qi;  // (Qx + q) + Hu with H == 0
      xQx += x[i] * s;
      qx += qi * x[i];
    }
#pragma unroll
    for (int i = 0; i < m; ++i) {
      T s = T(0);
#pragma unroll
      for (int j = 0; j < m; ++j) {
        T r = C.shared(g.R_off + i + j * m);
        huu[i + j * m] = r;
        s += r * u[j];
      }
      T ri = C.par(g.r_pi, g.r_off, i);
      gu[i] = s + ri;
      uRu += u[i] * s;
      ru += ri * u[i];
    }
#pragma unroll
    for (int e = 0; e < n * m; ++e) hxu[e] = T(0);
    J = T(0.5) * xQx + T(0.5) * uRu + qx + ru + C.par(g.c_pi, g.c_off, 0);
  }

  for (int ci = 0; ci < kc.ncon; ++ci) {
    const ConDesc& cd = kc.con[ci];
    const int r0 = rb + cd.row_off;
    const T rho = C.pen(r0);
    T a = T(0), bsum = T(0);
    if (cd.kind == ALTRO_CON_GOAL) {
      // C_x = I, C_u = 0: gradient -(P C)^T lambda_bar, Gauss-Newton Hessian rho (PC)^T(PC)
#pragma unroll
      for (int i = 0; i < n; ++i) {
        T c = x[i] - C.par(cd.per_instance, cd.param_off, i);
        T lam = C.lam(r0 + i);
        T v = lam - rho * c;
        T lp = dual_proj(cd.type, v);
        T pj = dual_proj_jac(cd.type, v);
        a += lp * lp;
        bsum += lam * lam;
        C.store_c(r0 + i, c);
        gx[i] += -(pj * lp);
        hxx[i + i * n] += (rho * pj) * pj;
      }
    } else if (cd.kind == ALTRO_CON_CONTROL_BOUND) {
      // rows: finite lower bounds (c = lb - u_j, dc/du_j = -1), then finite upper bounds (+1)
      T sg[m], sh[m];
#pragma unroll
      for (int j = 0; j < m; ++j) sg[j] = sh[j] = T(0);
      int r = r0, pi = cd.param_off;
#pragma unroll
      for (int j = 0; j < m; ++j)
        if ((cd.lo_mask >> j) & 1u) {
          T c = C.shared(pi) - u[j];
          T lam = C.lam(r);
          T v = lam - rho * c;
          T lp = dual_proj(1, v);
          T jp = dual_proj_jac(1, v) * T(-1);
          a += lp * lp;
          bsum += lam * lam;
          C.store_c(r, c);
          sg[j] += jp * lp;
          sh[j] += (rho * jp) * jp;
          ++r;
          ++pi;
        }
#pragma unroll
      for (int j = 0; j < m; ++j)
        if ((cd.hi_mask >> j) & 1u) {
          T c = u[j] - C.shared(pi);
          T lam = C.lam(r);
          T v = lam - rho * c;
          T lp = dual_proj(1, v);
          T jp = dual_proj_jac(1, v);
          a += lp * lp;
          bsum += lam * lam;
          C.store_c(r, c);
          sg[j] += jp * lp;
          sh[j] += (rho * jp) * jp;
          ++r;
          ++pi;
        }
#pragma unroll
      for (int j = 0; j < m; ++j) {
        gu[j] += -sg[j];
        huu[j + j * m] += sh[j];
      }
    } else if (kHasUserCon && cd.kind == ALTRO_CON_USER) {
      // the user's constraint: con_->Evaluate / Jacobian, then ConstraintValues::AugLagGradient / AugLagHessian
      // (constraint_values.hpp:131-177) with the full p x (n+m) Jacobian
      UserDispatch<UserConList>::call((int)cd.lo_mask, [&](auto tag) {
      using F = typename decltype(tag)::type;
      constexpr int P = F::p, NP = F::nparams, nm = n + m;
      T par[NP > 0 ? NP : 1], c[P], jac[P * nm], lp[P];
      load_user_params<T, NP>(C, cd.per_instance, cd.param_off, par);
      F::eval(x, u, par, c);
      F::jacobian(x, u, par, jac);
#pragma unroll
      for (int r = 0; r < P; ++r) {
        T lam = C.lam(r0 + r);
        T v = lam - rho * c[r];
        lp[r] = dual_proj(cd.type, v);
        T pj = dual_proj_jac(cd.type, v);
        a += lp[r] * lp[r];
        bsum += lam * lam;
        C.store_c(r0 + r, c[r]);
#pragma unroll
        for (int j = 0; j < nm; ++j) jac[r + j * P] = pj * jac[r + j * P];  // jac_proj = proj_jac * jac
      }
#pragma unroll
      for (int i = 0; i < n; ++i) {
        T sg = T(0);
#pragma unroll
        for (int r = 0; r < P; ++r) sg += jac[r + i * P] * lp[r];
        gx[i] += -sg;
      }
#pragma unroll
      for (int i = 0; i < m; ++i) {
        T sg = T(0);
#pragma unroll
        for (int r = 0; r < P; ++r) sg += jac[r + (n + i) * P] * lp[r];
        gu[i] += -sg;
      }
#pragma unroll
      for (int j = 0; j < n; ++j)
#pragma unroll
        for (int i = 0; i < n; ++i) {
          T sh = T(0);
#pragma unroll
          for (int r = 0; r < P; ++r) sh += (rho * jac[r + i * P]) * jac[r + j * P];
          hxx[i + j * n] += sh;
        }
#pragma unroll
      for (int j = 0; j < m; ++j)
#pragma unroll
        for (int i = 0; i < n; ++i) {
          T sh = T(0);
#pragma unroll
          for (int r = 0; r < P; ++r) sh += (rho * jac[r + i * P]) * jac[r + (n + j) * P];
          hxu[i + j * n] += sh;
        }
#pragma unroll
      for (int j = 0; j < m; ++j)
#pragma unroll
        for (int i = 0; i < m; ++i) {
          T sh = T(0);
#pragma unroll
          for (int r = 0; r < P; ++r) sh += (rho * jac[r + (n + i) * P]) * jac[r + (n + j) * P];
          huu[i + j * m] += sh;
        }
      });
    } else {  // CIRCLE: dc_i/d(px,py) = (2(cx-px), 2(cy-py)) (obstacle_constraints.hpp:109-121)
      T g0 = T(0), g1 = T(0), h00 = T(0), h10 = T(0), h01 = T(0), h11 = T(0);
      for (int i = 0; i < cd.p; ++i) {
        T cx = C.par(cd.per_instance, cd.param_off, 3 * i);
        T cy = C.par(cd.per_instance, cd.param_off, 3 * i + 1);
        T rr = C.par(cd.per_instance, cd.param_off, 3 * i + 2);
        T dx = x[0] - cx, dy = x[1] - cy;
        T c = circle_value(dx, dy, rr);
        T lam = C.lam(r0 + i);
        T v = lam - rho * c;
        T lp = dual_proj(1, v);
        T pj = dual_proj_jac(1, v);
        a += lp * lp;
        bsum += lam * lam;
        C.store_c(r0 + i, c);
        T j0 = pj * (2 * (cx - x[0]));
        T j1 = pj * (2 * (cy - x[1]));
        g0 += j0 * lp;
        g1 += j1 * lp;
        h00 += (rho * j0) * j0;
        h10 += (rho * j1) * j0;
        h01 += (rho * j0) * j1;
        h11 += (rho * j1) * j1;
      }
      gx[0] += -g0;
      gx[1] += -g1;
      hxx[0 + 0 * n] += h00;
      hxx[1 + 0 * n] += h10;
      hxx[0 + 1 * n] += h01;
      hxx[1 + 1 * n] += h11;
    }
    T Jc = a - bsum;
    J += Jc / (2 * rho);
  }
  return J;
}

// -------------------------------------------------------------------------------------------------
// One knot of the backward Riccati recursion: CalcActionValueExpansion, RegularizeActionValue,
// CalcGains, CalcCostToGo, AddCostToGo (altro/ilqr/knot_point_function_type.hpp:149-235).
// In: [A|B], cost expansion, P/p of knot k+1, regularisation rho.  Out (only on success): K, d,
// P/p of knot k (overwritten in place), dV += (d^T Qu, 0.5 d^T Quu d).  Returns false when the
// Cholesky factorisation of Quu + rho I hits a non-positive pivot (Eigen::NumericalIssue).
// Gains come from the REGULARISED Q, cost-to-go from the UN-regularised Q (quirk Q3).
// -------------------------------------------------------------------------------------------------
// Part 1: action-value expansion (knot_point_function_type.hpp:149-164).  Consumes [A|B] and the
// cost expansion; after it returns the caller may overwrite those registers with the next knot's.
template <class T, int n, int m>
struct QExp {
  T Qxx[n * n], Qxu[n * m], Quu[m * m], Qx[n], Qu[m];
};
template <class T, int n, int m>
ALTRO_DEV void riccati_q(const T* AB, const T* lxx, const T* lxu, const T* luu, const T* lx, const T* lu,
                         const T* P, const T* p, QExp<T, n, m>& Q) {
  const T* A = AB;
  const T* Bm = AB + n * n;
  T AtP[n * n], BtP[m * n];
#pragma unroll
  for (int j = 0; j < n; ++j) {
#pragma unroll
    for (int i = 0; i < n; ++i) {
      T s = T(0);
#pragma unroll
      for (int l = 0; l < n; ++l) s += A[l + i * n] * P[l + j * n];
      AtP[i + j * n] = s;
    }
#pragma unroll
    for (int i = 0; i < m; ++i) {
      T s = T(0);
#pragma unroll
      for (int l = 0; l < n; ++l) s += Bm[l + i * n] * P[l + j * n];
      BtP[i + j * m] = s;
    }
  }
#pragma unroll
  for (int j = 0; j < n; ++j)
#pragma unroll
    for (int i = 0; i < n; ++i) {
      T s = T(0);
#pragma unroll
      for (int l = 0; l < n; ++l) s += AtP[i + l * n] * A[l + j * n];
      Q.Qxx[i + j * n] = lxx[i + j * n] + s;
    }
#pragma unroll
  for (int j = 0; j < m; ++j)
#pragma unroll
    for (int i = 0; i < n; ++i) {
      T s = T(0);
#pragma unroll
      for (int l = 0; l < n; ++l) s += AtP[i + l * n] * Bm[l + j * n];
      Q.Qxu[i + j * n] = lxu[i + j * n] + s;
    }
#pragma unroll
  for (int j = 0; j < m; ++j)
#pragma unroll
    for (int i = 0; i < m; ++i) {
      T s = T(0);
#pragma unroll
      for (int l = 0; l < n; ++l) s += BtP[i + l * m] * Bm[l + j * n];
      Q.Quu[i + j * m] = luu[i + j * m] + s;
    }
#pragma unroll
  for (int i = 0; i < n; ++i) {
    T s = T(0);
#pragma unroll
    for (int l = 0; l < n; ++l) s += A[l + i * n] * p[l];
    Q.Qx[i] = lx[i] + s;
  }
#pragma unroll
  for (int i = 0; i < m; ++i) {
    T s = T(0);
#pragma unroll
    for (int l = 0; l < n; ++l) s += Bm[l + i * n] * p[l];
    Q.Qu[i] = lu[i] + s;
  }
}

// Part 2: RegularizeActionValue, CalcGains, CalcCostToGo, AddCostToGo
// (knot_point_function_type.hpp:175-235).  Out (only on success): K, d, P/p of this knot
// (overwriting the next knot's), dV += (d^T Qu, 0.5 d^T Quu d).  Returns false when the Cholesky
// factorisation of Quu + rho I hits a non-positive pivot (Eigen::NumericalIssue).
// Gains come from the REGULARISED Q, cost-to-go from the UN-regularised Q (quirk Q3).
template <class T, int n, int m>
ALTRO_DEV bool riccati_gains(const QExp<T, n, m>& Q, T rho, T* P, T* p, T* K, T* d, double* dV0, double* dV1) {
  const T* Qxx = Q.Qxx;
  const T* Qxu = Q.Qxu;
  const T* Quu = Q.Quu;
  const T* Qx = Q.Qx;
  const T* Qu = Q.Qu;
  // Eigen::LLT of Quu + rho I (lower); a pivot <= 0 is a failure
  T L[m * m], Linv[m];
#pragma unroll
  for (int e = 0; e < m * m; ++e) L[e] = Quu[e];
#pragma unroll
  for (int i = 0; i < m; ++i) L[i + i * m] += rho;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < m; ++j) {
    T xjj = L[j + j * m];
#pragma unroll
    for (int l = 0; l < j; ++l) xjj -= L[j + l * m] * L[j + l * m];
    if (xjj <= T(0)) ok = false;
    T ljj = sqrt_(xjj);
    L[j + j * m] = ljj;
    // one reciprocal per pivot; the triangular solves below multiply by it instead of dividing
    // (an fp64 division costs ~35 instructions on gfx950 and there would be (n+1)*2m of them)
    Linv[j] = T(1) / ljj;
#pragma unroll
    for (int i = j + 1; i < m; ++i) {
      T s = L[i + j * m];
#pragma unroll
      for (int l = 0; l < j; ++l) s -= L[i + l * m] * L[j + l * m];
      L[i + j * m] = s * Linv[j];
    }
  }
  if (!ok) return false;
  // K = -(L L^T)^-1 Qxu^T (m x n), d = -(L L^T)^-1 Qu
#pragma unroll
  for (int j = 0; j <= n; ++j) {
    T col[m];
#pragma unroll
    for (int i = 0; i < m; ++i) col[i] = (j < n) ? Qxu[(j < n ? j : 0) + i * n] : Qu[i];
#pragma unroll
    for (int i = 0; i < m; ++i) {
      T s = col[i];
#pragma unroll
      for (int l = 0; l < i; ++l) s -= L[i + l * m] * col[l];
      col[i] = s * Linv[i];
    }
#pragma unroll
    for (int i = m - 1; i >= 0; --i) {
      T s = col[i];
#pragma unroll
      for (int l = i + 1; l < m; ++l) s -= L[l + i * m] * col[l];
      col[i] = s * Linv[i];
    }
#pragma unroll
    for (int i = 0; i < m; ++i) {
      if (j < n)
        K[i + (j < n ? j : 0) * m] = -col[i];
      else
        d[i] = -col[i];
    }
  }
  // cost-to-go with the un-regularised Q (knot_point_function_type.hpp:220-230)
  T KtQuu[n * m];
#pragma unroll
  for (int j = 0; j < m; ++j)
#pragma unroll
    for (int i = 0; i < n; ++i) {
      T s = T(0);
#pragma unroll
      for (int l = 0; l < m; ++l) s += K[l + i * m] * Quu[l + j * m];
      KtQuu[i + j * n] = s;
    }
#pragma unroll
  for (int i = 0; i < n; ++i) {
    T a = T(0), b = T(0), c = T(0);
#pragma unroll
    for (int l = 0; l < m; ++l) {
      a += KtQuu[i + l * n] * d[l];
      b += K[l + i * m] * Qu[l];
      c += Qxu[i + l * n] * d[l];
    }
    p[i] = Qx[i] + a + b + c;
  }
#pragma unroll
  for (int j = 0; j < n; ++j)
#pragma unroll
    for (int i = 0; i < n; ++i) {
      T a = T(0), b = T(0), c = T(0);
#pragma unroll
      for (int l = 0; l < m; ++l) {
        a += KtQuu[i + l * n] * K[l + j * m];
        b += K[l + i * m] * Qxu[j + l * n];
        c += Qxu[i + l * n] * K[l + j * m];
      }
      P[i + j * n] = Qxx[i + j * n] + a + b + c;
    }
  T v0 = T(0), v1 = T(0);
#pragma unroll
  for (int i = 0; i < m; ++i) {
    v0 += d[i] * Qu[i];
    T s = T(0);
#pragma unroll
    for (int l = 0; l < m; ++l) s += Quu[i + l * m] * d[l];
    v1 += d[i] * s;
  }
  *dV0 += (double)v0;
  *dV1 += (double)(T(0.5) * v1);
  return true;
}


// iLQR::IncreaseRegularization / DecreaseRegularization (altro/ilqr/ilqr.hpp:770-786)
ALTRO_DEV void increase_reg(const DevOpts& o, double* rho, double* drho) {
  *drho = max_(*drho * o.bp_reg_increase_factor, o.bp_reg_increase_factor);
  *rho = max_(*rho * *drho, o.bp_reg_min);
  *rho = min_(*rho, o.bp_reg_max);
}
ALTRO_DEV void decrease_reg(const DevOpts& o, double* rho, double* drho) {
  *drho = min_(*drho / o.bp_reg_increase_factor, 1.0 / o.bp_reg_increase_factor);
  *rho = max_(*rho * *drho, o.bp_reg_min);
  *rho = min_(*rho, o.bp_reg_max);
}

}  // namespace altro_hip
