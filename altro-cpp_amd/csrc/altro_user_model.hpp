// altro_user_model.hpp — body of a USER-MODEL PLUGIN.
//
// The reference's extension point for dynamics is C++ inheritance: a user subclasses
// problem::ContinuousDynamics (altro/problem/dynamics.hpp:59-95) -- Evaluate(x, u, t, xdot) and Jacobian(x, u, t,
// jac) over Eigen::Ref arguments -- wraps it in DiscretizedModel<Model, RungeKutta4>
// (discretized_model.hpp:24-65) and hands it to Problem::SetDynamics.  A GPU kernel cannot call host virtual
// functions, so here the user hands over the SOURCE of the two functions instead,
//
//     struct UserModel {
//       static constexpr int n = 4, m = 1;                                            // StateDimension / ControlDimension
//       template <class T> ALTRO_MODEL_FN static void f(const T* x, const T* u, T* xdot);    // Evaluate
//       template <class T> ALTRO_MODEL_FN static void jac(const T* x, const T* u, T* J);     // Jacobian: n x (n+m), column-major
//     };
//
// altro_register_model_source() wraps it into a translation unit that includes the engine headers and this file,
// compiles it with hipcc for the device's architecture into a small shared object (cached on disk by content
// hash), and loads it: the plugin carries Engine<double, UserModel> -- every kernel of the solver instantiated for
// the user's dynamics, exactly as the built-in models are -- plus the device-side counterpart of
// FunctionBase::CheckJacobian (altro/common/functionbase.cpp:35-73), which is run once at registration.
// No file of the library is edited to add a model.
//
// The same source may define the reference's other two plug-in classes (include/altro_hip.h): a cost function
// (problem::CostFunction, costfunction.hpp:52-73 -> struct UserCost, #define ALTRO_USER_COST UserCost) and a
// constraint (constraints::Constraint<ConType>, constraint.hpp:173-202 -> struct UserConstraint,
// #define ALTRO_USER_CONSTRAINT UserConstraint) -- or SEVERAL types of each, as a reference problem may hold any
// mix of CostFunction / Constraint subclasses (problem.hpp:66-133): #define ALTRO_USER_COSTS TypeA, TypeB and
// #define ALTRO_USER_CONSTRAINTS TypeC, TypeD; the index in the list is the `type` of altro_set_user_cost_type /
// altro_add_user_constraint_type.  altro_device.hpp picks them up (UserCostList / UserConList); the derivatives of
// every type are checked on the device like the model's (k_check_functors below).
#pragma once

#include "altro_engine.hpp"

namespace altro_hip {

#define ALTRO_USER_PLUGIN_ABI 7  // bump when EngineBase or the entry points below change

// a user's model struct with the flags the engine asks every model for (no hand-fused RK4 / carried trigonometry)
template <class S>
struct UserSub : S {
  static constexpr bool kHasFusedRk4 = false;
  static constexpr bool kHasCarriedTrig = false;
  static constexpr bool kHasFusedJacobian = false;
};
#if defined(ALTRO_USER_MODELS)
// SEVERAL models of equal dimensions, one per knot (Problem::SetDynamics(model, k), problem.hpp:155-166, 187-191): the
// index of a model in the list is what altro_set_knot_models assigns to a knot.  Each may be continuous (RK4 or
// explicit Euler) or discrete, time-varying or not (altro_device.hpp: discrete_step / discrete_jacobian).
namespace user_models_ {
using namespace ::altro_user;
template <class... Ms>
struct Wrap {
  using type = UserTypeList<UserSub<Ms>...>;
};
template <class M0, class... Ms>
struct First {
  using type = M0;
  static constexpr bool same_dims = ((Ms::n == M0::n && Ms::m == M0::m) && ...);
};
using List = Wrap<ALTRO_USER_MODELS>::type;
using Head = First<ALTRO_USER_MODELS>;
}  // namespace user_models_
static_assert(user_models_::Head::same_dims, "the models of ALTRO_USER_MODELS must share n and m");
struct UserM {
  using Models = user_models_::List;
  static constexpr int n = user_models_::Head::type::n, m = user_models_::Head::type::m;
  static constexpr bool kHasFusedRk4 = false;
  static constexpr bool kHasCarriedTrig = false;
  static constexpr bool kHasFusedJacobian = false;
};
#else
struct UserM : UserSub<altro_user::UserModel> {};
#endif
using UserModelTypes = model_list<UserM>::type;

// FunctionBase::CheckJacobian (functionbase.cpp:42-73) for the continuous dynamics, one sample point per thread:
// finite differences with step eps against the user's Jacobian; err[s] = the largest ENTRY-WISE error
// |J_fd - J|_ij / max(1, |J_ij|) (an absolute 1e-4 for entries of order one, as MatrixComparison's tolerance,
// functionbase.cpp:15-30, relative for large ones: a norm-wise relative error would let a wrong O(1) entry through
// once the Jacobian as a whole is large).  The reference's helper differences forward
// (utils::FiniteDiffJacobian, derivative_checker.hpp:10-40) and is an opt-in test utility that returns a bool; here the
// check gates registration, so it must not reject a CORRECT Jacobian: central differences (truncation error of order
// eps^2 times the third derivative instead of eps times the second: a model with large second derivatives passes)
// and a relative tolerance.
// A model with `discrete = true` (DiscreteDynamics) is checked the same way on step() / step_jac() with a step of 0.05.
template <class M>
__global__ void k_check_jacobian(const double* __restrict__ z, double* __restrict__ err, int samples, double eps) {
  constexpr int n = M::n, m = M::m, nm = n + m;
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= samples) return;
  double x[nm], f0[n], f1[n], J[n * nm];
#pragma unroll
  for (int i = 0; i < nm; ++i) x[i] = z[(size_t)s * nm + i];
  const float t = 0.37f + 0.01f * (float)(s & 15);  // (a time-varying model is checked at a few knot times)
  constexpr float hcheck = 0.05f;
  auto eval = [&](double* out) {
    if constexpr (model_discrete<M>::value) M::step(x, x + n, t, hcheck, out);
    else model_f<double, M>(x, x + n, t, out);
  };
  if constexpr (model_discrete<M>::value) M::step_jac(x, x + n, t, hcheck, J);
  else model_jac<double, M>(x, x + n, t, J);
  double emax = 0.0;
#pragma unroll
  for (int j = 0; j < nm; ++j) {
    const double keep = x[j];
    x[j] = keep + eps;
    eval(f1);
    x[j] = keep - eps;
    eval(f0);
    x[j] = keep;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const double d = (f1[i] - f0[i]) / (2.0 * eps) - J[i + j * n];
      emax = fmax(emax, fabs(d) / fmax(1.0, fabs(J[i + j * n])));
    }
  }
  err[s] = emax;
}

// ScalarFunction::CheckGradient, FunctionBase::CheckHessian (functionbase.cpp:75-125) for one of the user's cost types
// and FunctionBase::CheckJacobian for one of the user's constraint types (indices into ALTRO_USER_COSTS /
// ALTRO_USER_CONSTRAINTS, -1: none), one sample (x, u, parameters) per thread, central differences:
// err[3 s + 0] = max_j |fd(eval) - gradient|_j, [3 s + 1] = max_ij |fd(gradient) - hessian|_ij,
// [3 s + 2] = max_rj |fd(eval) - jacobian|_rj, each entry relative to max(1, |the user's entry|).
template <int n, int m>
__global__ void k_check_functors(const double* __restrict__ z, const double* __restrict__ par_cost,
                                 const double* __restrict__ par_con, double* __restrict__ err, int samples, double eps,
                                 int cost_type, int con_type) {
  constexpr int nm = n + m;
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= samples) return;
  double x[nm];
#pragma unroll
  for (int i = 0; i < nm; ++i) x[i] = z[(size_t)s * nm + i];
  double eg = 0.0, eh = 0.0, ej = 0.0;
  UserDispatch<UserCostList>::call(cost_type, [&](auto tag) {
    using F = typename decltype(tag)::type;
    constexpr int NP = F::nparams;
    double par[NP > 0 ? NP : 1];
    for (int i = 0; i < NP; ++i) par[i] = par_cost[(size_t)s * NP + i];
    double g0[nm], g1[nm], H[nm * nm], hxx[n * n], hxu[n * m], huu[m * m];
    F::gradient(x, x + n, par, g0, g0 + n);
    F::hessian(x, x + n, par, hxx, hxu, huu);
    for (int j = 0; j < nm; ++j)
      for (int i = 0; i < nm; ++i)
        H[i + j * nm] = (i < n && j < n) ? hxx[i + j * n] : (i < n) ? hxu[i + (j - n) * n] : (j < n) ? hxu[j + (i - n) * n]
                                                                                                     : huu[(i - n) + (j - n) * m];
    for (int j = 0; j < nm; ++j) {  // central differences, see k_check_jacobian
      const double keep = x[j];
      x[j] = keep + eps;
      const double Jp = F::eval(x, x + n, par);
      F::gradient(x, x + n, par, g1, g1 + n);
      x[j] = keep - eps;
      const double Jm = F::eval(x, x + n, par);
      double gm[nm];
      F::gradient(x, x + n, par, gm, gm + n);
      x[j] = keep;
      const double dg = (Jp - Jm) / (2.0 * eps) - g0[j];
      eg = fmax(eg, fabs(dg) / fmax(1.0, fabs(g0[j])));
      for (int i = 0; i < nm; ++i) {
        const double dh = (g1[i] - gm[i]) / (2.0 * eps) - H[i + j * nm];
        eh = fmax(eh, fabs(dh) / fmax(1.0, fabs(H[i + j * nm])));
      }
    }
  });
  UserDispatch<UserConList>::call(con_type, [&](auto tag) {
    using F = typename decltype(tag)::type;
    constexpr int P = F::p, NP = F::nparams;
    double par[NP > 0 ? NP : 1];
    for (int i = 0; i < NP; ++i) par[i] = par_con[(size_t)s * NP + i];
    double c0[P], c1[P], J[P * nm];
    F::jacobian(x, x + n, par, J);
    for (int j = 0; j < nm; ++j) {
      const double keep = x[j];
      x[j] = keep + eps;
      F::eval(x, x + n, par, c1);
      x[j] = keep - eps;
      F::eval(x, x + n, par, c0);
      x[j] = keep;
      for (int r = 0; r < P; ++r) {
        const double dj = (c1[r] - c0[r]) / (2.0 * eps) - J[r + j * P];
        ej = fmax(ej, fabs(dj) / fmax(1.0, fabs(J[r + j * P])));
      }
    }
  });
  err[3 * s + 0] = eg;
  err[3 * s + 1] = eh;
  err[3 * s + 2] = ej;
}

}  // namespace altro_hip

extern "C" {

int altro_user_abi() { return ALTRO_USER_PLUGIN_ABI; }

// the library's book of engines that own chains of sweeps (altro_engine.hpp: ChainClaim): plugins share it
void altro_user_set_chain_hook(int (*fn)(int, int)) {
  if (fn) altro_hip::ChainClaimHook() = fn;
}

void altro_user_dims(int* n, int* m) {
  *n = altro_hip::UserM::n;
  *m = altro_hip::UserM::m;
}
// models the source defines (ALTRO_USER_MODELS; 1 without a list): the valid indices of altro_set_knot_models
int altro_user_model_count() { return altro_hip::UserModelTypes::size; }

// ALTRO_F64: everything fp64; ALTRO_F32: fp32 expansion / gain records (WithRec32<>), like the built-in models
altro_hip::EngineBase* altro_user_make_engine(const altro_desc* d, std::string* err) {
  using namespace altro_hip;
  if (d->dtype == ALTRO_F64) return MakeEngineImpl<double, UserM>(*d, err);
  return MakeEngineImpl<double, WithRec32<UserM>>(*d, err);
}

// What the source defines besides the model: bit 0 cost types, bit 1 constraint types; how many of each
// (ALTRO_USER_COSTS / ALTRO_USER_CONSTRAINTS lists, or the single ALTRO_USER_COST / ALTRO_USER_CONSTRAINT).
int altro_user_functor_info(int* n_cost_types, int* n_con_types) {
  using namespace altro_hip;
  *n_cost_types = UserCostList::size;
  *n_con_types = UserConList::size;
  return (kHasUserCost ? 1 : 0) | (kHasUserCon ? 2 : 0);
}

// Device-side CheckGradient / CheckHessian / CheckJacobian of EVERY cost and constraint type of the source at
// `samples` points: z = (x, u) from the caller, parameters uniform in [0.5, 1.5] (weights, radii, limits: positive).
// errs[0..2] = the largest gradient, Hessian and constraint-Jacobian error over the types, worst[0..2] = the index
// of the type that has it.  Returns 0 or a HIP error code.
int altro_user_check_functors(int device, const double* z_host, int samples, double eps, double* errs, int* worst) {
  using namespace altro_hip;
  errs[0] = errs[1] = errs[2] = 0.0;
  worst[0] = worst[1] = worst[2] = 0;
  if (!kHasUserCost && !kHasUserCon) return 0;
  constexpr int nm = UserM::n + UserM::m;
  constexpr int kCosts = UserCostList::size, kCons = UserConList::size;
  if (hipSetDevice(device) != hipSuccess) return 1;
  unsigned long long st = 0x9E3779B97F4A7C15ull;
  auto next = [&]() {  // splitmix64
    unsigned long long v = (st += 0x9E3779B97F4A7C15ull);
    v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull;
    v = (v ^ (v >> 27)) * 0x94D049BB133111EBull;
    return 0.5 + (double)((v ^ (v >> 31)) >> 11) * 0x1p-53;
  };
  double *dz = nullptr, *derr = nullptr;
  int rc = 0;
  if (hipMalloc((void**)&dz, (size_t)samples * nm * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&derr, (size_t)samples * 3 * sizeof(double)) != hipSuccess)
    rc = 2;
  if (!rc && hipMemcpy(dz, z_host, (size_t)samples * nm * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = 3;
  std::vector<double> herr((size_t)samples * 3, 0.0);
  for (int t = 0; !rc && t < (kCosts > kCons ? kCosts : kCons); ++t) {  // pass t: cost type t beside constraint type t
    const int ct = t < kCosts ? t : -1, kt = t < kCons ? t : -1;
    int dummy_p = 0, dummy_eq = 0, npk = 0;
    const int npc = ct >= 0 ? UserCostParams(ct) : 0;
    if (kt >= 0) UserConInfo(kt, &npk, &dummy_p, &dummy_eq);
    std::vector<double> pc((size_t)samples * (npc > 0 ? npc : 1)), pk((size_t)samples * (npk > 0 ? npk : 1));
    for (double& v : pc) v = next();
    for (double& v : pk) v = next();
    double *dpc = nullptr, *dpk = nullptr;
    if (hipMalloc((void**)&dpc, pc.size() * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&dpk, pk.size() * sizeof(double)) != hipSuccess)
      rc = 2;
    if (!rc && (hipMemcpy(dpc, pc.data(), pc.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(dpk, pk.data(), pk.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess))
      rc = 3;
    if (!rc) {
      hipLaunchKernelGGL((k_check_functors<UserM::n, UserM::m>), dim3((samples + 63) / 64), dim3(64), 0, nullptr, dz, dpc, dpk,
                         derr, samples, eps, ct, kt);
      if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) rc = 4;
    }
    if (!rc && hipMemcpy(herr.data(), derr, herr.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = 3;
    hipFree(dpc);
    hipFree(dpk);
    if (rc) break;
    for (int s = 0; s < samples; ++s)
      for (int q = 0; q < 3; ++q) {
        const double e = herr[(size_t)3 * s + q];
        if (e > errs[q] || e != e) {
          errs[q] = e;
          worst[q] = t;
        }
      }
  }
  hipFree(dz);
  hipFree(derr);
  return rc;
}

// Device-side CheckJacobian at `samples` points z = (x, u) given by the caller (uniform in [-1, 1], like
// VectorXd::Random).  Returns 0 and the largest error, or a HIP error code.
int altro_user_check_jacobian(int device, const double* z_host, int samples, double eps, double* max_err) {
  using namespace altro_hip;
  constexpr int nm = UserM::n + UserM::m;
  if (hipSetDevice(device) != hipSuccess) return 1;
  double *dz = nullptr, *derr = nullptr;
  if (hipMalloc((void**)&dz, (size_t)samples * nm * sizeof(double)) != hipSuccess) return 2;
  if (hipMalloc((void**)&derr, (size_t)samples * sizeof(double)) != hipSuccess) {
    hipFree(dz);
    return 2;
  }
  int rc = 0;
  std::vector<double> herr(samples, 0.0);
  if (hipMemcpy(dz, z_host, (size_t)samples * nm * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = 3;
  double mx = 0.0;
  for (int which = 0; !rc && which < UserModelTypes::size; ++which) {  // every model of the source (one without a list)
    UserDispatch<UserModelTypes>::call(which, [&](auto tag) {
      using S = typename decltype(tag)::type;
      hipLaunchKernelGGL((k_check_jacobian<S>), dim3((samples + 63) / 64), dim3(64), 0, nullptr, dz, derr, samples, eps);
    });
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) rc = 4;
    if (!rc && hipMemcpy(herr.data(), derr, (size_t)samples * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = 3;
    for (double e : herr) mx = (e > mx || e != e) ? e : mx;
  }
  hipFree(dz);
  hipFree(derr);
  *max_err = mx;
  return rc;
}

}  // extern "C"
