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
#pragma once

#include "altro_engine.hpp"

namespace altro_hip {

#define ALTRO_USER_PLUGIN_ABI 2  // bump when EngineBase or the entry points below change

struct UserM : altro_user::UserModel {
  static constexpr bool kHasFusedRk4 = false;
  static constexpr bool kHasCarriedTrig = false;
  static constexpr bool kHasFusedJacobian = false;
};

// FunctionBase::CheckJacobian (functionbase.cpp:42-73) for the continuous dynamics, one sample point per thread:
// forward differences with step eps (utils::FiniteDiffJacobian, derivative_checker.hpp:10-40) against the
// user's Jacobian; err[s] = Frobenius norm of the difference (MatrixComparison, functionbase.cpp:15-30).
template <class M>
__global__ void k_check_jacobian(const double* __restrict__ z, double* __restrict__ err, int samples, double eps) {
  constexpr int n = M::n, m = M::m, nm = n + m;
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= samples) return;
  double x[nm], f0[n], f1[n], J[n * nm];
#pragma unroll
  for (int i = 0; i < nm; ++i) x[i] = z[(size_t)s * nm + i];
  M::jac(x, x + n, J);
  M::f(x, x + n, f0);
  double e2 = 0.0;
#pragma unroll
  for (int j = 0; j < nm; ++j) {
    const double keep = x[j];
    x[j] = keep + eps;
    M::f(x, x + n, f1);
    x[j] = keep;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const double d = (f1[i] - f0[i]) / eps - J[i + j * n];
      e2 += d * d;
    }
  }
  err[s] = sqrt(e2);
}

}  // namespace altro_hip

extern "C" {

int altro_user_abi() { return ALTRO_USER_PLUGIN_ABI; }

void altro_user_dims(int* n, int* m) {
  *n = altro_hip::UserM::n;
  *m = altro_hip::UserM::m;
}

// ALTRO_F64: everything fp64; ALTRO_F32: fp32 expansion / gain records (WithRec32<>), like the built-in models
altro_hip::EngineBase* altro_user_make_engine(const altro_desc* d, std::string* err) {
  using namespace altro_hip;
  if (d->dtype == ALTRO_F64) return MakeEngineImpl<double, UserM>(*d, err);
  return MakeEngineImpl<double, WithRec32<UserM>>(*d, err);
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
  if (!rc) {
    hipLaunchKernelGGL((k_check_jacobian<UserM>), dim3((samples + 63) / 64), dim3(64), 0, nullptr, dz, derr, samples, eps);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) rc = 4;
  }
  if (!rc && hipMemcpy(herr.data(), derr, (size_t)samples * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = 3;
  hipFree(dz);
  hipFree(derr);
  double mx = 0.0;
  for (double e : herr) mx = (e > mx || e != e) ? e : mx;
  *max_err = mx;
  return rc;
}

}  // extern "C"
