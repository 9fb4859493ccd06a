// Cart-pole in a gusting cross wind: the cart-pole of cartpole.hpp with a force a sin(w t) on the cart -- a TIME-VARYING
// user model (`time_varying = true`: f and jac take the knot time as a 32-bit float, the counterpart of
// ContinuousDynamics::Evaluate(x, u, t, xdot), altro/problem/dynamics.hpp:59-95).  The same text is compiled for the
// host into the test oracle (oracle/Makefile: liboracle_cartpole_wind.so).
struct UserModel {
  static constexpr int n = 4, m = 1;
  static constexpr bool time_varying = true;
  template <class T>
  ALTRO_MODEL_FN static T gust(float t) {
    return T(0.8) * sin(T(2.5) * T(t));
  }
  template <class T>
  ALTRO_MODEL_FN static void f(const T* x, const T* u, float t, T* xd) {
    const T mc = T(1.0), mp = T(0.2), l = T(0.5), g = T(9.81);
    const T s = sin(x[1]), c = cos(x[1]), q = x[3];
    const T D = mc + mp * s * s;
    const T F = u[0] + gust<T>(t);
    xd[0] = x[2];
    xd[1] = q;
    xd[2] = (F + mp * s * (l * q * q + g * c)) / D;
    xd[3] = (-F * c - mp * l * q * q * c * s - (mc + mp) * g * s) / (l * D);
  }
  template <class T>
  ALTRO_MODEL_FN static void jac(const T* x, const T* u, float t, T* J) {  // n x (n + m), column-major
    const T mc = T(1.0), mp = T(0.2), l = T(0.5), g = T(9.81);
    const T s = sin(x[1]), c = cos(x[1]), q = x[3];
    const T D = mc + mp * s * s, dD = T(2) * mp * s * c;
    const T F = u[0] + gust<T>(t);
    const T N1 = F + mp * s * (l * q * q + g * c);
    const T dN1 = mp * (c * l * q * q + g * (c * c - s * s));
    const T N2 = -F * c - mp * l * q * q * c * s - (mc + mp) * g * s;
    const T dN2 = F * s - mp * l * q * q * (c * c - s * s) - (mc + mp) * g * c;
    for (int i = 0; i < n * (n + m); ++i) J[i] = T(0);
    J[0 + 2 * n] = T(1);
    J[1 + 3 * n] = T(1);
    J[2 + 1 * n] = (dN1 * D - N1 * dD) / (D * D);
    J[2 + 3 * n] = mp * s * T(2) * l * q / D;
    J[2 + 4 * n] = T(1) / D;
    J[3 + 1 * n] = (dN2 * D - N2 * dD) / (l * D * D);
    J[3 + 3 * n] = -T(2) * mp * q * c * s / D;
    J[3 + 4 * n] = -c / (l * D);
  }
};
