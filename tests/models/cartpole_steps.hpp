// Three DISCRETE dynamics of one cart-pole (x = (p, theta, pdot, thetadot), u = horizontal force; tests/models/cartpole.hpp)
// in one user source -- what Problem::SetDynamics(model, k) may put on a knot in the reference (problem.hpp:155-166):
//   index 0  CartpoleRk4         DiscretizedModel<Model, RungeKutta4>  (integration.hpp:123-169, the default)
//   index 1  CartpoleEuler       DiscretizedModel<Model, ExplicitEuler> (integration.hpp:87-104): `integrator = 1`
//   index 2  CartpoleSymplectic  the caller's own problem::DiscreteDynamics (dynamics.hpp:148-187): `discrete = true`,
//                                step(x, u, t, h, xnext) = semi-implicit Euler (v+ = v + a h, q+ = q + v+ h) with a gust that
//                                reads the knot time t, and its analytic Jacobian step_jac
// and the list that lets every knot pick one (altro_set_knot_models).  The same text is compiled for the host into the
// test oracle (oracle/Makefile: liboracle_cartpole_steps.so).
struct CartpoleBase {
  static constexpr int n = 4, m = 1;
  template <class T>
  ALTRO_MODEL_FN static void f(const T* x, const T* u, T* xd) {
    const T mc = T(1.0), mp = T(0.2), l = T(0.5), g = T(9.81);
    const T s = sin(x[1]), c = cos(x[1]), q = x[3];
    const T D = mc + mp * s * s;
    xd[0] = x[2];
    xd[1] = q;
    xd[2] = (u[0] + mp * s * (l * q * q + g * c)) / D;
    xd[3] = (-u[0] * c - mp * l * q * q * c * s - (mc + mp) * g * s) / (l * D);
  }
  template <class T>
  ALTRO_MODEL_FN static void jac(const T* x, const T* u, T* J) {  // n x (n + m), column-major
    const T mc = T(1.0), mp = T(0.2), l = T(0.5), g = T(9.81);
    const T s = sin(x[1]), c = cos(x[1]), q = x[3];
    const T D = mc + mp * s * s, dD = T(2) * mp * s * c;
    const T N1 = u[0] + mp * s * (l * q * q + g * c);
    const T dN1 = mp * (c * l * q * q + g * (c * c - s * s));
    const T N2 = -u[0] * c - mp * l * q * q * c * s - (mc + mp) * g * s;
    const T dN2 = u[0] * s - mp * l * q * q * (c * c - s * s) - (mc + mp) * g * c;
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
struct CartpoleRk4 : CartpoleBase {};
struct CartpoleEuler : CartpoleBase {
  static constexpr int integrator = 1;  // ExplicitEuler
};
struct CartpoleSymplectic : CartpoleBase {
  static constexpr bool discrete = true;
  // force on the cart: the control plus a gust that depends on the knot time
  template <class T>
  ALTRO_MODEL_FN static T force(const T* u, float t) {
    return u[0] + T(0.3) * sin(T(1.7) * T(t));
  }
  template <class T>
  ALTRO_MODEL_FN static void step(const T* x, const T* u, float t, float h, T* xn) {
    const T hh = T(h), F = force(u, t);
    T xd[4];
    f(x, &F, xd);
    const T v0 = x[2] + xd[2] * hh, v1 = x[3] + xd[3] * hh;
    xn[0] = x[0] + v0 * hh;
    xn[1] = x[1] + v1 * hh;
    xn[2] = v0;
    xn[3] = v1;
  }
  template <class T>
  ALTRO_MODEL_FN static void step_jac(const T* x, const T* u, float t, float h, T* J) {  // n x (n + m), column-major
    const T hh = T(h), F = force(u, t);
    T Jc[4 * 5];
    jac(x, &F, Jc);
    for (int j = 0; j < 5; ++j) {
      const T dv0 = (j == 2 ? T(1) : T(0)) + Jc[2 + j * 4] * hh, dv1 = (j == 3 ? T(1) : T(0)) + Jc[3 + j * 4] * hh;
      J[0 + j * 4] = (j == 0 ? T(1) : T(0)) + dv0 * hh;
      J[1 + j * 4] = (j == 1 ? T(1) : T(0)) + dv1 * hh;
      J[2 + j * 4] = dv0;
      J[3 + j * 4] = dv1;
    }
  }
};
#define ALTRO_USER_MODELS CartpoleRk4, CartpoleEuler, CartpoleSymplectic
