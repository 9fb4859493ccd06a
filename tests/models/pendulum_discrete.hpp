// A user model that is ONLY a problem::DiscreteDynamics (altro/problem/dynamics.hpp:148-187: Evaluate(x, u, t, h, xnext),
// Jacobian(x, u, t, h, jac)) -- no continuous f / jac at all: a damped pendulum advanced by the implicit-midpoint-flavoured
// map the caller derived by hand, x = (theta, omega), u = torque.  `struct UserModel` with `discrete = true`, no list.
struct UserModel {
  static constexpr int n = 2, m = 1;
  static constexpr bool discrete = true;
  template <class T>
  ALTRO_MODEL_FN static void step(const T* x, const T* u, float, float h, T* xn) {
    const T hh = T(h), g = T(9.81), l = T(1.0), b = T(0.1);
    // half step in the angle, full step in the rate from the mid-point torque balance, second half step in the angle
    const T thm = x[0] + T(0.5) * hh * x[1];
    const T om = (x[1] + hh * (u[0] - g / l * sin(thm))) / (T(1) + b * hh);
    xn[0] = thm + T(0.5) * hh * om;
    xn[1] = om;
  }
  template <class T>
  ALTRO_MODEL_FN static void step_jac(const T* x, const T*, float, float h, T* J) {  // 2 x 3, column-major
    const T hh = T(h), g = T(9.81), l = T(1.0), b = T(0.1);
    const T thm = x[0] + T(0.5) * hh * x[1];
    const T den = T(1) + b * hh, gc = g / l * cos(thm);
    // d om / d (theta, omega, u)
    const T o0 = -hh * gc / den, o1 = (T(1) - hh * gc * T(0.5) * hh) / den, o2 = hh / den;
    J[0 + 0 * 2] = T(1) + T(0.5) * hh * o0;
    J[1 + 0 * 2] = o0;
    J[0 + 1 * 2] = T(0.5) * hh + T(0.5) * hh * o1;
    J[1 + 1 * 2] = o1;
    J[0 + 2 * 2] = T(0.5) * hh * o2;
    J[1 + 2 * 2] = o2;
  }
};
