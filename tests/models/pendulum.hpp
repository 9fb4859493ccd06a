// A damped pendulum with a torque input as a CONTINUOUS user model, n = 2, m = 1 (x = (theta, omega)): the smallest
// shape of the 4 x 4 matrix-core backward pass and of the persistent tail kernel (round 4: n <= 3, m <= 2 instead of the
// unicycle's n = 3, m = 2 only).  Compiled for the host into oracle/_build/liboracle_pendulum.so as well.
struct UserModel {
  static constexpr int n = 2, m = 1;
  template <class T>
  ALTRO_MODEL_FN static void f(const T* x, const T* u, T* xd) {
    const T g = T(9.81), l = T(1.0), b = T(0.1);
    xd[0] = x[1];
    xd[1] = u[0] - g / l * sin(x[0]) - b * x[1];
  }
  template <class T>
  ALTRO_MODEL_FN static void jac(const T* x, const T*, T* J) {  // 2 x 3, column-major
    const T g = T(9.81), l = T(1.0), b = T(0.1);
    J[0] = T(0);
    J[1] = -g / l * cos(x[0]);
    J[2] = T(1);
    J[3] = -b;
    J[4] = T(0);
    J[5] = T(1);
  }
};
