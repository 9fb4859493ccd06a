// The reference's TripleIntegrator (examples/triple_integrator.cpp:9-33, dof = 2: n = 6, m = 2) written as a USER model,
// once per integrator the reference's tests discretise it with:
//   index 0  TripleIntRk4    DiscretizedModel<TripleIntegrator>                 test/problem/triple_integrator_test.cpp:87-133
//   index 1  TripleIntEuler  DiscretizedModel<TripleIntegrator, ExplicitEuler>  test/problem/triple_integrator_test.cpp:135-156
// (the Euler test's known answer -- A = I + h * shift, B = h on the jerk rows -- pins ExplicitEuler::Jacobian,
// integration.hpp:95-101).  Compiled for the host into oracle/_build/liboracle_tripleint_euler.so as well.
struct TripleIntBase {
  static constexpr int n = 6, m = 2;
  template <class T>
  ALTRO_MODEL_FN static void f(const T* x, const T* u, T* xd) {
    for (int i = 0; i < 2; ++i) {
      xd[i] = x[i + 2];
      xd[i + 2] = x[i + 4];
      xd[i + 4] = u[i];
    }
  }
  template <class T>
  ALTRO_MODEL_FN static void jac(const T*, const T*, T* J) {  // n x (n + m), column-major
    for (int i = 0; i < n * (n + m); ++i) J[i] = T(0);
    for (int i = 0; i < 2; ++i) {
      J[i + (i + 2) * n] = T(1);
      J[(i + 2) + (i + 4) * n] = T(1);
      J[(i + 4) + (n + i) * n] = T(1);
    }
  }
};
struct TripleIntRk4 : TripleIntBase {};
struct TripleIntEuler : TripleIntBase {
  static constexpr int integrator = 1;  // problem::ExplicitEuler
};
#define ALTRO_USER_MODELS TripleIntRk4, TripleIntEuler
