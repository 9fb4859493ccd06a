// A two-state model whose Jacobian holds one LARGE entry (1000) beside entries of order one: the derivative check must
// judge every entry on its own (tests/test_user_model_gpu.py::test_wrong_entry_does_not_hide_behind_a_large_jacobian).
struct UserModel {
  static constexpr int n = 2, m = 1;
  template <class T>
  ALTRO_MODEL_FN static void f(const T* x, const T* u, T* xd) {
    xd[0] = T(1000) * x[1];
    xd[1] = u[0] - x[0];
  }
  template <class T>
  ALTRO_MODEL_FN static void jac(const T*, const T*, T* J) {  // 2 x 3, column-major
    J[0] = T(0);
    J[1] = T(-1);
    J[2] = T(1000);
    J[3] = T(0);
    J[4] = T(0);
    J[5] = T(1);
  }
};
