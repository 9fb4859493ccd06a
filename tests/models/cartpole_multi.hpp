// Cart-pole (tests/models/cartpole.hpp) with SEVERAL cost and constraint classes in one source, as a reference
// problem may hold any mix of problem::CostFunction / constraints::Constraint subclasses knot by knot
// (altro/problem/problem.hpp:66-133): two costs with different parameter counts, three constraints with different
// output dimensions and cones.  The lists at the end give each class its `type` index for
// altro_set_user_cost_type / altro_add_user_constraint_type (include/altro_hip.h).  The same text is compiled for the
// host into the test oracle (oracle/Makefile: liboracle_cartpole_multi.so).
struct UserModel {
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

// Cost type 0 (stage): J = 1/2 wp (p - pg)^2 + wt (1 - cos theta) + 1/2 wv pdot^2 + 1/2 ww thetadot^2 + 1/2 r u^2,
// par = (pg, wp, wt, wv, ww, r).
struct SwingCost {
  static constexpr int nparams = 6;
  template <class T>
  ALTRO_MODEL_FN static T eval(const T* x, const T* u, const T* par) {
    const T dp = x[0] - par[0];
    return T(0.5) * par[1] * dp * dp + par[2] * (T(1) - cos(x[1])) + T(0.5) * par[3] * x[2] * x[2] +
           T(0.5) * par[4] * x[3] * x[3] + T(0.5) * par[5] * u[0] * u[0];
  }
  template <class T>
  ALTRO_MODEL_FN static void gradient(const T* x, const T* u, const T* par, T* dx, T* du) {
    dx[0] = par[1] * (x[0] - par[0]);
    dx[1] = par[2] * sin(x[1]);
    dx[2] = par[3] * x[2];
    dx[3] = par[4] * x[3];
    du[0] = par[5] * u[0];
  }
  template <class T>
  ALTRO_MODEL_FN static void hessian(const T* x, const T*, const T* par, T* dxdx, T* dxdu, T* dudu) {
    for (int i = 0; i < 16; ++i) dxdx[i] = T(0);
    for (int i = 0; i < 4; ++i) dxdu[i] = T(0);
    dxdx[0 + 0 * 4] = par[1];
    dxdx[1 + 1 * 4] = par[2] * cos(x[1]);
    dxdx[2 + 2 * 4] = par[3];
    dxdx[3 + 3 * 4] = par[4];
    dudu[0] = par[5];
  }
};

// Cost type 1 (terminal): the pole TIP (p + l sin theta) at the goal and everything at rest,
//   J = 1/2 w (p + l sin theta - pg)^2 + 1/2 wr (theta^2 + pdot^2 + thetadot^2),   par = (pg, w, wr).
struct TipCost {
  static constexpr int nparams = 3;
  template <class T>
  ALTRO_MODEL_FN static T eval(const T* x, const T*, const T* par) {
    const T e = x[0] + T(0.5) * sin(x[1]) - par[0];
    return T(0.5) * par[1] * e * e + T(0.5) * par[2] * (x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
  }
  template <class T>
  ALTRO_MODEL_FN static void gradient(const T* x, const T*, const T* par, T* dx, T* du) {
    const T e = x[0] + T(0.5) * sin(x[1]) - par[0], dt = T(0.5) * cos(x[1]);
    dx[0] = par[1] * e;
    dx[1] = par[1] * e * dt + par[2] * x[1];
    dx[2] = par[2] * x[2];
    dx[3] = par[2] * x[3];
    du[0] = T(0);
  }
  template <class T>
  ALTRO_MODEL_FN static void hessian(const T* x, const T*, const T* par, T* dxdx, T* dxdu, T* dudu) {
    for (int i = 0; i < 16; ++i) dxdx[i] = T(0);
    for (int i = 0; i < 4; ++i) dxdu[i] = T(0);
    const T e = x[0] + T(0.5) * sin(x[1]) - par[0], dt = T(0.5) * cos(x[1]);
    dxdx[0 + 0 * 4] = par[1];
    dxdx[1 + 0 * 4] = dxdx[0 + 1 * 4] = par[1] * dt;
    dxdx[1 + 1 * 4] = par[1] * (dt * dt - e * T(0.5) * sin(x[1])) + par[2];
    dxdx[2 + 2 * 4] = par[2];
    dxdx[3 + 3 * 4] = par[2];
    dudu[0] = T(0);
  }
};

// Constraint type 0, Constraint<NegativeOrthant>, 2 rows: the sway of the pole tip, l sin(theta), inside [lo, hi];
// par = (lo, hi).
struct SwayLimit {
  static constexpr int p = 2, nparams = 2;
  static constexpr bool equality = false;
  template <class T>
  ALTRO_MODEL_FN static void eval(const T* x, const T*, const T* par, T* c) {
    const T sway = T(0.5) * sin(x[1]);
    c[0] = sway - par[1];
    c[1] = par[0] - sway;
  }
  template <class T>
  ALTRO_MODEL_FN static void jacobian(const T* x, const T*, const T*, T* J) {  // p x (n + m), column-major
    for (int i = 0; i < 2 * 5; ++i) J[i] = T(0);
    const T dt = T(0.5) * cos(x[1]);
    J[0 + 1 * 2] = dt;
    J[1 + 1 * 2] = -dt;
  }
};

// Constraint type 1, Constraint<Equality>, 1 row: the pole tip exactly over the goal; par = (pg).
struct TipAtGoal {
  static constexpr int p = 1, nparams = 1;
  static constexpr bool equality = true;
  template <class T>
  ALTRO_MODEL_FN static void eval(const T* x, const T*, const T* par, T* c) {
    c[0] = x[0] + T(0.5) * sin(x[1]) - par[0];
  }
  template <class T>
  ALTRO_MODEL_FN static void jacobian(const T* x, const T*, const T*, T* J) {
    for (int i = 0; i < 5; ++i) J[i] = T(0);
    J[0] = T(1);
    J[1] = T(0.5) * cos(x[1]);
  }
};

// Constraint type 2, Constraint<NegativeOrthant>, 1 row: kinetic limit on the cart, pdot^2 <= vmax^2; par = (vmax).
struct SpeedLimit {
  static constexpr int p = 1, nparams = 1;
  static constexpr bool equality = false;
  template <class T>
  ALTRO_MODEL_FN static void eval(const T* x, const T*, const T* par, T* c) {
    c[0] = x[2] * x[2] - par[0] * par[0];
  }
  template <class T>
  ALTRO_MODEL_FN static void jacobian(const T* x, const T*, const T*, T* J) {
    for (int i = 0; i < 5; ++i) J[i] = T(0);
    J[2] = T(2) * x[2];
  }
};

#define ALTRO_USER_COSTS SwingCost, TipCost
#define ALTRO_USER_CONSTRAINTS SwayLimit, TipAtGoal, SpeedLimit
