// altro/problems.hpp — C++ problem factories, counterparts of the reference's
// examples/problems/unicycle.{hpp,cpp} and examples/problems/triple_integrator.hpp, on the facade.
#pragma once

#include <cmath>
#include <memory>
#include <random>
#include <vector>

#include "altro.hpp"

namespace altro {
namespace problems {

inline std::vector<double> Diag(int n, double v) {
  std::vector<double> M((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) M[i + (size_t)i * n] = v;
  return M;
}

// examples/problems/unicycle.hpp:22-81, unicycle.cpp:11-89
class UnicycleProblem {
 public:
  static constexpr int NStates = 3;
  static constexpr int NControls = 2;
  enum Scenario { kTurn90, kThreeObstacles };

  int N = 100;
  int batch = 1;
  std::vector<double> xf = {1.5, 1.5, M_PI / 2};  // [3] or [batch][3]
  std::vector<double> x0 = {0, 0, 0};
  std::vector<double> u0 = {0.1, 0.1};
  std::vector<double> circles;  // (cx, cy, r) triples: one instance's, or batch blocks
  double v_bnd = 1.5, w_bnd = 1.5;

  void SetScenario(Scenario s) { scenario_ = s; }
  float GetTimeStep() const { return tf_ / N; }  // float arithmetic, as in the reference (quirk Q1)

  problem::Problem MakeProblem(bool add_constraints = true) {
    problem::Problem prob(N);
    prob.SetBatch(batch);
    std::vector<double> Q, R, Qf, lb, ub;
    float h;
    if (scenario_ == kTurn90) {
      tf_ = 3.0f;
      h = GetTimeStep();
      lb = {-v_bnd, -w_bnd};
      ub = {+v_bnd, +w_bnd};
      Q = Diag(3, 1e-2 * h);
      R = Diag(2, 1e-2 * h);
      Qf = Diag(3, 100.0);
    } else {
      tf_ = 5.0f;
      h = GetTimeStep();
      Q = Diag(3, 1.0 * h);
      R = Diag(2, 0.5 * h);
      Qf = Diag(3, 10.0);
      x0 = {0, 0, 0};
      xf = {3, 3, 0};
      u0 = {0.01, 0.01};
      if (circles.empty()) {
        const double scaling = 3.0;
        for (double c : {0.25, 0.5, 0.75}) {
          circles.push_back(c * scaling);
          circles.push_back(c * scaling);
          circles.push_back(0.425);
        }
      }
      lb = {0, -3};
      ub = {3, +3};
      if (add_constraints) {  // obstacles first: first in the inequality list (unicycle.cpp:55-59)
        examples::CircleConstraint obs;
        obs.SetBatchObstacles(circles, 9);
        for (int k = 1; k < N; ++k) prob.SetConstraint(obs, k);
      }
    }
    const std::vector<double> uref = {0, 0};
    for (int k = 0; k < N; ++k) prob.SetCostFunction(examples::QuadraticCost::LQRCost(Q, R, xf, uref), k);
    prob.SetCostFunction(examples::QuadraticCost::LQRCost(Qf, Diag(2, 0.0), xf, uref, true), N);
    const problem::DiscretizedModel<examples::Unicycle> model{examples::Unicycle()};
    for (int k = 0; k < N; ++k) prob.SetDynamics(model, k);
    if (add_constraints) {
      for (int k = 0; k < N; ++k) prob.SetConstraint(examples::ControlBound(lb, ub), k);
      prob.SetConstraint(examples::GoalConstraint(xf, 3), N);
    }
    prob.SetInitialState(x0);
    return prob;
  }

  std::shared_ptr<Trajectory<3, 2>> InitialTrajectory() const {  // unicycle.hpp:84-92
    auto Z = std::make_shared<Trajectory<3, 2>>(N, batch);
    for (int b = 0; b < batch; ++b)
      for (int k = 0; k < N; ++k) {
        Z->Control(k, b)[0] = u0[0];
        Z->Control(k, b)[1] = u0[1];
      }
    Z->SetUniformStep(GetTimeStep());
    return Z;
  }

  // Seeded synthetic batch of BASELINE config 3 (instance 0 = the reference problem)
  void MakeTurn90Batch(int B, unsigned long long seed = 20260930ULL) {
    batch = B;
    std::mt19937_64 gen(seed);
    std::uniform_real_distribution<double> dxy(-0.5, 0.5), dth(-0.3, 0.3);
    xf.assign((size_t)B * 3, 0.0);
    for (int b = 0; b < B; ++b) {
      xf[3 * b + 0] = 1.5 + (b ? dxy(gen) : 0.0);
      xf[3 * b + 1] = 1.5 + (b ? dxy(gen) : 0.0);
      xf[3 * b + 2] = M_PI / 2 + (b ? dth(gen) : 0.0);
    }
  }

 private:
  Scenario scenario_ = kTurn90;
  float tf_ = 3.0f;
};

// examples/problems/triple_integrator.hpp:22-105 (dof = 2; uref = 0, quirk Q9)
class TripleIntegratorProblem {
 public:
  static constexpr int NStates = 6;
  static constexpr int NControls = 2;
  int N = 10;
  float h = 0.1f;
  std::vector<double> xf = {1, 2, 0, 0, 0, 0};
  std::vector<double> x0 = {-1, -2, 0, 0, 0, 0};
  problem::Problem MakeProblem(bool add_constraints = false) {
    problem::Problem prob(N);
    const std::vector<double> uref = {0, 0};
    for (int k = 0; k < N; ++k)
      prob.SetCostFunction(examples::QuadraticCost::LQRCost(Diag(6, 1.0), Diag(2, 1e-3), xf, uref), k);
    prob.SetCostFunction(examples::QuadraticCost::LQRCost(Diag(6, 1e5), Diag(2, 0.0), xf, uref, true), N);
    const problem::DiscretizedModel<examples::TripleIntegrator> model{examples::TripleIntegrator(2)};
    for (int k = 0; k < N; ++k) prob.SetDynamics(model, k);
    prob.SetInitialState(x0);
    if (add_constraints) {
      for (int k = 0; k < N; ++k) prob.SetConstraint(examples::ControlBound({-100, -200}, {100, 200}), k);
      prob.SetConstraint(examples::GoalConstraint(xf, 6), N);
    }
    return prob;
  }
  std::shared_ptr<Trajectory<6, 2>> InitialTrajectory() const {
    auto Z = std::make_shared<Trajectory<6, 2>>(N, 1);
    Z->SetUniformStep(h);
    return Z;
  }
};

}  // namespace problems
}  // namespace altro
