// The host-side types of the reference either side of the hot path, as its own tests use them -- no GPU involved:
//   test/problem/problem_test.cpp:24-62   (GetDynamics / GetCostFunction: nullptr or an error while undefined)
//   test/ilqr/ilqr_class_test.cpp:36-70   (base-class pointers out of a problem, dimensions)
//   test/common/trajectory_test.cpp:40-100 (CheckTimeConsistency)
// Same statements against include/ (modulo Eigen and gtest); exit code 0 = all hold.  Runs on CPU (tests/test_facade_compile.py).
#include <cstdio>

#include "altro/common/trajectory.hpp"
#include "altro/problem/discretized_model.hpp"
#include "altro/problem/problem.hpp"
#include "examples/quadratic_cost.hpp"
#include "examples/triple_integrator.hpp"
#include "examples/problems/triple_integrator.hpp"

#define EXPECT(cond)                                                                \
  do {                                                                              \
    if (!(cond)) {                                                                  \
      std::fprintf(stderr, "%s:%d: EXPECT(%s) failed\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                                   \
    }                                                                               \
  } while (0)

int main() {
  using namespace altro;
  int failures = 0;
  const int N = 10;
  {  // problem_test.cpp:24-41 (AddDynamics)
    problem::Problem prob(N);
    problem::DiscretizedModel<examples::TripleIntegrator> model{examples::TripleIntegrator(2)};
    prob.SetDynamics(std::make_shared<problem::DiscretizedModel<examples::TripleIntegrator>>(model), 0);
    EXPECT(prob.GetDynamics(0) != nullptr);
    bool threw = false;
    try {
      prob.GetDynamics(1);
    } catch (const std::runtime_error&) {
      threw = true;  // "Dynamics have not been defined."
    }
    EXPECT(threw);
    for (int k = 1; k < N; ++k) prob.SetDynamics(model, k);
    for (int k = 0; k <= N; ++k) EXPECT(prob.GetDynamics(k) != nullptr);
    EXPECT(prob.GetDynamics(N)->StateDimension() == prob.GetDynamics(N - 1)->StateDimension());  // ilqr_class_test.cpp:69
    EXPECT(prob.GetDynamics(0)->StateDimension() == 6 && prob.GetDynamics(0)->ControlDimension() == 2);
  }
  {  // problem_test.cpp:43-62 (AddCostFunctions)
    problem::Problem prob(N);
    const std::vector<double> Q = problems::Diag(6, 1.0), R = problems::Diag(2, 0.1), xref(6, 0.0), uref(2, 0.0);
    std::shared_ptr<examples::QuadraticCost> costfun = std::make_shared<examples::QuadraticCost>(examples::QuadraticCost::LQRCost(Q, R, xref, uref));
    prob.SetCostFunction(costfun, 5);
    EXPECT(prob.GetCostFunction(5) != nullptr);
    EXPECT(prob.GetCostFunction(0) == nullptr);
    std::vector<std::shared_ptr<examples::QuadraticCost>> costfuns(4, costfun);
    prob.SetCostFunction(costfuns);
    for (int k = 0; k < 4; ++k) EXPECT(prob.GetCostFunction(k) != nullptr);
    EXPECT(prob.GetCostFunction(4) == nullptr);
    std::shared_ptr<problem::CostFunction> base = prob.GetCostFunction(0);  // ilqr_class_test.cpp:40
    EXPECT(std::dynamic_pointer_cast<examples::QuadraticCost>(base) != nullptr);
  }
  {  // the factory's problem is fully defined and hands out its initial state (problem.hpp:242, ilqr.hpp:121)
    problems::TripleIntegratorProblem<2> def;
    problem::Problem prob = def.MakeProblem(true);
    EXPECT(prob.IsFullyDefined());
    EXPECT(prob.GetInitialStatePointer() != nullptr && prob.GetInitialStatePointer()->size() == 6);
    EXPECT(prob.GetDynamics(0)->StateDimension() == 6);
  }
  {  // trajectory_test.cpp:40-52, 84-100 (CheckTimeConsistency)
    Trajectory<3, 2> traj(3, 2, N);
    traj.SetUniformStep(0.1f);
    EXPECT(traj.CheckTimeConsistency());
    traj.SetStep(4, 0.2f);
    EXPECT(!traj.CheckTimeConsistency());
    traj.SetStep(4, 0.1f);
    EXPECT(traj.CheckTimeConsistency());
    traj.SetTime(N, 5.0f);
    EXPECT(!traj.CheckTimeConsistency(1e-6, true));
  }
  std::printf("host_types: %d failures\n", failures);
  return failures == 0 ? 0 : 1;
}
