// The reference's callers look INTO the AL solver: test/examples/example_unicycle_test.cpp:95-106 (Init(), then the penalty
// of the first inequality of knot 0 is the initial penalty), :62-66 (after Solve(), the duals of the goal constraint) and
// test/augmented_lagrangian/auglag_test.cpp:250-275 (a ConstraintValues pointer taken BEFORE the solve shows the duals after
// UpdateDuals()).  Same statements against include/ (modulo Eigen: vectors are std::vector<double>); exit code 0 = all hold.
#include <cmath>
#include <cstdio>

#include "altro/augmented_lagrangian/al_solver.hpp"
#include "examples/problems/unicycle.hpp"

#define EXPECT(cond)                                                      \
  do {                                                                    \
    if (!(cond)) {                                                        \
      std::fprintf(stderr, "%s:%d: EXPECT(%s) failed\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                         \
    }                                                                     \
  } while (0)

int main() {
  using namespace altro;
  int failures = 0;
  constexpr int NStates = 3, NControls = 2;
  try {
    problems::UnicycleProblem def;
    def.SetScenario(problems::UnicycleProblem::kThreeObstacles);
    const int N = def.N;
    {  // example_unicycle_test.cpp:86-106
      problem::Problem prob = def.MakeProblem(true);
      augmented_lagrangian::AugmentedLagrangianiLQR<NStates, NControls> solver_al(prob);
      solver_al.SetTrajectory(std::make_shared<altro::Trajectory<NStates, NControls>>(def.InitialTrajectory<NStates, NControls>()));
      solver_al.GetOptions().initial_penalty = 10.0;
      solver_al.GetOptions().verbose = altro::LogLevel::kDebug;
      solver_al.Init();
      auto pen = solver_al.GetALCost(0)->GetInequalityConstraints()[0]->GetPenalty();
      EXPECT(!pen.empty());
      for (double v : pen) EXPECT(v == 10.0);
      EXPECT(solver_al.GetALCost(N)->GetEqualityConstraints().size() == 1);
      EXPECT(solver_al.GetALCost(N)->GetEqualityConstraints()[0]->OutputDimension() == NStates);
      EXPECT(solver_al.GetALCost(N)->NumConstraints() == solver_al.NumConstraints(N));
    }
    {  // auglag_test.cpp:250-275
      problem::Problem prob = def.MakeProblem(true);
      augmented_lagrangian::AugmentedLagrangianiLQR<NStates, NControls> alsolver(prob);
      std::shared_ptr<altro::Trajectory<NStates, NControls>> Z =
          std::make_shared<altro::Trajectory<NStates, NControls>>(def.InitialTrajectory<NStates, NControls>());
      alsolver.SetTrajectory(Z);
      std::shared_ptr<augmented_lagrangian::ALCost<NStates, NControls>> alcost_term = alsolver.GetALCost(N);
      std::shared_ptr<constraints::ConstraintValues<NStates, NControls, constraints::Equality>> goal_vals =
          alcost_term->GetEqualityConstraints()[0];
      for (double v : goal_vals->GetDuals()) EXPECT(v == 0.0);

      ilqr::iLQR<NStates, NControls>& ilqr_solver = alsolver.GetiLQRSolver();
      double J0 = ilqr_solver.Cost();
      double viol0 = alsolver.GetMaxViolation();
      ilqr_solver.Solve();
      double J = ilqr_solver.Cost();
      double viol = alsolver.GetMaxViolation();

      alsolver.UpdateDuals();
      alsolver.UpdatePenalties();
      double dual_norm = 0.0;
      for (double v : goal_vals->GetDuals()) dual_norm += v * v;  // the pointer taken before the solve sees the update
      std::printf("Goal Duals: %g %g %g\n", goal_vals->GetDuals()[0], goal_vals->GetDuals()[1], goal_vals->GetDuals()[2]);
      EXPECT(dual_norm > 0.0);
      double J_penalty = ilqr_solver.Cost();
      EXPECT(J_penalty > J);
      EXPECT(viol < viol0);
      EXPECT(J < J0);
      // lambda = lambda - rho * c for an equality (al_cost / constraint_values.hpp dual update): rho was 1 before UpdatePenalties
      const std::vector<double> pen = goal_vals->GetPenalty();
      std::printf("Goal penalty after UpdatePenalties: %g (J0 %g J %g J_penalty %g viol0 %g viol %g)\n", pen[0], J0, J, J_penalty, viol0, viol);
      for (double v : pen) EXPECT(v == pen[0] && v > 1.0);  // scaled up from the initial penalty
      alsolver.ResetDualVariables();
      for (double v : goal_vals->GetDuals()) EXPECT(v == 0.0);
      for (double v : alsolver.GetALCost(3)->GetInequalityConstraints()[0]->GetDuals()) EXPECT(v == 0.0);
    }
    {  // ilqr_class_test.cpp:72-82 (CopyFromProblem over the whole range), ilqr.hpp:163 (GetCosts)
      problem::Problem prob = def.MakeProblem(false);
      ilqr::iLQR<NStates, NControls> ilqr(N);
      ilqr.CopyFromProblem(prob, 0, N + 1);
      EXPECT(ilqr.NumSegments() == N);
      ilqr.SetTrajectory(std::make_shared<altro::Trajectory<NStates, NControls>>(def.InitialTrajectory<NStates, NControls>()));
      ilqr.Rollout();
      const double J = ilqr.Cost();
      const std::vector<double> costs = ilqr.GetCosts();
      EXPECT((int)costs.size() == N + 1);
      double sum = 0.0;
      for (double c : costs) sum += c;
      EXPECT(std::abs(sum - J) <= 1e-12 * std::abs(J));
      bool threw = false;
      try {
        ilqr::iLQR<NStates, NControls> part(N);
        part.CopyFromProblem(prob, 0, N);
      } catch (const std::runtime_error&) {
        threw = true;
      }
      EXPECT(threw);
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 2;
  }
  std::printf("al_cost_views: %d failures\n", failures);
  return failures == 0 ? 0 : 1;
}
