// A caller written the way the reference's own callers are written -- same include paths, same statements, same
// types by shared pointer -- compiled against include/ WITHOUT edits to the call sites.  What it mirrors:
//   perf/benchmark_unicycle.cpp:18-75          (SolveUnicycle / SolveUnicycleLoop: options the drivers set, by-value
//                                               InitialTrajectory(), InitialTrajectory<n, m>(), *traj_ptr = ...)
//   perf/benchmarks.hpp:15-22                  (SetProfilerOptions: profiler_output_to_file, log_directory, ...)
//   perf/benchmark_triple_integrator.cpp:17-45 (TripleIntegratorProblem<dof>, ProbType<dof>::NStates)
//   examples/problems/unicycle.cpp:55-82       (SetConstraint / SetCostFunction / SetDynamics by std::shared_ptr)
//   examples/problems/triple_integrator.hpp:46-88 (ConstraintPtr<Inequality>, DiscretizedModel<Model, Integrator>)
// Modulo Eigen (vectors are std::vector<double>) and fmt (printf).  tests/test_facade_compile.py compiles it with
// `g++ -fsyntax-only` on CPU; tests/test_facade_gpu.py builds and runs it on the GPU (50 iterations / 5 outer / kSolved,
// the reference's known answer for this driver's problem: test/examples/example_unicycle_test.cpp:65-80).
#include <chrono>
#include <cstdio>
#include <iostream>

#include "altro/augmented_lagrangian/al_solver.hpp"
#include "altro/common/solver_options.hpp"
#include "altro/ilqr/ilqr.hpp"
#include "examples/problems/triple_integrator.hpp"
#include "examples/problems/unicycle.hpp"

namespace altro {
namespace benchmarks {

constexpr const char* kLogDir = "";

template <class Solver>
void SetProfilerOptions(Solver& solver, const std::string& basename) {
  solver.GetOptions().profiler_enable = true;
  solver.GetOptions().profiler_output_to_file = true;
  solver.GetOptions().log_directory = kLogDir;
  solver.GetOptions().profile_filename = "profiler_" + basename + ".out";
}

int SolveUnicycle(int nthreads) {
  problems::UnicycleProblem prob_def;
  prob_def.SetScenario(problems::UnicycleProblem::kThreeObstacles);
  const bool add_constraints = true;
  problem::Problem prob = prob_def.MakeProblem(add_constraints);

  constexpr int NStates = problems::UnicycleProblem::NStates;
  constexpr int NControls = problems::UnicycleProblem::NControls;
  augmented_lagrangian::AugmentedLagrangianiLQR<NStates, NControls> solver(prob);
  std::shared_ptr<altro::Trajectory<NStates, NControls>> traj_ptr =
      std::make_shared<altro::Trajectory<NStates, NControls>>(prob_def.InitialTrajectory());
  solver.SetTrajectory(traj_ptr);

  SetProfilerOptions(solver, "unicycle");
  solver.SetPenalty(10.0);
  solver.GetOptions().verbose = LogLevel::kDebug;
  solver.GetOptions().nthreads = nthreads;

  auto start = std::chrono::high_resolution_clock::now();
  solver.Solve();
  auto stop = std::chrono::high_resolution_clock::now();

  std::chrono::microseconds duration = std::chrono::duration_cast<std::chrono::microseconds>(stop - start);
  std::printf("Total Compute Time: %.3f ms\n", duration.count() / 1000.0);
  std::printf("SolveUnicycle: iters = %d, outer = %d, status = %d\n", solver.GetStats().iterations_total,
              solver.GetStats().iterations_outer, static_cast<int>(solver.GetStatus()));
  return solver.GetStats().iterations_total;
}

void SolveUnicycleLoop(int nruns, int nthreads) {
  problems::UnicycleProblem prob_def;
  prob_def.SetScenario(problems::UnicycleProblem::kThreeObstacles);
  const bool add_constraints = true;
  problem::Problem prob = prob_def.MakeProblem(add_constraints);

  constexpr int NStates = problems::UnicycleProblem::NStates;
  constexpr int NControls = problems::UnicycleProblem::NControls;
  augmented_lagrangian::AugmentedLagrangianiLQR<NStates, NControls> solver(prob);
  std::shared_ptr<altro::Trajectory<NStates, NControls>> traj_ptr =
      std::make_shared<altro::Trajectory<NStates, NControls>>(prob_def.InitialTrajectory<NStates, NControls>());
  solver.SetTrajectory(traj_ptr);

  SetProfilerOptions(solver, "unicycle-loop");
  std::vector<std::chrono::duration<double, std::milli>> times;
  for (int iter = 0; iter < nruns; ++iter) {
    solver.SetPenalty(10.0);
    solver.GetOptions().verbose = LogLevel::kSilent;
    solver.GetOptions().nthreads = nthreads;
    *traj_ptr = prob_def.InitialTrajectory<NStates, NControls>();

    auto start = std::chrono::high_resolution_clock::now();
    solver.Solve();
    auto stop = std::chrono::high_resolution_clock::now();
    times.emplace_back(stop - start);
    std::printf("Iteration %d: Cost = %.12g, iters = %d, Time = %.3f ms\n", iter, solver.GetiLQRSolver().Cost(),
                solver.GetStats().iterations_total, times.back().count());
  }
}

template <int dof>
using ProbType = altro::problems::TripleIntegratorProblem<dof>;

void SolveTripleIntegrator(const bool add_constraints) {
  constexpr int dof = 2;
  problems::TripleIntegratorProblem<dof> prob_def;
  problem::Problem prob = prob_def.MakeProblem(add_constraints);

  constexpr int NStates = ProbType<dof>::NStates;
  constexpr int NControls = ProbType<dof>::NControls;
  augmented_lagrangian::AugmentedLagrangianiLQR<NStates, NControls> solver(prob);
  std::shared_ptr<altro::Trajectory<NStates, NControls>> traj_ptr =
      std::make_shared<altro::Trajectory<NStates, NControls>>(prob_def.InitialTrajectory());
  solver.SetTrajectory(traj_ptr);

  if (add_constraints) {
    SetProfilerOptions(solver, "triple_integrator");
  } else {
    SetProfilerOptions(solver, "triple_integrator_uncon");
  }
  auto start = std::chrono::high_resolution_clock::now();
  solver.Solve();
  auto stop = std::chrono::high_resolution_clock::now();
  std::chrono::duration<double> duration = std::chrono::duration_cast<std::chrono::milliseconds>(stop - start);
  std::printf("Total Compute Time: %.4f ms\n", duration.count());
  std::printf("SolveTripleIntegrator(%d): iters = %d, status = %d\n", add_constraints ? 1 : 0,
              solver.GetStats().iterations_total, static_cast<int>(solver.GetStatus()));
}

// the problem definition of examples/problems/unicycle.cpp:55-82, statement by statement, in a caller's own code
problem::Problem DefineByPointers(int N, const std::vector<double>& xf, const std::vector<double>& x0, float h) {
  using ModelType = problem::DiscretizedModel<examples::Unicycle>;
  problem::Problem prob(N);
  const std::vector<double> Q = problems::Diag(3, 1e-2 * h), R = problems::Diag(2, 1e-2 * h), Qf = problems::Diag(3, 100.0);
  const std::vector<double> uref = {0, 0};
  std::vector<double> lb = {-1.5, -1.5}, ub = {+1.5, +1.5};
  examples::CircleConstraint obstacles;
  obstacles.AddObstacle(0.75, 0.75, 0.2);

  for (int k = 1; k < N; ++k) {
    std::shared_ptr<altro::constraints::Constraint<altro::constraints::Inequality>> obs =
        std::make_shared<altro::examples::CircleConstraint>(obstacles);
    prob.SetConstraint(obs, k);
  }
  std::shared_ptr<examples::QuadraticCost> qcost, qterm;
  for (int k = 0; k < N; ++k) {
    qcost = std::make_shared<examples::QuadraticCost>(examples::QuadraticCost::LQRCost(Q, R, xf, uref));
    prob.SetCostFunction(qcost, k);
  }
  qterm = std::make_shared<examples::QuadraticCost>(examples::QuadraticCost::LQRCost(Qf, problems::Diag(2, 0.0), xf, uref, true));
  prob.SetCostFunction(qterm, N);

  ModelType model{examples::Unicycle()};
  for (int k = 0; k < N; ++k) {
    prob.SetDynamics(std::make_shared<ModelType>(model), k);
  }
  for (int k = 0; k < N; ++k) {
    prob.SetConstraint(std::make_shared<altro::examples::ControlBound>(lb, ub), k);
  }
  prob.SetConstraint(std::make_shared<examples::GoalConstraint>(xf), N);
  prob.SetInitialState(x0);

  // the interval overloads (problem.hpp:133-139, 187-193)
  std::vector<std::shared_ptr<examples::QuadraticCost>> costs(N, qcost);
  prob.SetCostFunction(costs);
  std::vector<std::shared_ptr<ModelType>> models(N, std::make_shared<ModelType>(model));
  prob.SetDynamics(models, 0);
  return prob;
}

}  // namespace benchmarks
}  // namespace altro

int main(int argc, char* argv[]) {
  int nruns = 1;
  if (argc > 1) {
    nruns = std::stoi(std::string(argv[1]));
  }
  int nthreads = 1;
  if (argc > 2) {
    nthreads = std::stoi(std::string(argv[2]));
  }
  try {
    const int iters = altro::benchmarks::SolveUnicycle(nthreads);
    altro::benchmarks::SolveUnicycleLoop(nruns, nthreads);
    altro::benchmarks::SolveTripleIntegrator(false);
    altro::benchmarks::SolveTripleIntegrator(true);

    // the pointer-built problem solves like the same problem built by value
    using namespace altro;
    problems::UnicycleProblem def;
    const float h = def.GetTimeStep();
    problem::Problem prob = benchmarks::DefineByPointers(def.N, def.xf, def.x0, h);
    augmented_lagrangian::AugmentedLagrangianiLQR<3, 2> solver(prob);
    altro::Trajectory<3, 2> Z(3, 2, def.N);  // the reference's (n, m, N) constructor
    for (int k = 0; k < def.N; ++k) Z.Control(k)[0] = Z.Control(k)[1] = 0.1;
    Z.SetUniformStep(h);
    altro::Trajectory<3, 2> Zcopy(Z);  // copy-constructible
    solver.SetTrajectory(std::make_shared<altro::Trajectory<3, 2>>(Zcopy));
    solver.Solve();
    std::printf("DefineByPointers: iters = %d, status = %d, constraints = %d, threads = %d\n", solver.GetStats().iterations_total,
                static_cast<int>(solver.GetStatus()), solver.NumConstraints(), solver.GetOptions().NumThreads());
    return iters == 50 && solver.GetStatus() == SolverStatus::kSolved ? 0 : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 2;
  }
}
