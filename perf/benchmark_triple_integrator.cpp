// perf/benchmark_triple_integrator.cpp — counterpart of the reference's perf/benchmark_triple_integrator.cpp:20-57
// on the MI355X solver: the unconstrained and the constrained dof-2 triple-integrator solve through the C++
// facade (profiler tree printed like perf/profiler_triple_integrator*.out), plus the batched run of BASELINE
// configs[1] (1024 instances, 51 knots, unconstrained iLQR).
//   usage: benchmark_triple_integrator [nruns] [batch]
#include <chrono>
#include <cstdio>
#include <string>

#include "altro/problems.hpp"

using namespace altro;

static void SolveTripleIntegrator(bool add_constraints, int nruns) {  // benchmark_triple_integrator.cpp:20-43
  problems::TripleIntegratorProblem<> prob_def;
  problem::Problem prob = prob_def.MakeProblem(add_constraints);
  augmented_lagrangian::AugmentedLagrangianiLQR<6, 2> solver(prob);
  solver.GetiLQRSolver().SetRecordCostToGo(false);
  auto traj_ptr = std::make_shared<Trajectory<6, 2>>(prob_def.InitialTrajectory());
  solver.SetTrajectory(traj_ptr);
  solver.GetOptions().profiler_enable = true;
  double best = 1e30;
  for (int r = 0; r < nruns; ++r) {
    *traj_ptr = prob_def.InitialTrajectory();
    const auto start = std::chrono::high_resolution_clock::now();
    solver.Solve();
    const auto stop = std::chrono::high_resolution_clock::now();
    best = std::min(best, std::chrono::duration<double, std::milli>(stop - start).count());
  }
  std::printf("%s triple integrator: iters = %d, outer = %d, status = %d, violation = %.3g, Total Compute Time: %.4f ms\n",
              add_constraints ? "Constrained" : "Unconstrained", solver.GetStats().iterations_total,
              solver.GetStats().iterations_outer, (int)solver.GetStatus(), solver.GetMaxViolation(), best);
  solver.PrintTimings(stdout);
}

static void SolveBatch(int B, int nruns) {  // BASELINE configs[1]
  problems::TripleIntegratorProblem<> def;
  def.MakeBatch(B);
  ilqr::iLQR<6, 2> solver(def.MakeProblem(false));
  solver.SetRecordCostToGo(false);
  solver.SetRecordHistory(false);
  auto traj = std::make_shared<Trajectory<6, 2>>(def.InitialTrajectory());
  solver.SetTrajectory(traj);
  for (int r = 0; r < nruns; ++r) {
    *traj = def.InitialTrajectory();
    solver.SetTrajectory(traj);
    solver.ResetStats();  // iLQR::Solve accumulates iterations_total across calls (quirk Q11)
    const auto start = std::chrono::high_resolution_clock::now();
    solver.Solve();
    const auto stop = std::chrono::high_resolution_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(stop - start).count();
    int solved = 0;
    long long iters = 0;
    for (const altro_stats& s : solver.GetStats().AllInstances()) {
      solved += (s.status_ilqr == 0);
      iters += s.iterations_total;
    }
    std::printf("batch %d run %d: solve + trajectory download %.3f ms, solved %d/%d, %lld instance-iterations -> %.0f trajectories/s\n",
                B, r, ms, solved, B, iters, solved / (ms * 1e-3));
  }
}

int main(int argc, char* argv[]) {
  const int nruns = argc > 1 ? std::stoi(argv[1]) : 3;
  const int batch = argc > 2 ? std::stoi(argv[2]) : 1024;
  try {
    SolveTripleIntegrator(false, nruns);
    SolveTripleIntegrator(true, nruns);
    SolveBatch(batch, nruns);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
