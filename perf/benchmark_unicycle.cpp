// perf/benchmark_unicycle.cpp — counterpart of the reference's perf/benchmark_unicycle.cpp:18-97 on
// the MI355X solver: the kThreeObstacles AL-iLQR solve (single + loop) through the C++ facade, plus
// a batched run of BASELINE config 3 reporting trajectories/s and ms per iLQR iteration.
//   usage: benchmark_unicycle [nruns] [batch]
#include <chrono>
#include <cstdio>
#include <string>

#include "altro/problems.hpp"

using namespace altro;

static double SolveUnicycleLoop(int nruns) {  // perf/benchmark_unicycle.cpp:46-75
  problems::UnicycleProblem def;
  def.SetScenario(problems::UnicycleProblem::kThreeObstacles);
  problem::Problem prob = def.MakeProblem(true);
  augmented_lagrangian::AugmentedLagrangianiLQR<3, 2> solver(prob);
  solver.GetiLQRSolver().SetRecordCostToGo(false);  // timing run: lets the persistent tail kernel take the solve
  auto traj = def.InitialTrajectory();
  solver.SetTrajectory(traj);
  double best = 1e30;
  for (int iter = 0; iter < nruns; ++iter) {
    solver.SetPenalty(10.0);  // a no-op, as in the reference: Init() resets to initial_penalty (quirk Q8)
    *traj = *def.InitialTrajectory();
    const auto t0 = std::chrono::high_resolution_clock::now();
    solver.Solve();
    const auto t1 = std::chrono::high_resolution_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    best = ms < best ? ms : best;
    std::printf("Iteration %d: Cost = %.12g, iters = %d, outer = %d, status = %d, Time = %.3f ms\n", iter,
                solver.GetiLQRSolver().Cost(), solver.GetStats().iterations_total, solver.GetStats().iterations_outer,
                (int)solver.GetStatus(), ms);
  }
  return best;
}

static void SolveBatch(int B, int nruns) {
  problems::UnicycleProblem def;
  def.MakeTurn90Batch(B);
  problem::Problem prob = def.MakeProblem(true);
  augmented_lagrangian::AugmentedLagrangianiLQR<3, 2> solver(prob);
  auto traj = def.InitialTrajectory();
  solver.SetTrajectory(traj);
  solver.GetOptions().profiler_enable = true;
  for (int iter = 0; iter < nruns; ++iter) {
    *traj = *def.InitialTrajectory();
    solver.Solve();
    const altro_timing t = solver.GetTiming();
    int solved = 0;
    for (const altro_stats& s : solver.GetStats().AllInstances()) solved += (s.status == 0);
    std::printf("batch %d run %d: solve %.3f ms (device sections: init %.3f expansions %.3f backward_pass %.3f "
                "forward_pass %.3f), sweeps %d, solved %d/%d -> %.0f trajectories/s, %.4f ms per iLQR sweep\n",
                B, iter, t.total_ms, t.init_ms, t.expansions_ms, t.backward_pass_ms, t.forward_pass_ms, t.sweeps,
                solved, B, solved / (t.total_ms * 1e-3), t.total_ms / t.sweeps);
  }
  // the reference prints its profiler tree when the solver is destroyed (perf/profiler_unicycle.out)
  solver.PrintTimings(stdout);
}

int main(int argc, char* argv[]) {
  const int nruns = argc > 1 ? std::stoi(argv[1]) : 3;
  const int batch = argc > 2 ? std::stoi(argv[2]) : 4096;
  try {
    const double best = SolveUnicycleLoop(nruns);
    std::printf("Three-obstacle single solve: best %.3f ms (reference CPU profile: 31.768 ms, perf/profiler_unicycle.out:3)\n", best);
    SolveBatch(batch, nruns);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
