// perf/benchmark_expansions.cpp — counterpart of the reference's perf/benchmark_expansions.cpp:22-98.
// The reference times iLQR::UpdateExpansions (cost / AL expansion + RK4 Jacobian of every knot point,
// ilqr.hpp:350-358, 670-677) serially and over its thread pool with different task decompositions.  Here every
// (instance, knot) pair is one GPU thread of k_expansions, so the sweep is over the BATCH size instead of the
// thread count: time per call, per knot-point expansion, and the bytes the kernel moves against HBM peak.
//   usage: benchmark_expansions [nruns]
#include <chrono>
#include <cstdio>
#include <string>
#include <vector>

#include "altro/problems.hpp"

using namespace altro;

static double TimeExpansions(ilqr::iLQR<3, 2>& solver, int nruns) {
  solver.UpdateExpansions();  // warm-up
  const auto start = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < nruns; ++i) solver.UpdateExpansions();
  const auto stop = std::chrono::high_resolution_clock::now();
  return std::chrono::duration<double, std::micro>(stop - start).count() / nruns;
}

int main(int argc, char* argv[]) {
  const int nruns = argc > 1 ? std::stoi(argv[1]) : 100;
  try {
    double t1 = 0.0;
    for (int B : {1, 64, 1024, 4096, 16384}) {
      problems::UnicycleProblem def;  // benchmark_expansions.cpp:42-47: kTurn90, N = 100, AL cost
      def.N = 100;
      def.MakeTurn90Batch(B);
      problem::Problem prob = augmented_lagrangian::BuildAugLagProblem<3, 2>(def.MakeProblem());
      ilqr::iLQR<3, 2> solver(prob);
      solver.SetRecordCostToGo(false);
      solver.SetRecordHistory(false);
      solver.SetTrajectory(std::make_shared<Trajectory<3, 2>>(def.InitialTrajectory()));
      solver.Rollout();
      solver.SolveSetup();
      const double us = TimeExpansions(solver, nruns);
      if (B == 1) t1 = us;
      // algorithmic bytes of the expansion step: per stage knot read n+m+2p, write S+1
      const double bytes = ((100.0 * ((3 + 2 + 2 * 4) + 40)) + (3 + 2 * 3 + 13)) * 8.0 * B;
      std::printf("batch %6d: UpdateExpansions %9.2f us per call (host-synchronous), %8.4f us per instance, %7.2f ns per "
                  "knot-point expansion, %7.1f GB/s algorithmic (HBM peak 8000), %.1fx the single-instance call\n",
                  B, us, us / B, 1e3 * us / (B * 101.0), bytes / (us * 1e-6) / 1e9, us / t1);
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
