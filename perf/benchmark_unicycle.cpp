// perf/benchmark_unicycle.cpp — counterpart of the reference's perf/benchmark_unicycle.cpp:18-97 on
// the MI355X solver: the kThreeObstacles AL-iLQR solve (single + loop) through the C++ facade, plus
// a batched run of BASELINE config 3 reporting trajectories/s and ms per iLQR iteration.
//   usage: benchmark_unicycle [nruns] [batch] [--gpus N]
// --gpus N: the same seeded batch sharded over N GPUs of this node through altro::BatchGroup (one solver and one host
// thread per device, no data-path collective, ONE RCCL all-gather of the 32-byte result records per solve).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>

#include "altro/group.hpp"
#include "altro/problems.hpp"

using namespace altro;

static double SolveUnicycleLoop(int nruns) {  // perf/benchmark_unicycle.cpp:46-75
  problems::UnicycleProblem def;
  def.SetScenario(problems::UnicycleProblem::kThreeObstacles);
  problem::Problem prob = def.MakeProblem(true);
  augmented_lagrangian::AugmentedLagrangianiLQR<3, 2> solver(prob);
  solver.GetiLQRSolver().SetRecordCostToGo(false);  // timing run: lets the persistent tail kernel take the solve
  auto traj = std::make_shared<Trajectory<3, 2>>(def.InitialTrajectory());
  solver.SetTrajectory(traj);
  double best = 1e30;
  for (int iter = 0; iter < nruns; ++iter) {
    solver.SetPenalty(10.0);  // a no-op, as in the reference: Init() resets to initial_penalty (quirk Q8)
    *traj = def.InitialTrajectory();
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
  auto traj = std::make_shared<Trajectory<3, 2>>(def.InitialTrajectory());
  solver.SetTrajectory(traj);
  solver.GetOptions().profiler_enable = true;
  for (int iter = 0; iter < nruns; ++iter) {
    *traj = def.InitialTrajectory();
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

// BASELINE configs[3] across the GPUs of one node: `batch` instances per GPU (weak scaling), block split of the seeded
// global batch, per-GPU wall times and the gathered records.
static int SolveSharded(int gpus, int batch_per_gpu, int nruns) {
  std::vector<int> devices(gpus);
  for (int i = 0; i < gpus; ++i) devices[i] = i;
  const int total = gpus * batch_per_gpu;
  std::vector<std::unique_ptr<augmented_lagrangian::AugmentedLagrangianiLQR<3, 2>>> solvers;
  std::vector<std::shared_ptr<Trajectory<3, 2>>> trajs;
  std::vector<problems::UnicycleProblem> defs(gpus);
  for (int part = 0; part < gpus; ++part) {
    problems::UnicycleProblem& def = defs[part];
    def.MakeThreeObstaclesBatch(total);  // the global batch, then this device's block of it
    const auto range = BatchGroup::ShardRange(total, gpus, part);
    def.TakeShard(range.first, range.second);
    problem::Problem prob = def.MakeProblem(true);
    solvers.push_back(std::make_unique<augmented_lagrangian::AugmentedLagrangianiLQR<3, 2>>(prob, ALTRO_F32, devices[part]));
    solvers.back()->GetiLQRSolver().SetRecordCostToGo(false);
    trajs.push_back(std::make_shared<Trajectory<3, 2>>(def.InitialTrajectory()));
    solvers.back()->SetTrajectory(trajs.back());
    solvers.back()->NumConstraints();  // (creates the device state -- and the solver's streams -- before RCCL's)
  }
  BatchGroup group(devices);
  for (int part = 0; part < gpus; ++part) group.Attach(part, *solvers[part]);
  int failures = 0;
  for (int run = 0; run < nruns; ++run) {
    for (int part = 0; part < gpus; ++part) *trajs[part] = defs[part].InitialTrajectory();
    const auto t0 = std::chrono::high_resolution_clock::now();
    group.Solve();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    const std::vector<double> rec = group.Results(0);
    int solved = 0;
    for (int i = 0; i < total; ++i) solved += rec[4 * (size_t)i + 3] == 0.0;
    // the gathered records are the per-solver statistics, instance for instance
    size_t o = 0;
    for (int part = 0; part < gpus; ++part)
      for (const altro_stats& st : solvers[part]->GetStats().AllInstances()) {
        failures += !(rec[4 * o + 0] == st.cost && rec[4 * o + 2] == (double)st.iterations_total && rec[4 * o + 3] == (double)st.status);
        ++o;
      }
    std::printf("%d GPU(s) x %d instances, run %d: %.3f ms (exchange %.3f ms), solved %d/%d -> %.0f trajectories/s, "
                "records %s the per-solver statistics\n",
                gpus, batch_per_gpu, run, ms, group.GatherMilliseconds(), solved, total, solved / (ms * 1e-3),
                failures ? "DIFFER from" : "match");
    if (run == nruns - 1) {
      // the optional second collective: every device receives all trajectories; compared with each solver's own
      group.GatherTrajectories();
      const auto XU = group.Trajectories(0);
      int traj_failures = 0;
      size_t inst = 0;
      for (int part = 0; part < gpus; ++part) {  // (Wait() has copied each solver's solution into its trajectory)
        for (int b = 0; b < trajs[part]->BatchSize(); ++b, ++inst)
          for (int k = 0; k <= 100; ++k)
            for (int i = 0; i < 3; ++i) traj_failures += XU.first[(inst * 101 + k) * 3 + i] != trajs[part]->State(k, b)[i];
      }
      std::printf("trajectory all-gather: %.3f ms for %zu doubles per device, trajectories %s the per-solver ones\n",
                  group.TrajectoryGatherMilliseconds(), XU.first.size() + XU.second.size(), traj_failures ? "DIFFER from" : "match");
      failures += traj_failures;
    }
  }
  return failures;
}

int main(int argc, char* argv[]) {
  int gpus = 0;
  for (int i = 1; i + 1 < argc; ++i)
    if (!std::strcmp(argv[i], "--gpus")) {
      gpus = std::stoi(argv[i + 1]);
      for (int j = i; j + 2 < argc; ++j) argv[j] = argv[j + 2];
      argc -= 2;
      break;
    }
  const int nruns = argc > 1 ? std::stoi(argv[1]) : 3;
  const int batch = argc > 2 ? std::stoi(argv[2]) : 4096;
  if (gpus > 0) {
    try {
      return SolveSharded(gpus, batch, nruns) ? 2 : 0;
    } catch (const std::exception& e) {
      std::fprintf(stderr, "error: %s\n", e.what());
      return 1;
    }
  }
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
