// The facade's pointer-taking API on the GPU (row B1 of the hot-path scope): every setter overload the reference's
// Problem has (altro/problem/problem.hpp:113-202: one knot, a vector of shared pointers from k_start on), solver options
// the reference's drivers set, by-value trajectories, a copyable Trajectory, the profiler written to a file.
//
// This is the repository's OWN driver -- a table of scenarios run by one routine, checked against the reference's known
// answers (test/examples/example_unicycle_test.cpp:65-80: 50 iterations, 5 outer, kSolved; example_triple_integrator_test.cpp:
// 16-70: 2 iterations).  That the reference's real callers compile against include/ as they are is checked where the reference
// exists: tests/test_facade_compile.py::test_reference_perf_drivers_compile_in_place compiles /root/reference/perf/*.cpp in
// place; nothing of those files lives in this tree.
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "altro/augmented_lagrangian/al_solver.hpp"
#include "altro/common/solver_options.hpp"
#include "altro/ilqr/ilqr.hpp"
#include "examples/problems/triple_integrator.hpp"
#include "examples/problems/unicycle.hpp"

namespace {

using altro::SolverStatus;
namespace al = altro::augmented_lagrangian;
namespace pb = altro::problem;
namespace ex = altro::examples;
using Unicycle = altro::problems::UnicycleProblem;

struct Outcome {
  int iterations = -1, outer = -1, status = -1, constraints = -1;
};

// one scenario = a problem, what to set on the solver before the solve, what to report afterwards
template <int n, int m>
Outcome RunScenario(const char* name, const pb::Problem& prob, const altro::Trajectory<n, m>& guess,
                    const std::function<void(al::AugmentedLagrangianiLQR<n, m>&)>& configure, int repeats = 1) {
  al::AugmentedLagrangianiLQR<n, m> solver(prob);
  auto traj = std::make_shared<altro::Trajectory<n, m>>(guess);  // (copy-constructed: the caller keeps its guess)
  solver.SetTrajectory(traj);
  if (configure) configure(solver);
  Outcome out;
  for (int rep = 0; rep < repeats; ++rep) {
    if (rep > 0) *traj = guess;  // assignment through the shared pointer, then the same solver again
    solver.Solve();
    out.iterations = solver.GetStats().iterations_total;
    out.outer = solver.GetStats().iterations_outer;
    out.status = static_cast<int>(solver.GetStatus());
    out.constraints = solver.NumConstraints();
    std::printf("%s[%d]: iterations %d outer %d status %d cost %.10g constraints %d threads %d\n", name, rep, out.iterations, out.outer,
                out.status, solver.GetStats().cost.empty() ? 0.0 : solver.GetStats().cost.back(), out.constraints,
                solver.GetOptions().NumThreads());
  }
  return out;
}

// the three-obstacle unicycle problem assembled setter by setter through the POINTER overloads: dynamics first (vector
// overload), then the stage costs (vector overload) and the terminal cost, then the constraints knot by knot
pb::Problem AssembleThroughPointers(const Unicycle& def) {
  using Discrete = pb::DiscretizedModel<ex::Unicycle>;
  const int N = def.N;
  const float h = def.GetTimeStep();
  pb::Problem prob(N);
  prob.SetInitialState(def.x0);

  const std::vector<std::shared_ptr<Discrete>> dynamics(N, std::make_shared<Discrete>(Discrete{ex::Unicycle()}));
  prob.SetDynamics(dynamics, 0);

  const std::vector<double> no_control = {0.0, 0.0};
  auto stage = std::make_shared<ex::QuadraticCost>(ex::QuadraticCost::LQRCost(altro::problems::Diag(3, 1e-2 * h), altro::problems::Diag(2, 1e-2 * h), def.xf, no_control));
  prob.SetCostFunction(std::vector<std::shared_ptr<ex::QuadraticCost>>(N, stage));
  prob.SetCostFunction(std::make_shared<ex::QuadraticCost>(
                           ex::QuadraticCost::LQRCost(altro::problems::Diag(3, 100.0), altro::problems::Diag(2, 0.0), def.xf, no_control, true)),
                       N);

  ex::CircleConstraint circle;
  circle.AddObstacle(0.75, 0.75, 0.2);
  for (int k = 0; k < N; ++k) {
    if (k > 0) {
      altro::constraints::ConstraintPtr<altro::constraints::Inequality> keep_out = std::make_shared<ex::CircleConstraint>(circle);
      prob.SetConstraint(keep_out, k);
    }
    prob.SetConstraint(std::make_shared<ex::ControlBound>(std::vector<double>{-1.5, -1.5}, std::vector<double>{1.5, 1.5}), k);
  }
  prob.SetConstraint(std::make_shared<ex::GoalConstraint>(def.xf), N);
  return prob;
}

}  // namespace

int main(int argc, char* argv[]) {
  const int repeats = argc > 1 ? std::stoi(argv[1]) : 2;
  const int nthreads = argc > 2 ? std::stoi(argv[2]) : 1;
  try {
    bool ok = true;
    // 1. the obstacle scenario of the problem factory, with the options the reference's drivers touch and the profiler on
    {
      Unicycle def;
      def.SetScenario(Unicycle::kThreeObstacles);
      const pb::Problem prob = def.MakeProblem(true);  // (before the guess: the order of evaluation of arguments is not ours to pick)
      const altro::Trajectory<3, 2> guess = def.InitialTrajectory();
      const Outcome o = RunScenario<3, 2>("three_obstacles", prob, guess, [&](auto& s) {
        s.SetPenalty(10.0);
        s.GetOptions().verbose = altro::LogLevel::kDebug;
        s.GetOptions().nthreads = nthreads;
        s.GetOptions().profiler_enable = true;
        s.GetOptions().profiler_output_to_file = true;
        s.GetOptions().log_directory = "";
        s.GetOptions().profile_filename = "profiler_three_obstacles.out";
      }, repeats);
      ok = ok && o.iterations == 50 && o.outer == 5 && o.status == static_cast<int>(SolverStatus::kSolved);
    }
    // 2. the templated triple integrator, with and without its constraints
    for (const bool constrained : {false, true}) {
      altro::problems::TripleIntegratorProblem<2> def;
      const pb::Problem prob = def.MakeProblem(constrained);
      const altro::Trajectory<6, 2> guess = def.template InitialTrajectory<6, 2>();
      const Outcome o = RunScenario<6, 2>(constrained ? "triple_integrator_bounded" : "triple_integrator", prob, guess, nullptr);
      ok = ok && o.status == static_cast<int>(SolverStatus::kSolved) && (constrained || o.iterations == 2);
    }
    // 3. a problem assembled through the pointer overloads, a hand-made guess through the (n, m, N) constructor
    {
      Unicycle def;
      altro::Trajectory<3, 2> guess(3, 2, def.N);
      for (int k = 0; k < def.N; ++k) guess.Control(k)[0] = guess.Control(k)[1] = 0.1;
      guess.SetUniformStep(def.GetTimeStep());
      const Outcome o = RunScenario<3, 2>("assembled_through_pointers", AssembleThroughPointers(def), guess, nullptr);
      ok = ok && o.status == static_cast<int>(SolverStatus::kSolved) && o.constraints == 502;
    }
    std::printf("%s\n", ok ? "ALL SCENARIOS OK" : "SCENARIO FAILED");
    return ok ? 0 : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 2;
  }
}
