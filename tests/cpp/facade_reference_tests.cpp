// tests/cpp/facade_reference_tests.cpp — the reference's own gtest cases for the hot path, replayed through
// the C++ facade (include/altro/) on the MI355X solver.  Each CASE cites the test it replays; the expected
// values are the reference's constants (tests/golden/reference_constants.json holds the same data).
//   build: make -C tests/cpp        run: tests/cpp/facade_reference_tests   (needs a GPU; exit code = failures)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "altro/problems.hpp"

using namespace altro;

static int g_failed = 0, g_checks = 0;
#define EXPECT(cond)                                                                   \
  do {                                                                                 \
    ++g_checks;                                                                        \
    if (!(cond)) {                                                                     \
      ++g_failed;                                                                      \
      std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);                  \
    }                                                                                  \
  } while (0)
#define CASE(name) std::printf("[ RUN ] %s\n", name)

static bool IsApprox(const std::vector<double>& a, const std::vector<double>& b, double prec) {
  // Eigen::isApprox: ||a - b|| <= prec * min(||a||, ||b||)
  double d = 0, na = 0, nb = 0;
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); ++i) {
    d += (a[i] - b[i]) * (a[i] - b[i]);
    na += a[i] * a[i];
    nb += b[i] * b[i];
  }
  return std::sqrt(d) <= prec * std::sqrt(std::min(na, nb));
}

// ---- test/ilqr/unicycle_ilqr_test.cpp ----------------------------------------------------------------------
static void UnicycleiLQRTest() {
  problems::UnicycleProblem def;
  CASE("UnicycleiLQRTest.BuildProblem (unicycle_ilqr_test.cpp:27-30)");
  EXPECT(def.MakeProblem().IsFullyDefined());

  CASE("UnicycleiLQRTest.Initialization (:32-37)");
  {
    ilqr::iLQR<3, 2> solver = def.MakeSolver();
    solver.Rollout();
    EXPECT(std::abs(solver.Cost() - 259.27636137767087) < 1e-5);
  }
  CASE("UnicycleiLQRTest.BackwardPass (:39-54)");
  {
    ilqr::iLQR<3, 2> solver = def.MakeSolver();
    solver.Rollout();
    solver.UpdateExpansions();
    solver.BackwardPass();
    EXPECT(IsApprox(solver.GetKnotPointFunction(0).GetCostToGoGradient(),
                    {0.024904637422419617, -0.46496022574032614, -0.0573096310550007}, 1e-5));
    EXPECT(IsApprox(solver.GetKnotPointFunction(0).GetFeedforwardGain(), {-2.565783457444465, 5.514158930898376}, 1e-5));
    // KnotPointFunctions::GetCostExpansion (knot_point_function_type.hpp:249): lxx of a stage knot is Q = 1e-2 h I
    const auto ce = solver.GetKnotPointFunction(3).GetCostExpansion();
    const double q = 1e-2 * (double)def.GetTimeStep();
    EXPECT(std::abs(ce.dxdx()[0] - q) < 1e-15 && std::abs(ce.dxdx()[4] - q) < 1e-15 && ce.dxdx()[1] == 0.0);
    EXPECT(ce.dudu().size() == 4 && std::abs(ce.dudu()[3] - q) < 1e-15 && ce.dx().size() == 3 && ce.du().size() == 2);
  }
  CASE("UnicycleiLQRTest.ForwardPass (:56-65)");
  {
    ilqr::iLQR<3, 2> solver = def.MakeSolver();
    solver.Rollout();
    solver.UpdateExpansions();
    solver.BackwardPass();
    const double J0 = solver.Cost();
    solver.ForwardPass();
    EXPECT(solver.Cost() < J0);
    EXPECT(solver.GetStats().alpha[0] == 0.0625);
  }
  CASE("UnicycleiLQRTest.TwoSteps (:67-88)");
  {
    ilqr::iLQR<3, 2> solver = def.MakeSolver();
    solver.Rollout();
    solver.UpdateExpansions();
    solver.BackwardPass();
    solver.ForwardPass();
    solver.UpdateExpansions();
    solver.BackwardPass();
    EXPECT(IsApprox(solver.GetKnotPointFunction(0).GetCostToGoGradient(),
                    {-0.0015143873973949232, -0.07854630832127288, -0.017945283678268698}, 1e-5));
    EXPECT(IsApprox(solver.GetKnotPointFunction(0).GetFeedforwardGain(), {0.21887571453613042, 1.3097976615154625}, 1e-5));
    solver.ForwardPass();
    EXPECT(solver.Cost() - 62.773696055304384 < 1e-5);
  }
  CASE("UnicycleiLQRTest.FullSolve (:90-100)");
  {
    ilqr::iLQR<3, 2> solver = def.MakeSolver();
    solver.Solve();
    EXPECT(solver.GetStats().iterations_inner == 9);
    EXPECT(solver.GetStatus() == SolverStatus::kSolved);
    EXPECT(std::abs(solver.Cost() - 0.0387016567) < 1e-5);
    EXPECT(solver.GetStats().gradient.back() < solver.GetOptions().gradient_tolerance);
    // the vectors hold one row per iteration plus the row NewIteration opened (solver_stats.cpp:54-66)
    EXPECT(solver.GetStats().alpha.size() == 10 && solver.GetStats().cost.size() == 10);
    EXPECT(solver.GetStats().alpha[0] == 0.0625);
  }
  CASE("UnicycleiLQRTest.AugLagForwardPass (:102-113)");
  {
    ilqr::iLQR<3, 2> solver = def.MakeSolver(true);
    solver.Rollout();
    solver.UpdateExpansions();
    solver.BackwardPass();
    const double J0 = solver.Cost();
    solver.ForwardPass();
    EXPECT(solver.Cost() < J0);
    EXPECT(solver.GetStats().alpha[0] == 0.0625);
  }
  CASE("UnicycleiLQRTest.AugLagFullSolve (:115-144)");
  {
    ilqr::iLQR<3, 2> solver = def.MakeSolver(true);
    solver.Solve();
    EXPECT(solver.GetStats().iterations_inner == 10);
    EXPECT(solver.GetStatus() == SolverStatus::kSolved);
    EXPECT(std::abs(solver.Cost() - 0.03893427133384412) / 0.03893427133384412 < 1e-6);
  }
}

// ---- test/augmented_lagrangian/auglag_test.cpp ---------------------------------------------------------------
static void AugLagTest() {
  problems::UnicycleProblem def;
  const int N = def.N;
  CASE("AugLagTest.InitializeAndSolve (auglag_test.cpp:325-351)");
  {
    problem::Problem prob = def.MakeProblem();
    augmented_lagrangian::AugmentedLagrangianiLQR<3, 2> alsolver(N);
    bool threw = false;
    try {
      alsolver.NumConstraints();  // "Cannot query the number of constraints before initializing the solver ..."
    } catch (const std::runtime_error&) {
      threw = true;
    }
    EXPECT(threw);
    alsolver.InitializeFromProblem(prob);
    EXPECT(alsolver.NumConstraints() == 4 * N + 3 && alsolver.NumConstraints(0) == 4 && alsolver.NumConstraints(N) == 3);
    auto Z = std::make_shared<Trajectory<3, 2>>(def.InitialTrajectory());
    alsolver.SetTrajectory(Z);
    alsolver.GetOptions().constraint_tolerance = 1e-6;
    alsolver.Solve();
    EXPECT(alsolver.GetStats().iterations_total == 14);
    EXPECT(alsolver.GetStats().iterations_outer == 5);
    EXPECT(std::abs(alsolver.GetiLQRSolver().Cost() - 0.03893465058924039) < 1e-12);
    EXPECT(alsolver.GetStatus() == SolverStatus::kSolved);
    EXPECT(alsolver.GetMaxViolation() < alsolver.GetOptions().constraint_tolerance);
    // SolverStats vectors: one row per inner iteration + the open row; violations / max_penalty per AL iteration
    EXPECT(alsolver.GetStats().cost.size() == 15);
    EXPECT(alsolver.GetStats().max_penalty.back() == 1e4);
  }
  CASE("AugLagTest.SolveTwice (:353-380)");
  {
    problem::Problem prob = def.MakeProblem();
    augmented_lagrangian::AugmentedLagrangianiLQR<3, 2> alsolver(N);
    alsolver.InitializeFromProblem(prob);
    auto Z = std::make_shared<Trajectory<3, 2>>(def.InitialTrajectory());
    alsolver.SetTrajectory(Z);
    alsolver.GetOptions().constraint_tolerance = 1e-6;
    alsolver.Solve();
    *Z = def.InitialTrajectory();
    alsolver.Solve();
    EXPECT(alsolver.GetStats().iterations_total == 14);
    EXPECT(alsolver.GetStats().iterations_outer == 5);
    EXPECT(std::abs(alsolver.GetiLQRSolver().Cost() - 0.03893465058924039) < 1e-12);
    EXPECT(alsolver.GetStatus() == SolverStatus::kSolved);
  }
  CASE("AugLagTest.PrintViolations (:382-399)");
  {
    problem::Problem prob = def.MakeProblem();
    augmented_lagrangian::AugmentedLagrangianiLQR<3, 2> alsolver(N);
    alsolver.InitializeFromProblem(prob);
    alsolver.SetTrajectory(std::make_shared<Trajectory<3, 2>>(def.InitialTrajectory()));
    alsolver.Solve();
    std::vector<constraints::ConstraintInfo> coninfo = alsolver.GetConstraintInfo();
    EXPECT((int)coninfo.size() == N + 1);
    EXPECT(coninfo[0].index == 0);
    EXPECT(coninfo[0].label == "Control Bound");
    EXPECT(coninfo[0].type == "Inequality Constraint" && coninfo[0].violation.size() == 4);
    coninfo = alsolver.GetConstraintInfo(/*sort=*/true);
    EXPECT(coninfo[0].index == alsolver.NumSegments());
    EXPECT(coninfo[0].label == "Goal Constraint");
    EXPECT(std::abs(coninfo[0].MaxViolation() - alsolver.GetMaxViolation()) < 1e-15);
    alsolver.PrintViolations(true, 4, stdout);
  }
  CASE("AugLagTest.TwoSolves (:249-287): second inner solve after a dual and penalty update takes one iteration");
  {
    auto alsolver = def.MakeALSolver();
    alsolver.GetiLQRSolver().Solve();
    EXPECT(alsolver.GetiLQRSolver().GetStats().iterations_inner == 10);
    EXPECT(std::abs(alsolver.GetMaxViolation() - 0.00017691645708972636) / 0.00017691645708972636 < 1e-6);
    alsolver.UpdateDuals();
    alsolver.UpdatePenalties();
    alsolver.GetiLQRSolver().Solve();
    EXPECT(alsolver.GetiLQRSolver().GetStats().iterations_inner == 1);
    EXPECT(std::abs(alsolver.MaxViolation() - 6.26e-5) / 6.26e-5 < 0.1);
  }
}

// ---- test/examples/example_unicycle_test.cpp, example_triple_integrator_test.cpp ---------------------------------
static void ExampleTests() {
  CASE("ExampleUnicycle three obstacles (example_unicycle_test.cpp:18-50, 69-89)");
  {
    problems::UnicycleProblem def;
    def.SetScenario(problems::UnicycleProblem::kThreeObstacles);
    ilqr::iLQR<3, 2> plain = def.MakeSolver();
    EXPECT(std::abs(plain.Cost() - 133.1151550141444) < 1e-6);
    ilqr::iLQR<3, 2> al = def.MakeSolver(true);
    EXPECT(std::abs(al.Cost() - 141.9639680271223) < 1e-6);
    auto solver = def.MakeALSolver();
    solver.SetPenalty(10.0);
    EXPECT(std::abs(solver.GetiLQRSolver().Cost() - 221.6032851439234) < 1e-6);
    solver.Solve();
    EXPECT(solver.GetStatus() == SolverStatus::kSolved);
    EXPECT(solver.MaxViolation() < 1e-4);
    EXPECT(solver.GetStats().cost_decrease.back() < 1e-4);
    EXPECT(solver.GetStats().gradient.back() < 1e-2);
    EXPECT(solver.GetStats().iterations_total == 50 && solver.GetStats().iterations_outer == 5);
  }
  CASE("ExampleTripleIntegrator (example_triple_integrator_test.cpp:16-70)");
  {
    problems::TripleIntegratorProblem<> def;
    augmented_lagrangian::AugmentedLagrangianiLQR<6, 2> uncon(def.MakeProblem(false));
    uncon.SetTrajectory(std::make_shared<Trajectory<6, 2>>(def.InitialTrajectory()));
    uncon.Solve();
    EXPECT(uncon.GetStats().iterations_total == 2 && uncon.GetStatus() == SolverStatus::kSolved);
    augmented_lagrangian::AugmentedLagrangianiLQR<6, 2> con(def.MakeProblem(true));
    auto Z = std::make_shared<Trajectory<6, 2>>(def.InitialTrajectory());
    con.SetTrajectory(Z);
    con.Solve();
    EXPECT(con.GetStatus() == SolverStatus::kSolved);
    EXPECT(con.MaxViolation() < 1e-4);
    EXPECT(std::abs(Z->Control(0)[0] - 100.0) < 1e-6 && std::abs(Z->Control(0)[1] - 200.0) < 1e-6);
    EXPECT(std::abs(Z->Control(def.N - 1)[0] - 100.0) < 1e-6 && std::abs(Z->Control(def.N - 1)[1] - 200.0) < 1e-6);
  }
  CASE("Facade: calls before the trajectory exists keep the problem usable (ADVICE r1: step set late)");
  {
    problems::UnicycleProblem def;
    augmented_lagrangian::AugmentedLagrangianiLQR<3, 2> solver(def.MakeProblem());
    EXPECT(solver.NumConstraints(0) == 4);  // creates the device state before any trajectory / step is known
    EXPECT(solver.GetDuals().size() == (size_t)solver.NumConstraints());
    bool threw = false;
    try {
      solver.GetiLQRSolver().Rollout();  // no trajectory, hence no step: refused, not integrated with h = 0
    } catch (const std::runtime_error&) {
      threw = true;
    }
    EXPECT(threw);
    solver.SetTrajectory(std::make_shared<Trajectory<3, 2>>(def.InitialTrajectory()));
    solver.Solve();
    EXPECT(solver.GetStatus() == SolverStatus::kSolved && solver.GetStats().iterations_total == 11);
  }
}

// ---- the reference's plug-in classes (ContinuousDynamics, CostFunction, Constraint<ConType>) on the facade ----------
// tests/models/cartpole_track.hpp defines all three; tests/test_user_model_gpu.py checks the same problem against
// the oracle, here the facade path (examples::UserModel / UserCost / UserConstraint -> Problem -> solver) is driven.
static void UserFunctorTests() {
  CASE("User-defined dynamics, cost function and constraint through the facade (dynamics.hpp:59-95, "
       "costfunction.hpp:52-73, constraint.hpp:173-202)");
  std::ifstream f(TRACK_SOURCE_PATH);
  std::stringstream src;
  src << f.rdbuf();
  EXPECT(!src.str().empty());
  const int N = 60;
  const double goal = 1.2, sway = 0.04, h = 0.05;
  examples::UserModel model("cartpole_track", src.str(), 4, 1);
  problem::Problem prob(N);
  const examples::UserCost stage({goal, 1e-1 * h, 2.0 * h, 1e-1 * h, 1e-1 * h, 1e-2 * h});
  const examples::UserCost term({goal, 100.0, 100.0, 100.0, 100.0, 0.0});
  for (int k = 0; k < N; ++k) {
    prob.SetDynamics(problem::DiscretizedModel<examples::UserModel>(model), k);
    prob.SetCostFunction(stage, k);
    prob.SetConstraint(examples::ControlBound({-3.0}, {3.0}), k);
  }
  prob.SetCostFunction(term, N);
  for (int k = 1; k <= N; ++k) prob.SetConstraint(examples::UserConstraint({-sway, sway}, 2), k);
  prob.SetConstraint(examples::GoalConstraint({goal, 0.0, 0.0, 0.0}), N);
  prob.SetInitialState({0.0, 0.0, 0.0, 0.0});
  EXPECT(prob.IsFullyDefined());
  EXPECT(prob.NumConstraints(0) == 2 && prob.NumConstraints(1) == 4 && prob.NumConstraints(N) == 6);
  augmented_lagrangian::AugmentedLagrangianiLQR<4, 1> solver(prob);
  auto Z = std::make_shared<Trajectory<4, 1>>(N);
  Z->SetUniformStep(static_cast<float>(h));
  solver.SetTrajectory(Z);
  solver.Solve();
  EXPECT(solver.GetStatus() == SolverStatus::kSolved);
  EXPECT(solver.MaxViolation() < 1e-4);
  double worst = 0.0;
  for (int k = 0; k <= N; ++k) worst = std::max(worst, std::abs(0.5 * std::sin(Z->State(k)[1])));
  EXPECT(worst > sway - 1e-3 && worst < sway + 1e-3);  // the sway limit is active and held
  EXPECT(std::abs(Z->State(N)[0] - goal) < 1e-3 && std::abs(Z->State(N)[2]) < 1e-3);
  const auto info = solver.GetConstraintInfo();
  bool seen = false;
  for (const auto& ci : info) seen = seen || ci.label == "User Constraint";
  EXPECT(seen);
}

// Several CostFunction / Constraint subclasses in one problem (problem.hpp:66-133): tests/models/cartpole_multi.hpp
// lists two cost classes and three constraint classes; tests/test_user_types_gpu.py checks the same problem against the
// oracle, here it goes through the facade (examples::UserCost / UserConstraint with a `type`).
static void UserTypeListTests() {
  CASE("Two user cost classes and three user constraint classes knot by knot (problem.hpp:66-133)");
  std::ifstream f(MULTI_SOURCE_PATH);
  std::stringstream src;
  src << f.rdbuf();
  EXPECT(!src.str().empty());
  const int N = 60;
  const double goal = 1.2, sway = 0.05, vmax = 0.6, h = 0.05;
  examples::UserModel model("cartpole_multi", src.str(), 4, 1);
  problem::Problem prob(N);
  const examples::UserCost stage({goal, 1e-1 * h, 2.0 * h, 1e-1 * h, 1e-1 * h, 1e-2 * h}, -1, /*type=*/0);
  const examples::UserCost term({goal, 100.0, 100.0}, -1, /*type=*/1);
  for (int k = 0; k < N; ++k) {
    prob.SetDynamics(problem::DiscretizedModel<examples::UserModel>(model), k);
    prob.SetCostFunction(stage, k);
    prob.SetConstraint(examples::ControlBound({-3.0}, {3.0}), k);
  }
  prob.SetCostFunction(term, N);
  for (int k = 1; k < N; ++k) {
    prob.SetConstraint(examples::UserConstraint({-sway, sway}, 2, false, -1, "Sway Limit", /*type=*/0), k);
    prob.SetConstraint(examples::UserConstraint({vmax}, 1, false, -1, "Speed Limit", /*type=*/2), k);
  }
  prob.SetConstraint(examples::UserConstraint({goal}, 1, true, -1, "Tip At Goal", /*type=*/1), N);
  prob.SetInitialState({0.0, 0.0, 0.0, 0.0});
  EXPECT(prob.IsFullyDefined());
  EXPECT(prob.NumConstraints(0) == 2 && prob.NumConstraints(1) == 5 && prob.NumConstraints(N) == 1);
  augmented_lagrangian::AugmentedLagrangianiLQR<4, 1> solver(prob);
  auto Z = std::make_shared<Trajectory<4, 1>>(N);
  Z->SetUniformStep(static_cast<float>(h));
  solver.SetTrajectory(Z);
  solver.Solve();
  EXPECT(solver.GetStatus() == SolverStatus::kSolved);
  EXPECT(solver.MaxViolation() < 1e-4);
  double worst_sway = 0.0, worst_speed = 0.0;
  for (int k = 0; k <= N; ++k) {
    worst_sway = std::max(worst_sway, std::abs(0.5 * std::sin(Z->State(k)[1])));
    worst_speed = std::max(worst_speed, std::abs(Z->State(k)[2]));
  }
  EXPECT(worst_sway > sway - 1e-3 && worst_sway < sway + 1e-3);    // both inequality classes are active and held
  EXPECT(worst_speed > vmax - 1e-3 && worst_speed < vmax + 1e-3);
  EXPECT(std::abs(Z->State(N)[0] + 0.5 * std::sin(Z->State(N)[1]) - goal) < 1e-4);  // the equality class
  // a class index the source does not have is refused with the count of classes it has
  problem::Problem bad = prob;
  bad.SetCostFunction(examples::UserCost({goal, 100.0, 100.0}, -1, /*type=*/2), N);
  bool thrown = false;
  try {
    augmented_lagrangian::AugmentedLagrangianiLQR<4, 1> s2(bad);
    auto Z2 = std::make_shared<Trajectory<4, 1>>(N);
    Z2->SetUniformStep(static_cast<float>(h));
    s2.SetTrajectory(Z2);
    s2.Solve();
  } catch (const std::exception& e) {
    thrown = std::string(e.what()).find("2 cost type") != std::string::npos;
  }
  EXPECT(thrown);
}

// ---- Trajectory::SetStep / SetTime per knot (trajectory.hpp:119-120) and Problem::SetDynamics per knot ----------------
static void KnotTimeTests() {
  CASE("Trajectory::SetStep(k, h) on every knot: the general kernels reach the uniform-step solution (trajectory.hpp:119-130)");
  {
    problems::UnicycleProblem def;
    auto pa = def.MakeALSolver();
    auto pb = def.MakeALSolver();
    augmented_lagrangian::AugmentedLagrangianiLQR<3, 2>&a = pa, &b = pb;
    auto Za = std::make_shared<Trajectory<3, 2>>(def.InitialTrajectory());
    auto Zb = std::make_shared<Trajectory<3, 2>>(def.InitialTrajectory());
    EXPECT(Za->IsUniformStep());
    const int N = Zb->NumSegments();
    for (int k = 0; k < N; ++k) Zb->SetStep(k, Za->GetStep(k) * (k == 0 ? 1.0f : 1.0f));
    Zb->SetStep(0, Za->GetStep(0) * 2.0f);  // (leaves the uniform state) ...
    Zb->SetStep(0, Za->GetStep(0));         // ... and the same values again: still handed over per knot
    EXPECT(!Zb->IsUniformStep() && Zb->GetStep(N) == 0.0f && Zb->GetTime(3) == 3.0f * Za->GetStep(0));
    a.SetTrajectory(Za);
    b.SetTrajectory(Zb);
    a.Solve();
    b.Solve();
    EXPECT(a.GetStatus() == SolverStatus::kSolved && b.GetStatus() == SolverStatus::kSolved);
    EXPECT(a.GetStats().iterations_total == b.GetStats().iterations_total);
    double worst = 0.0;
    for (int k = 0; k <= N; ++k)
      for (int i = 0; i < 3; ++i) worst = std::max(worst, std::abs(Za->State(k)[i] - Zb->State(k)[i]));
    EXPECT(worst < 1e-9);
    // a genuinely non-uniform grid (finer at the start) still solves, to a different trajectory
    auto Zc = std::make_shared<Trajectory<3, 2>>(def.InitialTrajectory());
    float t = 0.0f;
    for (int k = 0; k < N; ++k) {
      const float h = Za->GetStep(0) * (0.5f + static_cast<float>(k) / static_cast<float>(N - 1));
      Zc->SetStep(k, h);
      Zc->SetTime(k, t);
      t += h;
    }
    Zc->SetTime(N, t);
    auto pc = def.MakeALSolver();
    augmented_lagrangian::AugmentedLagrangianiLQR<3, 2>& c = pc;
    c.SetTrajectory(Zc);
    c.Solve();
    EXPECT(c.GetStatus() == SolverStatus::kSolved && c.MaxViolation() < 1e-4);
    double diff = 0.0;
    for (int k = 0; k <= N; ++k) diff = std::max(diff, std::abs(Za->State(k)[0] - Zc->State(k)[0]));
    EXPECT(diff > 1e-3);
  }
  CASE("Problem::SetDynamics with a different model on another knot is refused (problem.hpp:155-166 keeps one per knot)");
  {
    problem::Problem prob(10);
    prob.SetDynamics(problem::DiscretizedModel<examples::TripleIntegrator>(examples::TripleIntegrator(1)), 0);
    prob.SetDynamics(problem::DiscretizedModel<examples::TripleIntegrator>(examples::TripleIntegrator(1)), 1);  // same model: fine
    bool threw = false;
    try {
      prob.SetDynamics(problem::DiscretizedModel<examples::TripleIntegrator>(examples::TripleIntegrator(2)), 2);
    } catch (const std::runtime_error&) {
      threw = true;
    }
    EXPECT(threw);
    threw = false;
    try {
      prob.SetDynamics(problem::DiscretizedModel<examples::Unicycle>(examples::Unicycle()), 3);
    } catch (const std::runtime_error&) {
      threw = true;
    }
    EXPECT(threw);
  }
}

// ---- what else Problem::SetDynamics accepts (VERDICT r3 missing #1 / #2): a time-varying model through the facade, a
//      different model per knot, DiscretizedModel<Model, ExplicitEuler> ----------------------------------------------------
static std::string ReadSource(const char* path) {
  std::ifstream f(path);
  std::stringstream src;
  src << f.rdbuf();
  return src.str();
}
static void CartpoleProblem(problem::Problem& prob, int N, double goal, double h) {
  const std::vector<double> Q = {1e-1 * h, 0, 0, 0, 0, 1e-1 * h, 0, 0, 0, 0, 1e-1 * h, 0, 0, 0, 0, 1e-1 * h};
  std::vector<double> Qf(16, 0.0);
  for (int i = 0; i < 4; ++i) Qf[i * 5] = 100.0;
  const std::vector<double> xf = {goal, 0.0, 0.0, 0.0};
  for (int k = 0; k < N; ++k) {
    prob.SetCostFunction(examples::QuadraticCost::LQRCost(Q, {1e-2 * h}, xf, {0.0}), k);
    prob.SetConstraint(examples::ControlBound({-3.0}, {3.0}), k);
  }
  prob.SetCostFunction(examples::QuadraticCost::LQRCost(Qf, {0.0}, xf, {0.0}, true), N);
  prob.SetConstraint(examples::GoalConstraint(xf), N);
  prob.SetInitialState({0.0, 0.0, 0.0, 0.0});
}
static void cartpole_f(const double* x, double F, double* xd) {  // tests/models/cartpole.hpp
  const double mc = 1.0, mp = 0.2, l = 0.5, g = 9.81, s = std::sin(x[1]), c = std::cos(x[1]), q = x[3], D = mc + mp * s * s;
  xd[0] = x[2];
  xd[1] = q;
  xd[2] = (F + mp * s * (l * q * q + g * c)) / D;
  xd[3] = (-F * c - mp * l * q * q * c * s - (mc + mp) * g * s) / (l * D);
}
static void DynamicsKindTests() {
  const int N = 60;
  const float hf = 0.05f;
  const double h = hf;
  CASE("A time-varying user model through the facade: the solver is built from the problem BEFORE the trajectory (and its "
       "step) exists (ContinuousDynamics::Evaluate(x, u, t, xdot), dynamics.hpp:59-95; ADVICE r3)");
  {
    examples::UserModel wind("cartpole_wind", ReadSource(WIND_SOURCE_PATH), 4, 1);
    problem::Problem prob(N);
    for (int k = 0; k < N; ++k) prob.SetDynamics(problem::DiscretizedModel<examples::UserModel>(wind), k);
    CartpoleProblem(prob, N, 1.0, h);
    augmented_lagrangian::AugmentedLagrangianiLQR<4, 1> solver(prob);  // uploads the problem: no step known yet
    auto Z = std::make_shared<Trajectory<4, 1>>(N);
    Z->SetUniformStep(hf);
    solver.SetTrajectory(Z);
    solver.Solve();
    EXPECT(solver.GetStatus() == SolverStatus::kSolved && solver.MaxViolation() < 1e-4);
    EXPECT(std::abs(Z->State(N)[0] - 1.0) < 1e-3);
  }
  CASE("Problem::SetDynamics(model, k) with a different model per knot (problem.hpp:155-166): RK4 / explicit Euler / the "
       "caller's own discrete map, from one user source");
  {
    examples::UserModel steps("cartpole_steps", ReadSource(STEPS_SOURCE_PATH), 4, 1);
    problem::Problem prob(N);
    for (int k = 0; k < N; ++k)
      prob.SetDynamics(problem::DiscretizedModel<examples::UserModel>(steps.Model(k < 20 ? 0 : (k < 40 ? 1 : 2))), k);
    CartpoleProblem(prob, N, 0.9, h);
    EXPECT(prob.IsFullyDefined());
    augmented_lagrangian::AugmentedLagrangianiLQR<4, 1> solver(prob);
    auto Z = std::make_shared<Trajectory<4, 1>>(N);
    Z->SetUniformStep(hf);
    solver.SetTrajectory(Z);
    solver.Solve();
    EXPECT(solver.GetStatus() == SolverStatus::kSolved && solver.MaxViolation() < 1e-4);
    // knot 25 steps with explicit Euler: x+ = x + f(x, u) h (integration.hpp:90-94)
    double xd[4], worst = 0.0;
    cartpole_f(Z->State(25), Z->Control(25)[0], xd);
    for (int i = 0; i < 4; ++i) worst = std::max(worst, std::abs(Z->State(26)[i] - (Z->State(25)[i] + xd[i] * h)));
    EXPECT(worst < 1e-12);
    // knot 45 with the symplectic map and its gust at t_45 = float(45) * h
    const double t = static_cast<float>(45) * hf;
    cartpole_f(Z->State(45), Z->Control(45)[0] + 0.3 * std::sin(1.7 * t), xd);
    const double v0 = Z->State(45)[2] + xd[2] * h, v1 = Z->State(45)[3] + xd[3] * h;
    worst = std::max({std::abs(Z->State(46)[2] - v0), std::abs(Z->State(46)[3] - v1),
                      std::abs(Z->State(46)[0] - (Z->State(45)[0] + v0 * h)), std::abs(Z->State(46)[1] - (Z->State(45)[1] + v1 * h))});
    EXPECT(worst < 1e-12);
    // a model of another SOURCE on another knot is still refused
    bool threw = false;
    try {
      prob.SetDynamics(problem::DiscretizedModel<examples::Unicycle>(examples::Unicycle()), 3);
    } catch (const std::runtime_error&) {
      threw = true;
    }
    EXPECT(threw);
  }
  CASE("DiscretizedModel<Model, ExplicitEuler> (integration.hpp:87-104; test/problem/triple_integrator_test.cpp:135-156)");
  {
    const std::string src = ReadSource(CARTPOLE_SOURCE_PATH);
    examples::UserModel euler = examples::UserModel::Euler("cartpole_euler", src, 4, 1);
    bool threw = false;
    try {
      problem::DiscretizedModel<examples::UserModel> wrong(euler);  // compiled for Euler: not an RK4 model
    } catch (const std::runtime_error&) {
      threw = true;
    }
    EXPECT(threw);
    threw = false;
    try {
      problem::DiscretizedModel<examples::Unicycle, problem::ExplicitEuler> wrong{examples::Unicycle()};
    } catch (const std::runtime_error&) {
      threw = true;
    }
    EXPECT(threw);
    problem::Problem prob(N);
    std::vector<problem::DiscretizedModel<examples::UserModel, problem::ExplicitEuler>> models(
        N, problem::DiscretizedModel<examples::UserModel, problem::ExplicitEuler>(euler));
    prob.SetDynamics(models);  // the vector overload, problem.hpp:187-191
    CartpoleProblem(prob, N, 0.8, h);
    ilqr::iLQR<4, 1> solver(prob);
    auto Z = std::make_shared<Trajectory<4, 1>>(N);
    Z->SetUniformStep(hf);
    for (int k = 0; k < N; ++k) Z->Control(k)[0] = 0.4;
    solver.SetTrajectory(Z);
    solver.Rollout();
    double xd[4], worst = 0.0;
    for (int k : {0, 17, 59}) {
      cartpole_f(Z->State(k), 0.4, xd);
      for (int i = 0; i < 4; ++i) worst = std::max(worst, std::abs(Z->State(k + 1)[i] - (Z->State(k)[i] + xd[i] * h)));
    }
    EXPECT(worst < 1e-13);
    // [A | B] = Identity(n, n + m) + jac h: column 2 of A is e_0 h + e_2, the B column is (0, 0, 1 / D, -c / (l D)) h
    solver.UpdateExpansions();
    const std::vector<double> AB = solver.GetKnotPointFunction(0).GetDynamicsExpansion();
    EXPECT(AB.size() == 20 && std::abs(AB[0 + 2 * 4] - h) < 1e-15 && std::abs(AB[2 + 2 * 4] - 1.0) < 1e-15);
    EXPECT(std::abs(AB[2 + 4 * 4] - h / 1.0) < 1e-15 && std::abs(AB[3 + 4 * 4] + h / 0.5) < 1e-15);  // x0 = 0: s = 0, c = 1, D = mc
  }
}

// ---- what the facade records by default (cost-to-go, history) and what that costs ------------------------------------
static void RecordingPolicyTests() {
  CASE("Solve() of a small batch takes the persistent kernel; the step-level BackwardPass() records the cost-to-go");
  problems::UnicycleProblem def;
  auto ps = def.MakeALSolver();
  augmented_lagrangian::AugmentedLagrangianiLQR<3, 2>& solver = ps;
  solver.GetOptions().profiler_enable = true;
  solver.Solve();
  EXPECT(solver.GetStatus() == SolverStatus::kSolved && solver.GetStats().iterations_total == 11);
  EXPECT(solver.GetTiming().fused_sweeps > 0);  // (round 2: the default recording kept the facade off this path)
  EXPECT(solver.GetStats().alpha.size() == 12 && solver.GetStats().alpha[0] == 0.0625);  // history: one call, all fields
  // P, p stay readable behind Solve(), as in the reference (knot_point_function_type.hpp:243-268): the persistent kernel kept
  // them in registers, so the read makes the library run the last iteration's backward pass once more with the records on
  const std::vector<double> p0 = solver.GetiLQRSolver().GetKnotPointFunction(0).GetCostToGoGradient();
  const std::vector<double> P50 = solver.GetiLQRSolver().GetKnotPointFunction(50).GetCostToGoHessian();
  EXPECT(p0.size() == 3 && P50.size() == 9);
  EXPECT(solver.GetStats().iterations_total == 11 && solver.GetStatus() == SolverStatus::kSolved);  // (the replay left the solver state alone)
  // ... the same values a solve that records them all the way gives (then on the batched kernels), and the next default
  // solve is back on the persistent kernel
  solver.GetiLQRSolver().SetRecordCostToGo(true);
  solver.SetTrajectory(std::make_shared<Trajectory<3, 2>>(def.InitialTrajectory()));
  solver.Solve();
  EXPECT(solver.GetStats().iterations_total == 11 && solver.GetTiming().fused_sweeps == 0);
  const std::vector<double> p0r = solver.GetiLQRSolver().GetKnotPointFunction(0).GetCostToGoGradient();
  const std::vector<double> P50r = solver.GetiLQRSolver().GetKnotPointFunction(50).GetCostToGoHessian();
  EXPECT(p0 == p0r && P50 == P50r);
  solver.GetiLQRSolver().SetRecordCostToGo(false);  // explicit: never
  solver.SetTrajectory(std::make_shared<Trajectory<3, 2>>(def.InitialTrajectory()));
  solver.Solve();
  EXPECT(solver.GetTiming().fused_sweeps > 0);
  solver.GetiLQRSolver().SetRecordCostToGo(true);
  solver.SetTrajectory(std::make_shared<Trajectory<3, 2>>(def.InitialTrajectory()));
  solver.Solve();
  EXPECT(solver.GetStats().iterations_total == 11 && solver.GetTiming().fused_sweeps == 0);
  EXPECT(solver.GetiLQRSolver().GetKnotPointFunction(0).GetCostToGoGradient().size() == 3);
  CASE("The history follows max_iterations_total (no silent truncation at 300 rows)");
  solver.GetOptions().max_iterations_total = 450;
  solver.SetTrajectory(std::make_shared<Trajectory<3, 2>>(def.InitialTrajectory()));
  solver.Solve();
  EXPECT(solver.GetiLQRSolver().HistoryRowsNeeded() == 452 && solver.GetStats().iterations_total == 11);
  EXPECT(solver.GetStats().cost.size() == 12);
}

int main() {
  try {
    RecordingPolicyTests();
    KnotTimeTests();
    UnicycleiLQRTest();
    AugLagTest();
    ExampleTests();
    UserFunctorTests();
    UserTypeListTests();
    DynamicsKindTests();
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 100;
  }
  std::printf("%d checks, %d failed\n", g_checks, g_failed);
  return g_failed;
}
