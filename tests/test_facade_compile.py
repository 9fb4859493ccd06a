"""Host-only: the C++ facade accepts a caller written in the reference's own call style.

VERDICT r4 item 3 (row B1): reference call sites must compile against include/ unchanged -- shared_ptr overloads of
Problem::SetCostFunction / SetConstraint / SetDynamics (altro/problem/problem.hpp:113-202), the SolverOptions fields the
reference's drivers set (altro/common/solver_options.hpp:49-56, perf/benchmark_unicycle.cpp:34-35, perf/benchmarks.hpp:
15-22), TripleIntegratorProblem<dof> (examples/problems/triple_integrator.hpp:22), by-value InitialTrajectory() and
InitialTrajectory<n, m>() (examples/problems/unicycle.hpp:84-92), a copyable Trajectory, and the reference's include paths.
No GPU and no library needed: -fsyntax-only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def _syntax(path, *flags):
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + INC, "-fsyntax-only", *flags, path],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]


def test_pointer_api_driver_compiles():
    _syntax(os.path.join(ROOT, "tests", "cpp", "pointer_api_driver.cpp"))  # the repository's own driver of the pointer-taking API
    _syntax(os.path.join(ROOT, "tests", "cpp", "al_cost_views.cpp"))  # Init(), GetALCost(k)->...->GetDuals(): auglag_test.cpp:250-275


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "perf")), reason="the reference tree only exists in the build container")
@pytest.mark.parametrize("driver", ["benchmark_unicycle.cpp", "benchmark_triple_integrator.cpp"])
def test_reference_perf_drivers_compile_in_place(tmp_path, driver):
    """Row B1, on the reference's OWN call sites: /root/reference/perf/benchmark_unicycle.cpp and benchmark_triple_integrator.cpp
    are compiled where they lie, unedited, against this repository's include/ (the forwarding headers at the reference's include
    paths).  The only things added are what the image lacks and the path does not need: a stand-in for the three fmt headers the
    drivers include (fmt is not installed; the drivers only print with it) and a copy of the drivers' own perf/benchmarks.hpp
    made at test time (it sits beside them in the reference and includes nothing but the solver headers).  Nothing of the
    reference is committed here or travels to the GPU box, where this test is skipped."""
    stub = tmp_path / "stub"
    (stub / "fmt").mkdir(parents=True)
    fmt_stub = ("#pragma once\n#include <string>\nnamespace fmt {\n"
                "template <class... A> void print(const A&...) {}\n"
                "template <class... A> std::string format(const A&...) { return {}; }\n}\n")
    for name in ("format.h", "ostream.h", "chrono.h"):
        (stub / "fmt" / name).write_text(fmt_stub)
    (stub / "perf").mkdir()
    (stub / "perf" / "benchmarks.hpp").write_text(open(os.path.join(REFERENCE, "perf", "benchmarks.hpp")).read())
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I" + INC, "-I" + str(stub), os.path.join(REFERENCE, "perf", driver)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]


def test_perf_drivers_and_reference_gtests_compile():
    for rel in ("perf/benchmark_unicycle.cpp", "perf/benchmark_triple_integrator.cpp", "perf/benchmark_expansions.cpp"):
        _syntax(os.path.join(ROOT, rel))


@pytest.mark.parametrize("header", [
    "altro/augmented_lagrangian/al_solver.hpp", "altro/augmented_lagrangian/al_problem.hpp", "altro/common/solver_options.hpp",
    "altro/common/solver_stats.hpp", "altro/common/trajectory.hpp", "altro/ilqr/ilqr.hpp", "altro/problem/problem.hpp",
    "altro/problem/discretized_model.hpp", "altro/constraints/constraint.hpp", "examples/problems/unicycle.hpp",
    "examples/problems/triple_integrator.hpp", "examples/quadratic_cost.hpp", "examples/basic_constraints.hpp",
    "examples/obstacle_constraints.hpp", "examples/unicycle.hpp", "examples/triple_integrator.hpp"])
def test_reference_include_paths_exist(header):
    assert os.path.exists(os.path.join(INC, header)), header


def test_solver_options_carry_every_reference_field(tmp_path):
    """Every member of the reference's SolverOptions (solver_options.hpp:23-56) by name, with its default."""
    src = tmp_path / "opts.cpp"
    src.write_text('''
#include "altro/common/solver_options.hpp"
#include <type_traits>
int main() {
  altro::SolverOptions o;
  static_assert(std::is_same<decltype(o.verbose), altro::LogLevel>::value, "verbose");
  static_assert(std::is_same<decltype(o.log_directory), std::string>::value, "log_directory");
  static_assert(altro::kPickHardwareThreads == -1, "kPickHardwareThreads");
  bool ok = o.max_iterations_total == 300 && o.max_iterations_outer == 30 && o.max_iterations_inner == 100 &&
            o.cost_tolerance == 1e-4 && o.gradient_tolerance == 1e-2 && o.bp_reg_increase_factor == 1.6 && o.bp_reg_enable &&
            o.bp_reg_initial == 0.0 && o.bp_reg_max == 1e8 && o.bp_reg_min == 1e-8 && o.bp_reg_fail_threshold == 100 &&
            o.check_forwardpass_bounds && o.state_max == 1e8 && o.control_max == 1e8 && o.line_search_max_iterations == 20 &&
            o.line_search_lower_bound == 1e-8 && o.line_search_upper_bound == 10.0 && o.line_search_decrease_factor == 2 &&
            o.constraint_tolerance == 1e-4 && o.maximum_penalty == 1e8 && o.initial_penalty == 1.0 && o.reset_duals &&
            o.header_frequency == 10 && o.verbose == altro::LogLevel::kSilent && !o.profiler_enable &&
            !o.profiler_output_to_file && o.log_directory.empty() && o.profile_filename == "profiler.out" && o.nthreads == 1 &&
            o.tasks_per_thread == 1 && o.NumThreads() == 1;
  o.nthreads = altro::kPickHardwareThreads;
  ok = ok && o.NumThreads() >= 1;
  return ok ? 0 : 1;
}
''')
    exe = tmp_path / "opts"
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-I" + INC, "-o", str(exe), str(src), "-pthread"], capture_output=True, text=True)
    # only the inline altro_default_options symbol is missing without the library: the options struct itself is header-only
    if r.returncode != 0 and "altro_default_options" in r.stderr:
        lib = os.path.join(ROOT, "altro-cpp_amd", "csrc")
        r = subprocess.run(["g++", "-std=c++17", "-Wall", "-I" + INC, "-o", str(exe), str(src), "-pthread", "-L" + lib, "-laltro_hip",
                            "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert subprocess.run([str(exe)]).returncode == 0


def test_host_side_types_behave_like_the_reference(tmp_path):
    """tests/cpp/host_types.cpp -- the statements of the reference's problem_test.cpp:24-62, ilqr_class_test.cpp:36-70 and
    trajectory_test.cpp:40-100 about Problem::GetDynamics / GetCostFunction / GetInitialStatePointer and
    Trajectory::CheckTimeConsistency -- built and RUN here: these types live on the host (the library is linked, no device
    call is made)."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "host_types"])
    r = subprocess.run([os.path.join(ROOT, "tests", "cpp", "host_types")], capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
    assert r.returncode == 0 and "host_types: 0 failures" in r.stdout, r.stdout + r.stderr
