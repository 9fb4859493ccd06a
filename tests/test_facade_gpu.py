"""The C++ facade + perf driver (counterpart of the reference's perf/benchmark_unicycle.cpp) on a GPU."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_benchmark_unicycle_driver():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "perf"), "benchmark_unicycle"])
    exe = os.path.join(ROOT, "perf", "benchmark_unicycle")
    r = subprocess.run([exe, "2", "256"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    # kThreeObstacles through the facade: 50 iLQR iterations in 5 AL iterations, kSolved
    m = re.search(r"Iteration 1: Cost = ([0-9.e+-]+), iters = (\d+), outer = (\d+), status = (\d+)", r.stdout)
    assert m, r.stdout
    assert (int(m.group(2)), int(m.group(3)), int(m.group(4))) == (50, 5, 0)
    # iLQR::Cost() after the solve re-evaluates the AL cost with the updated duals (oracle: 9.437361800688565)
    assert abs(float(m.group(1)) - 9.437361800688565) < 1e-8
    assert re.search(r"batch 256 run 1: .* solved (\d+)/256", r.stdout), r.stdout
    # profiler tree in the layout of the reference's perf/profiler_unicycle.out
    assert "Description                  Time (us)   %Total  %Parent" in r.stdout
    tree = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s*(\w+)\s+(\d+)\s+\d+\s+\d+\s*$", r.stdout, re.M)}
    assert {"al", "ilqr", "backward_pass", "expansions", "forward_pass", "sweep_fused", "init"} <= set(tree)
    assert tree["al"] > 0 and tree["ilqr"] > 0 and tree["ilqr"] <= tree["al"] * 1.05


def _build(dirname, exe):
    path = os.path.join(ROOT, dirname, exe)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, dirname), exe])
    return path


@pytest.mark.gpu
def test_reference_gtests_through_the_facade():
    """tests/cpp/facade_reference_tests.cpp: the reference's own gtest cases for the hot path
    (unicycle_ilqr_test.cpp:27-144, auglag_test.cpp:249-399, example_*_test.cpp) replayed through the facade."""
    exe = _build(os.path.join("tests", "cpp"), "facade_reference_tests")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert re.search(r"(\d+) checks, 0 failed", r.stdout)
    assert "Goal Constraint at index 100" in r.stdout  # PrintViolations(sort = true), al_solver.hpp:68-81


@pytest.mark.gpu
def test_benchmark_triple_integrator_driver():
    exe = _build("perf", "benchmark_triple_integrator")
    r = subprocess.run([exe, "2", "1024"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    # example_triple_integrator_test.cpp:16-70: 2 iterations unconstrained; constrained solve reaches the tolerance
    assert re.search(r"Unconstrained triple integrator: iters = 2, outer = 1, status = 0", r.stdout), r.stdout
    m = re.search(r"Constrained triple integrator: iters = (\d+), outer = (\d+), status = 0, violation = ([0-9.e+-]+)", r.stdout)
    assert m and float(m.group(3)) < 1e-4, r.stdout
    assert re.search(r"batch 1024 run 1: .* solved 1024/1024, 2048 instance-iterations", r.stdout), r.stdout
    assert "Description                  Time (us)   %Total  %Parent" in r.stdout


@pytest.mark.gpu
def test_benchmark_expansions_driver():
    exe = _build("perf", "benchmark_expansions")
    r = subprocess.run([exe, "20"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rows = re.findall(r"batch\s+(\d+): UpdateExpansions\s+([0-9.]+) us per call", r.stdout)
    assert [int(b) for b, _ in rows] == [1, 64, 1024, 4096, 16384], r.stdout
    # batching is the point: 4096 instances cost far less than 4096 single-instance calls
    t = {int(b): float(us) for b, us in rows}
    assert t[4096] < 50 * t[1]


@pytest.mark.gpu
def test_pointer_api_driver_runs(tmp_path):
    """tests/cpp/pointer_api_driver.cpp -- the repository's own driver of the facade's pointer-taking API (shared_ptr setters and
    their vector overloads, the options the reference's drivers set, by-value trajectories, the profiler file) -- solves what the
    reference's tests say these problems solve: example_unicycle_test.cpp:65-80 (50 iterations, 5 outer, kSolved),
    example_triple_integrator_test.cpp:16-70 (2 iterations)."""
    exe = _build(os.path.join("tests", "cpp"), "pointer_api_driver")
    r = subprocess.run([exe, "2", "4"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert len(re.findall(r"three_obstacles\[\d\]: iterations 50 outer 5 status 0 ", r.stdout)) == 2, r.stdout
    assert "triple_integrator[0]: iterations 2 outer" in r.stdout, r.stdout
    assert re.search(r"triple_integrator_bounded\[0\]: iterations \d+ outer \d+ status 0 ", r.stdout), r.stdout
    assert re.search(r"assembled_through_pointers\[0\]: iterations \d+ outer \d+ status 0 cost \S+ constraints 502 threads 1", r.stdout), r.stdout
    assert "ALL SCENARIOS OK" in r.stdout
    # verbose = kDebug: the log of the solve (every column of solver_stats.cpp:80-116), printed after it
    assert re.search(r"iters\s+cost\s+viol\s+dJ\s+grad\s+alpha\s+reg\s+z\s+pen", r.stdout), r.stdout[:2000]
    # profiler_output_to_file: the tree of the solver's solves, written at its destruction (timer.cpp:10-14)
    text = (tmp_path / "profiler_three_obstacles.out").read_text()
    assert "Description                  Time (us)   %Total  %Parent" in text and "backward_pass" in text, text


@pytest.mark.gpu
def test_al_cost_views_program_runs(tmp_path):
    """tests/cpp/al_cost_views.cpp: the reference's callers that look INTO the AL solver -- Init() sets the initial penalty
    (example_unicycle_test.cpp:95-106), a ConstraintValues pointer taken from GetALCost(N) before the solve shows the goal
    duals after UpdateDuals() (auglag_test.cpp:250-275), ResetDualVariables() zeroes them."""
    exe = _build(os.path.join("tests", "cpp"), "al_cost_views")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0 and "al_cost_views: 0 failures" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
