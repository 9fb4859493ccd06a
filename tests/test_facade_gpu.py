"""The C++ facade + perf driver (counterpart of the reference's perf/benchmark_unicycle.cpp) on a GPU."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_benchmark_unicycle_driver():
    exe = os.path.join(ROOT, "perf", "benchmark_unicycle")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "perf")])
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
