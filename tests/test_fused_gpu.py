"""The persistent fused tail kernel (k_sweep_fused) is the same device code as the three sweep kernels:
solving with and without it must give the same bits."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_SCRIPT = r'''
import importlib, sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
out = {}
for name, s in (("turn90", P.batch_turn90(make, batch=40)), ("obstacles", P.batch_three_obstacles(make, batch=24, dtype=A.F64))):
    s.solve()
    X, U = s.get_trajectory()
    st = s.get_stats()
    tm = s.get_timing()
    out[name + "_X"] = X; out[name + "_U"] = U; out[name + "_K"] = s.get_gains()[0]
    out[name + "_it"] = st["iterations_total"]; out[name + "_status"] = st["status"]; out[name + "_cost"] = st["cost"]
    out[name + "_fused"] = np.array([tm["fused_sweeps"]])
np.savez(sys.argv[1], **out)
'''


def _run(tmp_path, tag, env_extra):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / f"{tag}.npz")
    env = dict(os.environ, **env_extra)
    subprocess.run([sys.executable, "-c", _SCRIPT % root, out], check=True, env=env, timeout=600)
    return np.load(out)


def test_fused_tail_matches_separate_kernels_bitwise(tmp_path):
    a = _run(tmp_path, "fused", {})
    b = _run(tmp_path, "plain", {"ALTRO_HIP_NO_FUSED_SWEEP": "1"})
    assert a["turn90_fused"][0] > 0 and a["obstacles_fused"][0] > 0   # the persistent kernel really ran
    assert b["turn90_fused"][0] == 0 and b["obstacles_fused"][0] == 0
    for k in a.files:
        if k.endswith("_fused"):
            continue
        assert np.array_equal(a[k], b[k]), k
