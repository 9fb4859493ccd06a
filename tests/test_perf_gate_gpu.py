"""A perf regression gate on the GPU box.

Round 5 lost GPU-minutes three times to a 10 % per-iteration regression of the persistent kernel that was mistaken for box
noise (profiles/r05_experiments.txt #3): the parity tests assert bits, nothing asserted speed.  This test runs the headline
bench command (config 2, ten steps) and holds the figures that move when a kernel regresses -- ms per step, the persistent
launch, its iteration, the three sweep kernels' shares -- to BANDS committed in profiles/perf_bands.json.  The bands file is
data: it is updated deliberately, in the same commit as the change that moves a figure, from a run of this test (which
prints what it measured).  Box-to-box spread of one build is +-1.5 ... 3 %; the bands are +-8 %, the
launch-count figures exact."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dig(d, path):
    for key in path.split("."):
        d = d[key]
    return d


def test_headline_figures_stay_inside_their_bands():
    bands = json.load(open(os.path.join(ROOT, "profiles", "perf_bands.json")))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + bands["command"]
    best = None
    for attempt in range(2):  # (a first run on a cold box pages the library in: the better of two)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        if best is None or line["ms_per_step"] < best["ms_per_step"]:
            best = line
    failures = []
    print("\n[perf gate] measured / band centre (tolerance):")
    for path, band in bands["bands"].items():
        got = float(_dig(best, path))
        lo, hi = band["value"] * (1 - band["tol"]), band["value"] * (1 + band["tol"])
        # a figure may always be BETTER than its band (time-like: lower) -- that asks for an update of the file, not a failure
        worse = got > hi if band.get("lower_is_better", True) else got < lo
        better = got < lo if band.get("lower_is_better", True) else got > hi
        print(f"  {path}: {got:.4g} / {band['value']:.4g} (+-{100 * band['tol']:.0f} %)" + ("  BETTER than the band: update profiles/perf_bands.json" if better else ""))
        if worse:
            failures.append(f"{path} = {got:.4g}, band {lo:.4g} .. {hi:.4g}")
    for path, want in bands.get("exact", {}).items():
        got = _dig(best, path)
        print(f"  {path}: {got} (exact: {want})")
        if got != want:
            failures.append(f"{path} = {got}, expected exactly {want}")
    assert not failures, "perf regression against profiles/perf_bands.json: " + "; ".join(failures)
