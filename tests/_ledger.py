"""Ledger of MEASURED parity errors: every comparison of the HIP path with the oracle records what it
measured next to the bar it asserted, and the session writes the maxima to gpurun_out/r06_parity_errors.json (copied to
profiles/ after a GPU run).  A bar is then pinned to the measurement instead of to a guess: the GPU and the oracle are both
deterministic, so the measured maxima are reproducible to the bit; the bars keep a small factor over them."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENTRIES = {}


def _test_id():
    cur = os.environ.get("PYTEST_CURRENT_TEST", "?")
    return cur.split(" ")[0].replace("tests/", "")


def record(label, err_abs, err_rel, used, rtol=None, atol=None, note=None):
    """err_abs / err_rel: measured maxima; used: max(err / tol) -- the fraction of the bar that was used (<= 1 passes)."""
    key = _test_id() + " :: " + label
    e = ENTRIES.setdefault(key, dict(max_abs=0.0, max_rel=0.0, used=0.0, rtol=rtol, atol=atol, calls=0))
    e["max_abs"] = max(e["max_abs"], float(err_abs))
    e["max_rel"] = max(e["max_rel"], float(err_rel))
    e["used"] = max(e["used"], float(used))
    e["calls"] += 1
    if note:
        e["note"] = note


def count(label, n, allowed, ids=()):
    """An explicit counter of tolerated events (schedule flips of chaotic instances): measured count, allowance, ids."""
    key = _test_id() + " :: " + label
    ENTRIES[key] = dict(count=int(n), allowed=int(allowed), ids=[int(i) for i in list(ids)[:20]])
    print(f"[ledger] {label}: {int(n)} (allowed {int(allowed)}) ids {[int(i) for i in list(ids)[:20]]}")


def close(a, b, rtol, atol, label="values"):
    """|a - b| <= atol + rtol * max(|a|, |b|) elementwise; records the measured maxima whatever the verdict."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    err = np.abs(a - b)
    scale = np.maximum(np.abs(a), np.abs(b))
    tol = atol + rtol * scale
    if err.size:
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = np.where(scale > 0, err / np.maximum(scale, 1e-300), 0.0)
            used = np.where(tol > 0, err / np.where(tol > 0, tol, 1.0), np.where(err > 0, np.inf, 0.0))
        record(label, err.max(), rel.max(), used.max(), rtol, atol)
    assert (err <= tol).all(), f"{label}: max err {err.max():.3e} (rel {(err / (np.abs(b) + 1e-300)).max():.3e}), bar rtol {rtol:g} atol {atol:g}"


def close_normwise(a, b, rtol, label="normwise"):
    """max-norm error relative to the max-norm of the reference block (per instance)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    ax = tuple(range(1, a.ndim))
    err = np.abs(a - b).max(axis=ax) if a.size else np.zeros(0)
    ref = np.maximum(np.abs(b).max(axis=ax), 1e-12) if a.size else np.ones(0)
    if err.size:
        record(label, err.max(), (err / ref).max(), (err / (rtol * ref)).max() if rtol > 0 else np.inf, rtol, 0.0)
    assert (err <= rtol * ref).all(), f"{label}: max normwise rel err {(err / ref).max():.3e}, bar {rtol:g}"


def dump():
    if not ENTRIES:
        return None
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "r06_parity_errors.json")
    old = {}
    if os.path.exists(path):  # several pytest invocations of one GPU visit add up
        try:
            old = json.load(open(path))
        except Exception:
            old = {}
    old.update(ENTRIES)
    with open(path, "w") as f:
        json.dump(old, f, indent=1, sort_keys=True)
    return path
