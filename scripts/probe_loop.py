#!/usr/bin/env python3
"""Device-side sweep loop (k_sweep_loop) against the host-paced sweeps: same bits, and the timings of both.

usage: probe_loop.py [turn90|obstacles|obstacles32] [batch] [reps]     (runs itself twice: ALTRO_HIP_SWEEP_LOOP unset / = 0)"""
import importlib
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(kind, batch, reps, out):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    A = g.load_package()
    P = importlib.import_module("altro_cpp_amd.problems")
    make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
    if kind == "turn90":
        s = P.batch_turn90(make, batch=batch, seed=P.SEED_BASE + 3)
    elif kind == "obstacles":
        s = P.batch_three_obstacles(make, batch=batch, dtype=A.F64)
    else:
        s = P.batch_three_obstacles(make, batch=batch, dtype=A.F32)
    o = s.get_options()
    o.profiler_enable = 1
    s.set_options(o)
    ms = []
    for rep in range(reps):
        s.reset_trajectory()
        t0 = time.perf_counter()
        s.solve()
        ms.append(1e3 * (time.perf_counter() - t0))
    tm = s.get_timing()
    st = s.get_stats()
    X, U = s.get_trajectory()
    K, d = s.get_gains()
    res = {"X": X, "U": U, "K": K, "d": d, "lam": s.get_duals(), "pen": s.get_penalties(), "c": s.get_constraint_values(),
           "costs": s.get_knot_costs()}
    for f in st.dtype.names:
        res["st_" + f] = st[f]
    np.savez(out, **res)
    keys = ("total_ms", "init_ms", "expansions_ms", "backward_pass_ms", "forward_pass_ms", "fused_ms", "loop_ms", "sweeps", "fused_sweeps",
            "launches", "sweep_launches", "loop_workgroups", "loop_iterations", "loop_handover", "loop_instance_iterations",
            "fused_instance_iterations", "instance_iterations", "twin_handovers", "fused_workgroup_iterations", "host_naps")
    print("  ms per solve", " ".join("%.3f" % m for m in ms), "| solved %.4f" % float((st["status"] == 0).mean()),
          "| it max %d mean %.2f" % (st["iterations_total"].max(), st["iterations_total"].mean()))
    print("  " + " ".join("%s=%s" % (k, ("%.3f" % tm[k]) if isinstance(tm[k], float) else tm[k]) for k in keys), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
        return
    kind = sys.argv[1] if len(sys.argv) > 1 else "turn90"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    res = []
    with tempfile.TemporaryDirectory() as tmp:
        for tag, env in (("loop", {}), ("sweeps", {"ALTRO_HIP_SWEEP_LOOP": "0"})):
            out = os.path.join(tmp, tag + ".npz")
            print(kind, batch, tag, flush=True)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", kind, str(batch), str(reps), out],
                               env=dict(os.environ, **env), timeout=900)
            if r.returncode != 0:
                print("  FAILED rc", r.returncode, flush=True)
                return 1
            res.append(dict(np.load(out)))
    a, b = res
    bad = [k for k in a if not np.array_equal(a[k], b[k], equal_nan=True)]
    print(kind, batch, "BIT-IDENTICAL" if not bad else "DIFFERENT: " + ", ".join(bad), flush=True)
    for k in bad[:6]:
        d = np.flatnonzero((a[k] != b[k]).reshape(a[k].shape[0], -1).any(axis=1))
        print("   ", k, "instances", d[:10], "of", len(d))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
