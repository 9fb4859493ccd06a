"""Round 5: segments of rejection streaks in the batched sweeps -- a solve with them against one without (ALTRO_HIP_SEGMENTS=0),
array by array, and the wall time of both."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, sys, numpy as np, os, time
sys.path.insert(0, %r)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
name = sys.argv[2]
fac = {"obstacles_4096_r32": lambda: P.batch_three_obstacles(make, batch=4096, dtype=A.F32),
       "obstacles_2048_f64": lambda: P.batch_three_obstacles(make, batch=2048, dtype=A.F64),
       "turn90_4096": lambda: P.batch_turn90(make, batch=4096, seed=P.SEED_BASE + 3)}[name]
s = fac()
out = {}
for rep in range(3):
    s.reset_trajectory()
    t0 = time.perf_counter(); s.solve(); dt = time.perf_counter() - t0
    tm = s.get_timing()
    print(name, "rep", rep, "ms", round(1e3 * dt, 2), "sweeps", tm["sweeps"], "sweep launches", tm["sweep_launches"], "fused sweeps", tm["fused_sweeps"],
          "twins", tm["twin_claims"], tm["twin_handovers"], "instance-iterations", tm["instance_iterations"], flush=True)
X, U = s.get_trajectory(); st = s.get_stats(); K, d = s.get_gains()
out.update(X=X, U=U, K=K, d=d, lam=s.get_duals(), pen=s.get_penalties(), c=s.get_constraint_values(), costs=s.get_knot_costs())
for f in st.dtype.names: out["st_" + f] = st[f]
for k in (0, 50, 100):
    for key, v in s.get_expansion(k).items():
        if k < 100 or key in ("lxx", "lx"): out["exp%%d_%%s" %% (k, key)] = v
np.savez(sys.argv[1], **out)
'''
for name in sys.argv[1:] or ["obstacles_2048_f64", "obstacles_4096_r32", "turn90_4096"]:
    res = {}
    for tag, env in (("seg", {}), ("noseg", {"ALTRO_HIP_SEGMENTS": "0"})):
        out = f"/tmp/probe_seg_{name}_{tag}.npz"
        r = subprocess.run([sys.executable, "-c", CHILD % ROOT, out, name], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        print(f"== {name} {tag}\n" + r.stdout + r.stderr[-600:])
        res[tag] = np.load(out) if os.path.exists(out) else None
    a, b = res["seg"], res["noseg"]
    if a is None or b is None:
        continue
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    print("differing arrays:", bad)
    for k in bad[:10]:
        x, y = np.asarray(a[k], float), np.asarray(b[k], float)
        diff = np.abs(x - y)
        inst = np.unique(np.argwhere(diff.reshape(diff.shape[0], -1).max(axis=1) > 0)[:, 0])
        print(f"  {k}: max diff {np.nanmax(diff):.3e}, instances {list(inst[:10])} ({len(inst)}) status {a['st_status'][inst[:8]]} its seg {a['st_iterations_total'][inst[:8]]} noseg {b['st_iterations_total'][inst[:8]]}")
