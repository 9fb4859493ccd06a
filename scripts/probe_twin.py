"""Round 5: what the twin workgroups of the persistent kernel did in one solve (ALTRO_HIP_TWIN_DEBUG=1 prints the mailboxes),
and whether a solve with twins equals one without, array by array."""
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import importlib, sys, numpy as np, os
sys.path.insert(0, %r)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
name = sys.argv[2]
fac = {"turn90_512": lambda: P.batch_turn90(make, batch=512, seed=P.SEED_BASE + 3),
       "turn90_4096": lambda: P.batch_turn90(make, batch=4096, seed=P.SEED_BASE + 3),
       "obstacles_192": lambda: P.batch_three_obstacles(make, batch=192, dtype=A.F64)}[name]
s = fac()
out = {}
for rep in range(2):
    s.reset_trajectory(); s.solve()
    tm = s.get_timing()
    print(name, "rep", rep, "ms", round(tm["total_ms"], 3), "twins", tm["twin_workgroups"], "claims", tm["twin_claims"], "handovers", tm["twin_handovers"],
          "sweeps", tm["sweeps"], "fused its", tm["fused_instance_iterations"], flush=True)
X, U = s.get_trajectory(); st = s.get_stats(); K, d = s.get_gains()
out.update(X=X, U=U, K=K, d=d, lam=s.get_duals(), pen=s.get_penalties(), c=s.get_constraint_values(), costs=s.get_knot_costs())
for f in st.dtype.names: out["st_" + f] = st[f]
for k in (0, 50, 100):
    for key, v in s.get_expansion(k).items():
        if k < 100 or key in ("lxx", "lx"): out["exp%%d_%%s" %% (k, key)] = v
np.savez(sys.argv[1], **out)
'''

for name in sys.argv[1:] or ["turn90_512", "turn90_4096", "obstacles_192"]:
    res = {}
    for tag, env in (("twin", {"ALTRO_HIP_TWIN_DEBUG": "1"}), ("solo", {"ALTRO_HIP_TWIN": "0"})):
        out = f"/tmp/probe_twin_{name}_{tag}.npz"
        r = subprocess.run([sys.executable, "-c", CHILD % ROOT, out, name], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        print(f"== {name} {tag}\n" + r.stdout + "\n".join(l for l in r.stderr.splitlines() if "twin" in l or "slot" in l or "Error" in l or "error" in l))
        res[tag] = np.load(out)
    a, b = res["twin"], res["solo"]
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    print("differing arrays:", bad)
    for k in bad[:12]:
        x, y = np.asarray(a[k], float), np.asarray(b[k], float)
        diff = np.abs(x - y)
        inst = np.unique(np.argwhere(diff.reshape(diff.shape[0], -1).max(axis=1) > 0)[:, 0]) if diff.ndim >= 1 and diff.shape[0] == a["st_status"].shape[0] else []
        print(f"  {k}: max diff {np.nanmax(diff):.3e}, instances {list(inst[:10])} ({len(inst)}), status there {a['st_status'][inst[:10]] if len(inst) else ''} "
              f"iters twin {a['st_iterations_total'][inst[:6]] if len(inst) else ''} solo {b['st_iterations_total'][inst[:6]] if len(inst) else ''}")
