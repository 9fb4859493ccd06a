"""Round 5, segments of rejection streaks: ms per solve (min / median of REPS solves in one process) of config 3 (or 2) for a
list of environment settings given as arguments, e.g.
    python scripts/probe_seg_policy.py c3 "ALTRO_HIP_SEGMENTS=0" "ALTRO_HIP_SEG_ABOVE=0 ALTRO_HIP_PERSIST_AT=256"
"""
import os
import subprocess
import sys

CHILD = r'''
import importlib, os, sys, time, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
name = sys.argv[1]
s = P.batch_three_obstacles(make, batch=4096, dtype=A.F32) if name == "c3" else P.batch_turn90(make, batch=4096, seed=P.SEED_BASE + 3)
ms, rows = [], []
for rep in range(int(os.environ.get("REPS", "9"))):
    s.reset_trajectory(); t0 = time.perf_counter(); s.solve(); dt = time.perf_counter() - t0; tm = s.get_timing()
    if rep: ms.append(1e3 * dt)
    rows.append((tm["sweep_launches"], tm["fused_sweeps"], tm["fused_workgroup_iterations"], tm["twin_handovers"]))
st = s.get_stats()
print("%s  min %.2f  median %.2f  max %.2f ms | (sweep launches, longest chain in the persistent launch, longest workgroup, hand-overs) %s | solved %d its %d" %
      (name, min(ms), statistics.median(ms), max(ms), rows[-1], int((st["status"] == 0).sum()), int(st["iterations_total"].sum())), flush=True)
'''
name = sys.argv[1]
for spec in sys.argv[2:]:
    env = dict(kv.split("=", 1) for kv in spec.split())
    r = subprocess.run([sys.executable, "-c", CHILD, name], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    print(f"{spec:70s} {r.stdout.strip() or r.stderr[-600:]}", flush=True)
