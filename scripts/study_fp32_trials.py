"""Study for VERDICT r2 item 8 (CPU only, on the oracle): what would fp32 line-search TRIALS do to the ALTRO_F32
engines?  Oracle dtype 2 = what the ALTRO_F32 engines compute today (fp64 arithmetic, fp32 expansion / gain records);
dtype 3 = the same with every trial (rollout + cost) evaluated on an all-fp32 shadow and only the accepted trial rolled
out and costed again in fp64; dtype 4 = as 3, and the fp64 re-evaluation has to pass the acceptance test again.

    python scripts/study_fp32_trials.py [batch]        (default 1024 instances of each workload)
"""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d, _lib=lib, _prefix="oracle_")
threads = os.cpu_count() or 1
names = {2: "fp64 trials (today)", 3: "fp32 trials, fp64 re-evaluation", 4: "fp32 trials, fp64 re-check"}
for wl, fn in (("kThreeObstacles shard (config 3)", P.batch_three_obstacles), ("kTurn90 shard (config 2)", P.batch_turn90),
               ("12-state model (config 4)", P.batch_quadrotor12)):
    nb = B if "12-state" not in wl else min(B, 256)
    print(f"== {wl}, {nb} instances")
    base = None
    for dt in (2, 3, 4):
        s = fn(make, batch=nb, dtype=dt)
        if hasattr(lib, "oracle_set_threads"):
            lib.oracle_set_threads(s._h, C.c_int(threads))
        t0 = time.perf_counter()
        s.solve()
        sec = time.perf_counter() - t0
        st = s.get_stats()
        X, U = s.get_trajectory()
        cnt = (C.c_longlong * 3)()
        lib.oracle_study_counters(s._h, cnt)
        solved = st["status"] == 0
        line = (f"  dtype {dt} {names[dt]:34s} solved {solved.mean():.4f}  iterations mean {st['iterations_total'].mean():7.2f} "
                f"max {st['iterations_total'].max():3d}  cost mean(solved) {st['cost'][solved].mean():.6f}  viol max(solved) {st['violation'][solved].max():.2e}  ({sec:.1f} s)")
        if dt == 2:
            base = (st.copy(), X.copy())
        else:
            b_st, b_X = base
            both = solved & (b_st["status"] == 0)
            same_it = (st["iterations_total"] == b_st["iterations_total"]).mean()
            dit = np.abs(st["iterations_total"].astype(int) - b_st["iterations_total"].astype(int))
            line += (f"\n           vs today: same status {np.mean(st['status'] == b_st['status']):.4f}  same iteration count {same_it:.4f}  "
                     f"|d iterations| <= 2: {(dit <= 2).mean():.4f}  max |dX| over both-solved {np.abs(X[both] - b_X[both]).max():.2e}  "
                     f"rel cost diff max {np.max(np.abs(st['cost'][both] - b_st['cost'][both]) / np.abs(b_st['cost'][both])):.2e}"
                     f"\n           fp32 trials {cnt[0]}, fp64 re-evaluations {cnt[1]}, re-check rejections {cnt[2]}")
        print(line, flush=True)
