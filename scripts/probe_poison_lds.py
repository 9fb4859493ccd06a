"""ALTRO_HIP_DEBUG_POISON: the full-size configurations solved with the LDS of every CU (and the candidate buffer) filled
with a pattern before every kernel, compared bitwise with the ordinary run: python scripts/probe_poison_lds.py"""
import importlib, os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, root)
    import __graft_entry__ as g
    A = g.load_package()
    P = importlib.import_module("altro_cpp_amd.problems")
    hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
    which = sys.argv[3]
    if which == "obstacles":
        s = P.batch_three_obstacles(hm, batch=4096, dtype=A.F32)
    elif which == "obstacles64":
        s = P.batch_three_obstacles(hm, batch=1024, dtype=A.F64)
    elif which == "turn90":
        s = P.batch_turn90(hm, batch=4096, dtype=A.F64)
    elif which == "tripleint":
        s = P.batch_triple_integrator(hm, batch=1024, dtype=A.F64)
    else:
        s = P.batch_quadrotor12(hm, batch=256, dtype=A.F32)
    outs = {}
    for rep in range(2):
        s.reset_trajectory()
        if which == "tripleint":
            s.reset_stats(); s.solve_ilqr()
        else:
            s.solve()
        X, U = s.get_trajectory(); st = s.get_stats()
        outs[f"X{rep}"] = X; outs[f"it{rep}"] = st["iterations_total"]; outs[f"status{rep}"] = st["status"]
    np.savez(sys.argv[2], **outs)
    sys.exit(0)
whichs = sys.argv[1:] or ["turn90", "obstacles", "obstacles64", "tripleint", "quad12"]
for which in whichs:
    ref = None
    for tag, env in (("none", {}), ("ff", {"ALTRO_HIP_DEBUG_POISON": "ffffffff"}), ("mix", {"ALTRO_HIP_DEBUG_POISON": "12345678,mix"}),
                     ("one", {"ALTRO_HIP_DEBUG_POISON": "3ff00000"}), ("big", {"ALTRO_HIP_DEBUG_POISON": "7e37e43c"})):
        f = f"/tmp/pl_{tag}.npz"
        subprocess.run([sys.executable, __file__, "child", f, which], check=True, env=dict(os.environ, **env))
        o = np.load(f)
        if ref is None:
            ref = o
        for rep in range(2):
            bad = np.nonzero((o[f"it{rep}"] != ref["it0"]) | (o[f"status{rep}"] != ref["status0"]) | (o[f"X{rep}"] != ref["X0"]).any(axis=(1, 2)))[0]
            print(which, tag, rep, "max it", int(o[f"it{rep}"].max()), "solved", float((o[f"status{rep}"] == 0).mean()), "differing instances", len(bad),
                  [(int(b), int(ref["it0"][b]), int(o[f"it{rep}"][b]), int(ref["status0"][b]), int(o[f"status{rep}"][b])) for b in bad[:6]], flush=True)
