"""Is the result independent of where the batched sweeps hand over to the persistent kernel?  Full config-3 and config-2
batches, bitwise, against the run without the persistent kernel: python scripts/probe_persist_bitwise.py"""
import importlib, os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, root)
    import __graft_entry__ as g
    A = g.load_package()
    P = importlib.import_module("altro_cpp_amd.problems")
    hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
    which, dt, B = sys.argv[3], sys.argv[4], int(sys.argv[5])
    fac = P.batch_three_obstacles if which == "obstacles" else P.batch_turn90
    s = fac(hm, batch=B, dtype=A.F32 if dt == "f32" else A.F64)
    for rep in range(2):
        s.reset_trajectory()
        s.solve()
        X, U = s.get_trajectory(); st = s.get_stats()
        np.savez(sys.argv[2] + f"_{rep}.npz", X=X, U=U, it=st["iterations_total"], status=st["status"], cost=st["cost"])
    sys.exit(0)
for which, dt, B in (("obstacles", "f32", 4096), ("turn90", "f64", 4096)):
    ref = None
    for tag, env in (("nofused", {"ALTRO_HIP_NO_FUSED_SWEEP": "1"}), ("at256", {}), ("at200", {"ALTRO_HIP_PERSIST_AT": "200"}),
                     ("at300", {"ALTRO_HIP_PERSIST_AT": "300"}), ("at1024", {"ALTRO_HIP_PERSIST_AT": "1024"}),
                     ("at256b", {}), ("at256_nospec", {"ALTRO_HIP_NO_SPECULATION": "1"})):
        f = f"/tmp/pb_{tag}"
        subprocess.run([sys.executable, __file__, "child", f, which, dt, str(B)], check=True, env=dict(os.environ, **env))
        for rep in range(2):
            o = np.load(f + f"_{rep}.npz")
            if ref is None:
                ref = o
            bad = np.nonzero((o["it"] != ref["it"]) | (o["status"] != ref["status"]) | (o["X"] != ref["X"]).any(axis=(1, 2)))[0]
            print(which, dt, tag, rep, "max it", int(o["it"].max()), "solved", float((o["status"] == 0).mean()), "differing instances", len(bad),
                  [(int(b), int(ref["it"][b]), int(o["it"][b]), int(ref["status"][b]), int(o["status"][b])) for b in bad[:8]], flush=True)
