"""Process-to-process reproducibility: the full config-3 batch (bench.py's step: reset + solve, three times) in many fresh
processes, each compared bitwise with the first: python scripts/probe_repeat_processes.py [processes]"""
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
    s = P.batch_three_obstacles(hm, batch=4096, dtype=A.F32) if which == "obstacles" else P.batch_turn90(hm, batch=4096, dtype=A.F64)
    s.set_options(profiler_enable=0)
    outs = {}
    for rep in range(3):
        s.reset_trajectory()
        s.solve()
        X, U = s.get_trajectory(); st = s.get_stats()
        outs[f"X{rep}"] = X; outs[f"it{rep}"] = st["iterations_total"]; outs[f"status{rep}"] = st["status"]
    s.set_options(profiler_enable=1)
    s.reset_trajectory(); s.solve()
    tm = s.get_timing()
    outs["sweeps"] = np.array([tm["sweeps"], tm["fused_sweeps"]]); outs["fused_ms"] = np.array([tm["fused_ms"]])
    np.savez(sys.argv[2], **outs)
    sys.exit(0)
nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ref = {}
for i in range(nproc):
    which = "turn90" if i % 4 == 3 else "obstacles"
    f = f"/tmp/rp_{i}.npz"
    subprocess.run([sys.executable, __file__, "child", f, which], check=True)
    o = np.load(f)
    if which not in ref:
        ref[which] = o
    r = ref[which]
    msgs = []
    for rep in range(3):
        bad = np.nonzero((o[f"it{rep}"] != r["it0"]) | (o[f"status{rep}"] != r["status0"]) | (o[f"X{rep}"] != r["X0"]).any(axis=(1, 2)))[0]
        if len(bad):
            msgs.append((rep, len(bad), [(int(b), int(r["it0"][b]), int(o[f"it{rep}"][b]), int(r["status0"][b]), int(o[f"status{rep}"][b])) for b in bad[:10]]))
    print(i, which, "sweeps", o["sweeps"].tolist(), "fused_ms", float(o["fused_ms"][0]), "max it", int(o["it0"].max()), "ANOMALY" if msgs else "same", msgs, flush=True)
