"""Does a solve read device memory it has not written?  Fill a few GB of device memory with a pattern, release them to the
driver, build the solver (its hipMalloc calls then reuse those pages), solve, and compare bitwise with the run on untouched memory:
python scripts/probe_poison.py"""
import importlib, os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, root)
    import torch
    import __graft_entry__ as g
    A = g.load_package()
    P = importlib.import_module("altro_cpp_amd.problems")
    hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
    which, pattern = sys.argv[3], sys.argv[4]
    if pattern != "none":
        chunks = []
        for i in range(16):  # 16 x 512 MB
            x = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
            if pattern == "ff":
                x.fill_(255)
            elif pattern == "random":
                x.random_(0, 256)
            elif pattern == "ones":
                x.view(torch.int32).fill_(1)
            elif pattern == "big":
                x.view(torch.float64).fill_(1e30)
            chunks.append(x)
        torch.cuda.synchronize()
        del chunks, x
        torch.cuda.empty_cache()
    if which == "obstacles":
        s = P.batch_three_obstacles(hm, batch=4096, dtype=A.F32)
    elif which == "turn90":
        s = P.batch_turn90(hm, batch=4096, dtype=A.F64)
    elif which == "tripleint":
        s = P.batch_triple_integrator(hm, batch=1024, dtype=A.F64)
    else:
        s = P.batch_quadrotor12(hm, batch=256, dtype=A.F32)
    if which == "tripleint":
        s.solve_ilqr()
    else:
        s.solve()
    X, U = s.get_trajectory(); st = s.get_stats()
    np.savez(sys.argv[2], X=X, U=U, it=st["iterations_total"], status=st["status"], cost=st["cost"], lam=s.get_duals())
    sys.exit(0)
for which in ("obstacles", "turn90", "tripleint", "quad12"):
    ref = None
    for pattern in ("none", "ff", "random", "ones", "big"):
        f = f"/tmp/poison_{pattern}.npz"
        subprocess.run([sys.executable, __file__, "child", f, which, pattern], check=True)
        o = np.load(f)
        if ref is None:
            ref = o
        bad = np.nonzero((o["it"] != ref["it"]) | (o["status"] != ref["status"]) | (o["X"] != ref["X"]).any(axis=(1, 2)))[0]
        print(which, pattern, "max it", int(o["it"].max()), "solved", float((o["status"] == 0).mean()), "differing instances", len(bad),
              [(int(b), int(ref["it"][b]), int(o["it"][b]), int(ref["status"][b]), int(o["status"][b])) for b in bad[:8]],
              "lam diffs", int((o["lam"] != ref["lam"]).sum()), flush=True)
