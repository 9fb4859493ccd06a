"""Does a batch solve faster as S concurrent sub-batches (one handle and one stream each, solve_async)?
python scripts/probe_split_batch.py [config: turn90|obstacles]"""
import importlib, os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
which = sys.argv[1] if len(sys.argv) > 1 else "turn90"
B = 4096
for S in (1, 2, 3, 4, 8, 1, 2, 3, 4, 8):
    os.environ["ALTRO_HIP_PERSIST_AT"] = str(256 // S)
    edges = [B * i // S for i in range(S + 1)]
    if which == "turn90":
        solvers = [P.batch_turn90(hm, batch=B, dtype=A.F64, shard=(edges[i], edges[i + 1])) for i in range(S)]
    else:
        solvers = [P.batch_three_obstacles(hm, batch=B, dtype=A.F32, shard=(edges[i], edges[i + 1])) for i in range(S)]
    for s in solvers:
        s.set_options(profiler_enable=0)
    def step():
        for s in solvers:
            s.reset_trajectory()
            s.solve_async()
        for s in solvers:
            s.wait()
    step()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    dt = (time.perf_counter() - t0) / 5
    solved = sum(int((s.get_stats()["status"] == 0).sum()) for s in solvers)
    print(which, "sub-batches", S, "ms per step %.3f" % (1e3 * dt), "solved", solved, flush=True)
    for s in solvers:
        s.close()
