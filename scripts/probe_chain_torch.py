"""Chains of sweeps beside a framework's stream pool: torch side streams created before / after the solver.
python scripts/probe_chain_torch.py <before|after>"""
import importlib, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
mode = sys.argv[1]
torch.cuda.set_device(0)
def pool():
    ss = [torch.cuda.Stream() for _ in range(3)]
    for st in ss:
        with torch.cuda.stream(st):
            x = torch.ones(1 << 20, device="cuda") + 1
    torch.cuda.synchronize()
    return ss
keep = pool() if mode == "before" else None
s = P.batch_turn90(hm, batch=4096, dtype=A.F64)
s.num_constraints()
if mode == "after":
    keep = pool()
s.set_options(profiler_enable=1)
for rep in range(3):
    s.reset_trajectory()
    t0 = time.perf_counter(); s.solve(); dt = time.perf_counter() - t0
    with torch.cuda.stream(keep[0]):
        y = torch.ones(1 << 20, device="cuda") * 2  # (a collective between the solves)
    torch.cuda.synchronize()
    tm = s.get_timing()
    print(mode, os.environ.get("GPU_MAX_HW_QUEUES"), "rep", rep, "ms %.3f" % (1e3 * dt), "sweep launches", tm["sweep_launches"], flush=True)
