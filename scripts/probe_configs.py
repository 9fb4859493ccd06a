"""Wall time of one solve for every BASELINE config on one GPU (config 3's per-GPU shard: 4096 instances)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
mk = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
cases = [
    ("configs[0] single kThreeObstacles fp64 (AL)", lambda: P.unicycle_three_obstacles(mk), "al"),
    ("configs[1] 1024 triple integrator fp64 (iLQR)", lambda: P.batch_triple_integrator(mk, batch=1024), "ilqr"),
    ("configs[2] 4096 unicycle turn90 fp64 (AL)", lambda: P.batch_turn90(mk, batch=4096), "al"),
    ("configs[3] 4096/GPU unicycle obstacles fp32 (AL)", lambda: P.batch_three_obstacles(mk, batch=4096), "al"),
    ("configs[3'] 4096/GPU unicycle obstacles fp64 (AL)", lambda: P.batch_three_obstacles(mk, batch=4096, dtype=A.F64), "al"),
    ("configs[4] 1024 quadrotor12 fp32 (AL)", lambda: P.batch_quadrotor12(mk, batch=1024), "al"),
]
for name, make, mode in cases:
    s = make()
    run = s.solve if mode == "al" else s.solve_ilqr
    run()
    best = 1e9
    for _ in range(3):
        s.reset_trajectory()
        t0 = time.perf_counter(); run(); best = min(best, time.perf_counter() - t0)
    st = s.get_stats()
    tm = s.get_timing()
    ok = int((st["status"] == 0).sum())
    print(f"{name}: {best*1e3:.2f} ms, solved {ok}/{len(st)}, sweeps {tm['sweeps']}, mean iterations {st['iterations_total'].mean():.1f}, "
          f"{ok/best:.0f} trajectories/s", flush=True)
    s.close()
