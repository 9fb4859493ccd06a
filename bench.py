#!/usr/bin/env python3
"""bench.py — benchmark of the MI355X-native batched AL-iLQR solver.

Metric (BASELINE.json): trajectories solved / s (AL-iLQR to tolerance) + ms per iLQR iteration.

Workloads (``--config``, index into BASELINE.json ``configs``; default 2 = the configuration the metric
is quoted on):
  1  batch 1024 triple integrator (n=6, m=2, 51 knots), unconstrained iLQR, fp64
  2  batch 4096 unicycle kTurn90 (n=3, m=2, 101 knots), goal + control bounds, full AL loop, fp64   [default]
  3  batch 4096 PER GPU unicycle kThreeObstacles with jittered obstacles, full AL loop, ALTRO_F32
     (with --gpus 8: the 32768-instance workload BASELINE names)
  4  batch 1024 12-state / 4-control model (201 knots), bounds + goal, full AL loop, ALTRO_F32
Seeded synthetic per-instance data; instance 0 of the global batch is the exact
reference problem.  With N > 1 ranks the global batch is N times the per-GPU batch and rank r owns block r
(weak scaling); one process per GPU, no data-path collective.

A "step" is one complete batched solve: device-side reset of the initial guess, the solve to convergence
for every instance, and the result exchange (``sharding.pack_and_gather``: the records are packed on the
device and, with N > 1, ONE RCCL all_gather moves 32 B per instance).  Inputs are resident in HBM before
the timed region.  ``value`` counts only instances that finished with status kSolved.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C] [--batch B] [--no-cpu-baseline]
        (N > 1: launched by torch.distributed.run, one rank per GPU)
"""
import argparse
import ctypes
import glob
import importlib
import json
import os
import sys
import time

# The solver runs the batched sweeps of a large batch on four streams of its own; the HIP runtime maps the streams of a
# process onto GPU_MAX_HW_QUEUES hardware queues (default 4) in order of creation.  Room for the framework's own streams:
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
CLOCK_GHZ = 2.4        # peak engine clock of the MI355X

# n, m, N (segments), constraint rows per stage / terminal knot, bytes per scalar of the SURVEY byte model
CONFIGS = {
    1: dict(factory="batch_triple_integrator", n=6, m=2, N=50, batch=1024, dtype="F64", mode="ilqr", p=(0, 0), item=8,
            seed=2, name="BASELINE configs[1]: batch 1024 triple integrator (n=6, m=2, 51 knots), unconstrained iLQR, fp64"),
    2: dict(factory="batch_turn90", n=3, m=2, N=100, batch=4096, dtype="F64", mode="al", p=(4, 3), item=8, seed=3,
            name="BASELINE configs[2]: batch 4096/GPU unicycle kTurn90 (n=3, m=2, 101 knots), goal + control-bound "
                 "constraints, full AL loop, default SolverOptions"),
    3: dict(factory="batch_three_obstacles", n=3, m=2, N=100, batch=4096, dtype="F32", mode="al", p=(7, 3), item=4,
            seed=4, name="BASELINE configs[3]: batch 4096/GPU (32768 on 8 GPUs) unicycle kThreeObstacles, jittered "
                         "obstacles + control bounds + goal, full AL loop, ALTRO_F32 (fp32 records, fp64 state/arithmetic)"),
    4: dict(factory="batch_quadrotor12", n=12, m=4, N=200, batch=1024, dtype="F32", mode="al", p=(8, 12), item=4,
            seed=5, name="BASELINE configs[4]: batch 1024 12-state/4-control model (201 knots), control bounds + goal, "
                         "full AL loop, ALTRO_F32 (fp32 records, fp64 state/arithmetic)"),
}


def algorithmic_bytes(n, m, N, p_stage, p_term, itemsize):
    """SURVEY.md section 8(d) three-kernel model: bytes per (trajectory, iLQR iteration), split by
    kernel.  Per stage knot (elements): expansions read n+m+2p, write S+1; backward read S, write
    mn+m; forward read (n+m)+(mn+m)+2p, write (n+m)+1, with S = n(n+m)+n^2+nm+m^2+n+m."""
    S = n * (n + m) + n * n + n * m + m * m + n + m
    exp_stage = (n + m + 2 * p_stage) + (S + 1)
    bwd_stage = S + (m * n + m)
    fwd_stage = (n + m) + (m * n + m) + 2 * p_stage + (n + m) + 1
    exp_term = (n + 2 * p_term) + (n * n + n + 1)
    bwd_term = n * n + n
    fwd_term = n + 2 * p_term + n + 1
    per = {
        "expansions": (N * exp_stage + exp_term) * itemsize,
        "backward_pass": (N * bwd_stage + bwd_term) * itemsize,
        "forward_pass": (N * fwd_stage + fwd_term) * itemsize,
    }
    per["total"] = sum(per.values())
    return per


# Dependent-LATENCY floor of one iLQR iteration of ONE instance (the limiter of the persistent tail kernel,
# which carries one straggler per workgroup): the longest chain of dependent instructions per knot, priced
# with the dependent-issue latencies measured on this GPU (profiles/r01_microbench.txt, latency_probe),
# as if every independent instruction were free.  n = 3, m = 2 path (MFMA backward, fused RK4):
#   backward, per knot : W = P A (mfma via B: 29) -> Q2 = B^T W (29) -> Quu to rows 0/1 (dpp 11 + permlane 16)
#                        -> det (fma 12) -> 1/det (rcp 22 + 4 fma 48) -> -adj/det (mul 9) -> [K|d] (29) -> G (29)
#                        -> P' = Q1 + ... three chained mfma (29 + 18 + 29)                              = 310 cycles
#   forward,  per knot : u = ubar + K dx + alpha d (3 fma 36) -> w h / 2 (mul 9) -> sincos kernel on the
#                        increment (8 dependent fma 96) -> angle addition (2 x 12) -> x' (fma fma 24)     = 189 cycles
CHAIN_CYCLES_PER_KNOT = {"backward": 310, "forward": 189}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="index into BASELINE.json configs")
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="EXTRA measurement, not the headline: keep this many solver handles in flight (asynchronous "
                         "solves on their own streams), so the latency-bound tail of one batch overlaps the next batch")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")

    import numpy as np
    import torch
    import torch.distributed as dist

    A = graft.load_package()
    P = importlib.import_module("altro_cpp_amd.problems")
    S = importlib.import_module("altro_cpp_amd.sharding")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)

    cfg = CONFIGS[args.config]
    B = args.batch or cfg["batch"]
    n, m, N = cfg["n"], cfg["m"], cfg["N"]
    dtype = getattr(A, cfg["dtype"])
    factory = getattr(P, cfg["factory"])
    seed = P.SEED_BASE + cfg["seed"]
    total_inst = world * B
    shard = S.shard_range(total_inst, world, rank)  # rank r owns block r of the global seeded batch
    make = lambda n_, m_, N_, b_, d_: A.BatchSolver(n_, m_, N_, b_, d_, device_id=local_rank)  # noqa: E731

    def new_solver():
        s_ = factory(make, batch=total_inst, dtype=dtype, seed=seed, shard=shard)
        s_.set_options(profiler_enable=0)
        return s_

    solver = new_solver()
    solver.num_constraints()  # (creates the device state, and with it the solver's streams, before RCCL's and torch's)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = f"cuda:{local_rank}"
    packed = torch.empty((B, 4), dtype=torch.float64, device=dev)
    gathered = torch.empty((world * B, 4), dtype=torch.float64, device=dev) if world > 1 else packed

    def solve(s_):
        s_.reset_trajectory()
        if cfg["mode"] == "ilqr":
            s_.reset_stats()  # a bare iLQR::Solve accumulates iterations_total across calls (quirk Q11)
            s_.solve_ilqr()
        else:
            s_.solve()

    def step():
        solve(solver)
        S.pack_and_gather(solver, packed, gathered, dist)  # RCCL over xGMI: 32 B per instance

    if args.pipeline > 1:
        if cfg["mode"] != "al":
            raise SystemExit("--pipeline needs an AL config")
        # Optional throughput mode (reported separately, never the headline): P handles with the same
        # problem, each step is still one complete solve, but up to P solves are in flight.
        pool = [solver] + [new_solver() for _ in range(args.pipeline - 1)]
        pending = [False] * len(pool)
        counter = [0]

        def step():  # noqa: F811
            i = counter[0] % len(pool)
            counter[0] += 1
            s_ = pool[i]
            if pending[i]:
                s_.wait()
                s_.pack_results_device(packed.data_ptr())
            s_.reset_trajectory()
            s_.solve_async()
            pending[i] = True

        def drain():
            for i, s_ in enumerate(pool):
                if pending[i]:
                    s_.wait()
                    s_.pack_results_device(packed.data_ptr())
                    pending[i] = False

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if args.pipeline > 1:
        drain()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if args.pipeline > 1:
        drain()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    res = gathered.cpu().numpy()
    status = res[:, 3].astype(int)
    iters = res[:, 2]
    solved = int((status == 0).sum())
    value = solved * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel: one more (untimed) solve with per-kernel HIP events on the
        #      solver's own stream (altro_get_timing) ----
        solver.set_options(profiler_enable=1)
        solve(solver)
        tm = solver.get_timing()
        solver.set_options(profiler_enable=0)
        ab = algorithmic_bytes(n, m, N, cfg["p"][0], cfg["p"][1], cfg["item"])
        # Kernels of one solve: the three sweep kernels while many instances iterate, then (n = 3, m = 2) ONE
        # persistent launch of k_sweep_fused that runs every remaining iteration of the stragglers
        # (expansions + backward + forward of one instance per workgroup).
        kern_ms = {"expansions": tm["expansions_ms"], "backward_pass": tm["backward_pass_ms"],
                   "forward_pass": tm["forward_pass_ms"], "sweep_fused": tm["fused_ms"]}
        # kern_ms are sums of launch durations.  A large batch runs its batched sweeps as several chains on streams of
        # their own: their launches overlap, so the three sweep kernels can add up to more than the
        # wall time.  The dominant kernel is the one with the largest share of the solve's WALL time: the sweep kernels
        # share what the solve took outside the initialisation and the persistent launch.
        sweep_sum = kern_ms["expansions"] + kern_ms["backward_pass"] + kern_ms["forward_pass"]
        sweep_wall = min(sweep_sum, max(0.0, tm["total_ms"] - tm["init_ms"] - tm["fused_ms"]))
        scale = sweep_wall / sweep_sum if sweep_sum > 0 else 1.0
        wall_ms = {k: (v if k == "sweep_fused" else v * scale) for k, v in kern_ms.items()}
        dom = max(wall_ms, key=wall_ms.get)
        units_total = tm["instance_iterations"]  # (trajectory, iteration) units of the whole solve
        units_fused = tm["fused_instance_iterations"]
        if dom == "sweep_fused":
            launches, units, bytes_per_unit = 1, units_fused, ab["total"]
        else:
            launches = tm.get("sweep_launches") or (tm["sweeps"] - tm["fused_sweeps"])  # all chains of sweeps together
            units, bytes_per_unit = units_total - units_fused, ab[dom]
        launches = max(launches, 1)
        avg_launch_ms = kern_ms[dom] / launches
        achieved = bytes_per_unit * units / launches / (avg_launch_ms * 1e-3) / 1e9  # GB/s
        sweep_ms = sum(wall_ms.values())
        achieved_all = ab["total"] * units_total / (sweep_ms * 1e-3) / 1e9
        # HBM traffic per launch of the dominant kernel, from the committed rocprofv3 PMC passes of this
        # same command (FETCH_SIZE / WRITE_SIZE in separate --pmc runs, gfx950 correction applied --
        # see profiles/rNN_traffic.json); bench.py itself cannot run the profiler.
        traffic, traffic_src = None, None
        key = {"expansions": "k_expansions", "backward_pass": "k_backward", "forward_pass": "k_forward",
               "sweep_fused": "k_sweep_fused"}[dom]
        tag = "" if args.config == 2 else f"_config{args.config}"
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*{tag}_traffic.json")), reverse=True):
            if args.config == 2 and "_config" in os.path.basename(f):
                continue
            try:
                tj = json.load(open(f))
                if key in tj:
                    traffic = round(tj[key]["traffic_bytes_per_launch"])
                    traffic_src = os.path.relpath(f, ROOT)
                    break
            except Exception:
                pass
        roofline = {
            "bound": "hbm", "kernel": key, "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic, "traffic_source": traffic_src,
            "avg_launch_us": round(1e3 * avg_launch_ms, 2), "launches": launches,
            "units_per_launch": round(units / launches, 1),
            "algorithmic_bytes_per_unit": bytes_per_unit,
            "algorithmic_bytes_per_launch": round(bytes_per_unit * units / launches),
            "all_kernels_achieved": round(achieved_all, 2),
            "all_kernels_frac": round(achieved_all / HBM_PEAK_GBS, 5),
            "kernel_ms": {k: round(v, 3) for k, v in kern_ms.items()},
            "kernel_wall_ms": {k: round(v, 3) for k, v in wall_ms.items()},
            "sweep_launches": tm.get("sweep_launches", 0),
            "concurrent_chains": max(1, round(tm.get("sweep_launches", 0) / max(1, tm["sweeps"] - tm["fused_sweeps"]))),
            "tail_iterations": tm["fused_sweeps"],
        }
        if tm["fused_sweeps"] > 0 and n == 3 and m == 2:
            # the real limiter of the dominant launch: the dependent chain of one instance's iteration
            # (chain_floor_us prices the two chains one after the other, as the reference runs them.  The persistent
            #  kernel overlaps them whenever a line search rejects every trial -- the fourth wave has the next backward
            #  pass ready by then -- so the floor of such an iteration is the longer chain alone: chain_floor_overlap_us)
            floor_us = sum(CHAIN_CYCLES_PER_KNOT.values()) * N / (CLOCK_GHZ * 1e3)
            floor_overlap_us = max(CHAIN_CYCLES_PER_KNOT.values()) * N / (CLOCK_GHZ * 1e3)
            iter_us = 1e3 * tm["fused_ms"] / tm["fused_sweeps"]
            roofline.update({
                "chain_floor_us": round(floor_us, 2), "chain_floor_overlap_us": round(floor_overlap_us, 2),
                "tail_iteration_us": round(iter_us, 2),
                "chain_floor_frac": round(floor_us / iter_us, 4),
                "limiter": "serial dependency chain: the persistent tail kernel carries one straggler instance per "
                           "workgroup through ~100 iterations x (N Riccati steps + N RK4 steps); chain_floor_us is "
                           "the dependent-instruction latency of one iteration (measured issue latencies, "
                           "profiles/r01_microbench.txt), tail_iteration_us what one iteration takes",
            })
        # ---- CPU baseline: the oracle (a port, see oracle/altro_oracle.cpp) on the host cores --------
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            lib_path = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
            lib = ctypes.CDLL(lib_path)
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            omake = lambda n_, m_, N_, b_, d_: A.BatchSolver(n_, m_, N_, b_, d_, _lib=lib, _prefix="oracle_")  # noqa: E731
            osolve = (lambda s_: s_.solve_ilqr()) if cfg["mode"] == "ilqr" else (lambda s_: s_.solve())
            obench = lib.oracle_bench_ilqr if cfg["mode"] == "ilqr" else lib.oracle_bench_al
            # single thread first: calibrates how many repetitions make ~10-30 s of CPU work
            o1 = factory(omake, batch=64, dtype=A.F64, seed=seed)
            lib.oracle_set_threads(o1._h, ctypes.c_int(1))
            c1 = time.perf_counter()
            osolve(o1)
            cdt1 = time.perf_counter() - c1
            o1st = o1.get_stats()
            per_inst = cdt1 / 64
            reps = int(max(1, min(256, round(20.0 / (per_inst * B)))))  # ~20 CPU-seconds in total
            o = factory(omake, batch=B, dtype=A.F64, seed=seed)
            lib.oracle_set_threads(o._h, ctypes.c_int(cores))
            c0 = time.perf_counter()
            obench(o._h, ctypes.c_int(reps))
            cdt = time.perf_counter() - c0
            ost = o.get_stats()
            cpu = {
                "value": round(float((ost["status"] == 0).sum()) * reps / cdt, 2), "unit": "trajectories/s",
                "cores": cores, "kind": "port",
                "sample": f"the same seeded {B}-instance workload (fp64) solved {reps}x back to back by {cores} host "
                          f"threads (one instance per task, one thread team), about {per_inst * B * reps:.0f} CPU-seconds",
                "single_thread_value": round(float((o1st["status"] == 0).sum()) / cdt1, 2),
                "single_thread_ms_per_ilqr_iter": round(1e3 * cdt1 / float(o1st["iterations_total"].sum()), 4),
            }
        # ---- the same step with the problem handed over as HOST buffers (the C-ABI boundary of an MPC caller: a new
        #      initial state and a warm start per instance go up, trajectories and statistics come back), PCIe and
        #      the host-side layout conversion included.  Reported beside `value`, never as `value`. ----
        host = None
        if args.pipeline == 1 and world == 1:
            solver.reset_trajectory()
            X0, U0 = solver.get_trajectory()
            x0 = np.ascontiguousarray(X0[:, 0, :])
            Uout = np.empty_like(U0)
            hreps = max(1, min(args.steps, 3))
            h0 = 0.0
            for hi in range(hreps + 1):
                if hi == 1:  # (the first pass is a warm-up: staging buffer, first touch of the result arrays)
                    torch.cuda.synchronize()
                    h0 = time.perf_counter()
                solver.set_initial_state(x0)
                solver.set_trajectory(None, U0)
                if cfg["mode"] == "ilqr":
                    solver.reset_stats()
                    solver.solve_ilqr()
                else:
                    solver.solve()
                Xs, Us = solver.get_trajectory(X0, Uout)  # (the caller's own buffers, reused every step)
                hst = solver.get_stats()
            hdt = (time.perf_counter() - h0) / hreps
            host = {
                "value": round(float((hst["status"] == 0).sum()) / hdt, 1), "unit": "trajectories/s",
                "ms_per_step": round(1e3 * hdt, 3), "steps": hreps,
                "bytes_up": int(x0.nbytes + U0.nbytes), "bytes_down": int(Xs.nbytes + Us.nbytes),
                "note": "altro_set_initial_state + altro_set_trajectory from pageable host arrays, solve, "
                        "altro_get_trajectory + altro_get_stats into host arrays (the caller's buffers, reused)",
            }
        name, cus = solver.device_info()
        out = {
            "metric": "trajectories solved/sec (AL-iLQR to tol), unicycle 101 knots, batched" if args.config in (2, 3)
                      else "trajectories solved/sec (to tol), batched",
            "value": round(value, 1), "unit": "trajectories/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if cfg["dtype"] == "F64" else "f64 arithmetic and state, f32 expansion/gain records (ALTRO_F32)",
            "data": "synthetic",
            "config": {
                "workload": cfg["name"], "baseline_config_index": args.config,
                "batch_per_gpu": B, "global_batch": total_inst, "knot_points": N + 1,
                "parallelism": f"instance-sharded x{world} (no data-path collective; RCCL all_gather of result records)",
                "solved_fraction": round(solved / total_inst, 5),
                "ms_per_ilqr_iter_sweep": round(ms_per_step / max(tm["sweeps"], 1), 4),
                "us_per_instance_iter": round(1e3 * ms_per_step / max(float(iters.sum()) / world, 1.0), 4),
                "mean_iterations": round(float(iters.mean()), 3), "max_iterations": int(iters.max()),
                "sweeps": tm["sweeps"], "device": name, "cus": cus,
                **({"pipeline": args.pipeline, "note": "NOT the headline configuration: solves of consecutive steps "
                    "overlap (asynchronous handles)"} if args.pipeline > 1 else {}),
                **({"fast_forward_stalls": True, "note_ff": "NOT the headline configuration: ALTRO_HIP_FAST_FORWARD_STALLS "
                    "counts the bit-identical repetitions of a rejected line search instead of recomputing them"}
                   if os.environ.get("ALTRO_HIP_FAST_FORWARD_STALLS") else {}),
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
            "host_boundary": host,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
