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

Beside the headline the default line (N = 1) carries: ``cpu_baseline`` (the oracle on the host cores, pinned
threads, instances built outside the timed region), ``host_boundary`` (the same step over host buffers),
``other_configs`` (BASELINE configs 1, 3, 4: a few steps each after the headline) and ``latency`` (one solve of a
batch of 1 / 8 / 64 -- the MPC use case -- next to one host core).

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C] [--batch B] [--no-cpu-baseline]
                        [--no-other-configs] [--no-latency] [--no-fast-forward] [--no-pipeline2] [--no-numa-bind]
        (N > 1: one rank per GPU -- under torch.distributed.run as the driver launches it, or started plainly: bench.py
         then launches its N ranks itself, `self_launch`.  `--launch-check`: the rank plumbing alone, no GPU, gloo.)
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


def dominant_kernel_roofline(cfg, tm, config_index):
    """Roofline block of one profiled solve (``tm`` = altro_get_timing after a solve with profiler_enable)."""
    n, m, N = cfg["n"], cfg["m"], cfg["N"]
    ab = algorithmic_bytes(n, m, N, cfg["p"][0], cfg["p"][1], cfg["item"])
    # Kernels of one solve: the three sweep kernels while many instances iterate, then (n = 3, m = 2) ONE
    # persistent launch of k_sweep_fused that runs every remaining iteration of the stragglers
    # (expansions + backward + forward of one instance per workgroup).
    kern_ms = {"expansions": tm["expansions_ms"], "backward_pass": tm["backward_pass_ms"],
               "forward_pass": tm["forward_pass_ms"], "sweep_fused": tm["fused_ms"]}
    # (round 6: a batch that fits the slots of the device-side sweep loop runs its bulk phase as ONE launch of k_sweep_loop)
    loop_ms = tm.get("loop_ms", 0.0) or 0.0
    if loop_ms > 0:
        kern_ms["sweep_loop"] = loop_ms
    # kern_ms are sums of launch durations.  A large batch runs its batched sweeps as several chains on streams of
    # their own: their launches overlap, so the three sweep kernels can add up to more than the
    # wall time.  The dominant kernel is the one with the largest share of the solve's WALL time: the sweep kernels
    # share what the solve took outside the initialisation and the persistent launch.
    sweep_sum = kern_ms["expansions"] + kern_ms["backward_pass"] + kern_ms["forward_pass"]
    sweep_wall = min(sweep_sum, max(0.0, tm["total_ms"] - tm["init_ms"] - tm["fused_ms"] - loop_ms))
    scale = sweep_wall / sweep_sum if sweep_sum > 0 else 1.0
    wall_ms = {k: (v if k in ("sweep_fused", "sweep_loop") else v * scale) for k, v in kern_ms.items()}
    dom = max(wall_ms, key=wall_ms.get)
    units_total = tm["instance_iterations"]  # (trajectory, iteration) units of the whole solve
    units_fused = tm["fused_instance_iterations"]
    units_loop = tm.get("loop_instance_iterations", 0) or 0
    if dom == "sweep_fused":
        launches, units, bytes_per_unit = max(1, tm.get("fused_launches", 1) or 1), units_fused, ab["total"]
    elif dom == "sweep_loop":
        launches, units, bytes_per_unit = 1, units_loop, ab["total"]  # (whole iterations: E + B + F of every unit)
    else:
        launches = tm.get("sweep_launches") or (tm["sweeps"] - tm["fused_sweeps"])  # all chains of sweeps together
        units, bytes_per_unit = units_total - units_fused - units_loop, ab[dom]
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
           "sweep_fused": "k_sweep_fused", "sweep_loop": "k_sweep_loop"}[dom]
    tag = "" if config_index == 2 else f"_config{config_index}"
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*{tag}_traffic.json")), reverse=True):
        if config_index == 2 and "_config" in os.path.basename(f):
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
        "tail_iterations": tm["fused_sweeps"],  # longest chain of iterations of one instance inside the persistent launch
        "tail_workgroup_iterations": tm.get("fused_workgroup_iterations", 0) or tm["fused_sweeps"],  # ... of one WORKGROUP (a twin or its primary: their share)
        # twin workgroups of the persistent launch: launched / claimed a streak's second half / confirmed
        "twin_workgroups": tm.get("twin_workgroups", 0), "twin_claims": tm.get("twin_claims", 0),
        "twin_handovers": tm.get("twin_handovers", 0),
        # segments of rejection streaks in the batched sweeps: shadow columns used by this solve
        "segment_columns": tm.get("segment_columns", 0),
        # the OTHER roof: how close the kernel's fp64 vector work comes to the fp64 VALU peak (PMC pass in profiles/)
        "compute": compute_side(key, config_index, 1e3 * avg_launch_ms),
    }
    if tm["fused_sweeps"] > 0 and n == 3 and m == 2:
        # the real limiter of the dominant launch: the dependent chain of one instance's iteration
        # (chain_floor_us prices the two chains one after the other, as the reference runs them.  The persistent
        #  kernel overlaps them whenever a line search rejects every trial -- the fourth wave has the next backward
        #  pass ready by then -- so the floor of such an iteration is the longer chain alone: chain_floor_overlap_us)
        floor_us = sum(CHAIN_CYCLES_PER_KNOT.values()) * N / (CLOCK_GHZ * 1e3)
        floor_overlap_us = max(CHAIN_CYCLES_PER_KNOT.values()) * N / (CLOCK_GHZ * 1e3)
        iter_us = 1e3 * tm["fused_ms"] / max(1, tm.get("fused_workgroup_iterations", 0) or tm["fused_sweeps"])
        roofline.update({
            "chain_floor_us": round(floor_us, 2), "chain_floor_overlap_us": round(floor_overlap_us, 2),
            "tail_iteration_us": round(iter_us, 2),
            "chain_floor_frac": round(floor_us / iter_us, 4),
            "limiter": "serial dependency chain: the persistent tail kernel carries one straggler instance per "
                       "workgroup through ~100 iterations x (N Riccati steps + N RK4 steps); chain_floor_us is "
                       "the dependent-instruction latency of one iteration (measured issue latencies, "
                       "profiles/r01_microbench.txt), tail_iteration_us = launch duration / the most iterations any one "
                       "workgroup ran (with twin workgroups the ~100 rejected iterations of a straggler are shared by two "
                       "workgroups: the figure then includes the twin's start-up and the hand-over)",
        })
    return roofline


def load_oracle():
    """The CPU oracle (test infrastructure): ONLY the cpu_baseline / latency legs below call it, as the thing a GPU number
    is reported beside -- never inside a timed GPU region."""
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
    lib.oracle_bench_seconds.restype = ctypes.c_double
    lib.oracle_bench_busy.restype = ctypes.c_double
    return lib


def cpu_quota_cores():
    """CPUs the container may use per wall second (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cfs_period_us), or None.
    The GPU boxes expose all 256 hardware threads of the host but run the container under a quota (measured in round 3:
    `1600000 100000` = 16 CPUs -- scripts/probe_cpu_scaling.py, profiles/r03_cpu_scaling.txt): more threads than the quota
    are descheduled, which is what made round 2's 256-thread figure look like 6 % parallel efficiency."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / p if q > 0 else None
    except Exception:
        return None


def cpu_baseline(A, P, cfg, B, seed, budget_cpu_s=24.0):
    """The oracle (kind "port": the literal Eigen path cannot be built here) on the host cores, on a
    bounded sample of the SAME workload.  Sound by construction: the instances are built outside the
    timed region (oracle_prepare), one warm-up pass touches every array, the threads are pinned (one per physical core
    first), the clock runs inside the library from the team's start barrier to its last task, and a task is one solve of
    one instance handed out repetition-major (the tail of the run is one straggler solve).  Two team sizes are timed --
    one thread per usable core, and twice that (SMT / oversubscription) -- and the better one is `value`.  "Usable" honours
    the container's CPU quota (cpu_quota_cores): threads beyond it only get descheduled."""
    lib = load_oracle()
    lib.oracle_bench_cpu_seconds.restype = ctypes.c_double
    factory = getattr(P, cfg["factory"])
    ilqr = cfg["mode"] == "ilqr"
    obench = lib.oracle_bench_ilqr if ilqr else lib.oracle_bench_al
    omake = lambda n_, m_, N_, b_, d_: A.BatchSolver(n_, m_, N_, b_, d_, _lib=lib, _prefix="oracle_")  # noqa: E731
    hw = int(lib.oracle_host_threads())
    phys = int(lib.oracle_host_physical_cores())
    quota = cpu_quota_cores()
    usable = max(1, min(phys, int(round(quota)))) if quota else phys  # cores the team can really run on at once
    o = factory(omake, batch=B, dtype=A.F64, seed=seed)
    lib.oracle_prepare(o._h)
    # warm-up + calibration: one pass over the batch (also sizes the per-instance vectors)
    lib.oracle_set_threads(o._h, ctypes.c_int(min(hw, 2 * usable)))
    obench(o._h, ctypes.c_int(1))
    pass_s = lib.oracle_bench_seconds(o._h)
    st = o.get_stats()
    iters_pass = float(st["iterations_total"].sum())
    solved_pass = float((st["status"] == 0).sum())
    # single thread: a slice of the batch with every 64th instance (the stragglers are spread over the batch), so
    # that per-iteration and per-trajectory cost are those of the same instance mix
    idx_stride = max(1, B // 64)
    o1 = factory(omake, batch=B, dtype=A.F64, seed=seed)  # same batch; only the sampled instances are solved
    lib.oracle_prepare(o1._h)
    lib.oracle_set_threads(o1._h, ctypes.c_int(1))
    lib.oracle_set_ilqr_mode(o1._h, ctypes.c_int(1 if ilqr else 0))
    lib.oracle_bench_subset(o1._h, ctypes.c_int(idx_stride), ctypes.c_int(1))  # warm-up
    lib.oracle_bench_subset(o1._h, ctypes.c_int(idx_stride), ctypes.c_int(1))
    t1 = lib.oracle_bench_seconds(o1._h)
    s1 = o1.get_stats()
    sel = slice(0, B, idx_stride)
    iters1 = float(s1["iterations_total"][sel].sum())
    solved1 = float((s1["status"][sel] == 0).sum())
    runs = []
    for nt in sorted({usable, min(hw, 2 * usable)}):
        lib.oracle_set_threads(o._h, ctypes.c_int(nt))
        # ~budget/2 CPU-seconds per team size: reps from the single-thread cost of one pass
        est_cpu_s_per_pass = t1 * (iters_pass / max(iters1, 1.0))
        reps = int(max(1, min(512, round(0.5 * budget_cpu_s / max(est_cpu_s_per_pass, 1e-6)))))
        obench(o._h, ctypes.c_int(reps))
        sec = lib.oracle_bench_seconds(o._h)
        runs.append({"threads": nt, "reps": reps, "seconds": round(sec, 4),
                     "value": round(solved_pass * reps / sec, 2),
                     "iterations_per_s": round(iters_pass * reps / sec, 1),
                     "busy_min_s": round(lib.oracle_bench_busy(o._h, 0), 4),
                     "busy_max_s": round(lib.oracle_bench_busy(o._h, 1), 4),
                     # CPU time the team's threads were given / (threads x wall): < 1 = descheduled (quota, other tenants)
                     "cpu_time_fraction": round(lib.oracle_bench_cpu_seconds(o._h) / (nt * sec), 3)})
    best = max(runs, key=lambda r: r["value"])
    it_rate_1 = iters1 / t1
    return {
        "value": best["value"], "unit": "trajectories/s", "cores": best["threads"], "kind": "port",
        # (the literal reference path needs Eigen >= 3.3: probed, never found on a box of this pool -- BASELINE.md section 2)
        "eigen_headers_found": any(os.path.exists(os.path.join(d, "Eigen", "Core")) for d in
                                   ("/usr/include/eigen3", "/usr/local/include/eigen3", "/usr/include", "/opt/rocm/include/eigen3")),
        "physical_cores": phys, "hardware_threads": hw,
        "cpu_quota_cores": quota, "usable_cores": usable,
        "sample": f"the same seeded {B}-instance workload (fp64) solved {best['reps']}x by {best['threads']} pinned host "
                  f"threads (one solve of one instance per task; instances built and warmed up outside the timed "
                  f"region; clock inside the library from the team's start barrier to its last task), "
                  f"{best['seconds']:.2f} s wall = about {best['seconds'] * best['threads']:.0f} CPU-seconds",
        "single_thread_value": round(solved1 / t1, 2),
        "single_thread_ms_per_ilqr_iter": round(1e3 / it_rate_1, 4),
        "single_thread_sample": f"every {idx_stride}th instance of the batch ({len(range(0, B, idx_stride))} solves)",
        # per-iteration throughput of the team / (one thread's x the cores the container may use at once): SMT, all-core
        # clocks and the other tenants of the host included
        "parallel_efficiency": round(best["iterations_per_s"] / (it_rate_1 * usable), 4),
        "parallel_efficiency_vs_physical_cores": round(best["iterations_per_s"] / (it_rate_1 * phys), 4),
        "teams": runs,
    }


def self_launch(ngpus, argv):
    """`python bench.py --gpus N` started plainly (no WORLD_SIZE in the environment): start the N ranks ourselves --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>
    bench.py <the same arguments>` -- one rank per GPU; rank 0 of that job prints the ONE JSON line on our stdout."""
    import subprocess
    # (--standalone: torchrun's c10d rendezvous picks its own free port -- no bind / close / reuse race between eight
    #  benchmark jobs starting together; --local-addr: the container's hostname may not resolve)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(ngpus), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (the host driver only supports dmabuf IPC: RCCL needs it)
    env["ALTRO_BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.run(cmd, env=env).returncode


def bind_to_gpu_numa_node(torch, local_rank):
    """Multi-rank runs: keep this rank's host threads (the sweep loop's polling thread, the async worker, RCCL's proxy) on
    the NUMA node its GPU hangs off -- the pinned, mapped counters the device publishes and the kernel launches then stay
    on the socket next to the GPU.  Best effort: returns what was done for the bench line, or None."""
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
        if node < 0:
            return {"pci": bdf, "numa_node": node, "bound": False, "why": "the platform reports no NUMA node for the device"}
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0) & cpus
        if not allowed:
            return {"pci": bdf, "numa_node": node, "bound": False, "why": "no CPU of the node in this process's affinity mask"}
        os.sched_setaffinity(0, allowed)
        return {"pci": bdf, "numa_node": node, "bound": True, "cpus": len(allowed)}
    except Exception as e:  # (never fail a run for an affinity hint)
        return {"bound": False, "why": str(e)[:120]}


def launch_check(args, rank, local_rank, world):
    """`--launch-check`: the rank / shard plumbing of the N-rank launch WITHOUT a GPU (gloo): every rank reports what it
    would do -- its rank, local rank, device ordinal, block of the global seeded batch -- one all_gather, and rank 0
    prints one JSON line with the checks (tests/test_bench_launch.py runs it with two ranks)."""
    import torch
    import torch.distributed as dist
    graft.load_package()
    S = importlib.import_module("altro_cpp_amd.sharding")
    cfg = CONFIGS[args.config]
    B = args.batch or cfg["batch"]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("gloo")
    lo, hi = S.shard_range(world * B, world, rank)
    mine = torch.tensor([rank, local_rank, lo, hi, os.getpid(), dist.get_world_size()], dtype=torch.int64)
    rows = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(rows, mine)
    rows = [r.tolist() for r in rows]
    out = None
    if rank == 0:
        blocks = [(r[2], r[3]) for r in rows]
        out = {"launch_check": {
            "backend": dist.get_backend(), "world_size": dist.get_world_size(), "env_world_size": world,
            "ranks": [r[0] for r in rows], "local_ranks": [r[1] for r in rows], "blocks": blocks,
            "distinct_processes": len({r[4] for r in rows}) == len(rows),
            "blocks_tile_the_global_batch": blocks[0][0] == 0 and blocks[-1][1] == world * B and
                                            all(blocks[i][1] == blocks[i + 1][0] for i in range(len(blocks) - 1)),
            "every_rank_saw_the_same_world": all(r[5] == dist.get_world_size() for r in rows),
            "self_launched": os.environ.get("ALTRO_BENCH_SELF_LAUNCHED") == "1",
            "args": {"gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "config": args.config, "batch": B}},
            "n_gpus": dist.get_world_size()}
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return out


def measure_hbm_copy_gbs(torch, dev, mib=1024, reps=10):
    """The achievable HBM ceiling of THIS device next to the 8 000 GB/s of the data sheet: a
    device-to-device copy of `mib` MiB (far beyond the 256 MiB Infinity Cache), bytes read + bytes written per second."""
    n = mib * 1024 * 1024 // 8
    a = torch.empty(n, dtype=torch.float64, device=dev).normal_()
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    return 2.0 * n * 8 / (ms * 1e-3) / 1e9


FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz


def compute_side(key, config_index, avg_launch_us):
    """The compute-side fraction of the dominant kernel: fp64 vector flops per launch from the committed rocprofv3 PMC pass
    (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64: wave-level instruction counts x 64 lanes, FMA = 2 flop; MFMA fp64 flops from
    SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 are listed beside them), divided by the LIVE launch duration and the 78.6 TFLOP/s
    fp64 vector peak.  None when profiles/ has no such pass for this kernel."""
    tag = "" if config_index == 2 else f"_config{config_index}"
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*{tag}_traffic.json")), reverse=True):
        if config_index == 2 and "_config" in os.path.basename(f):
            continue
        try:
            tj = json.load(open(f)).get(key)
            if not tj or tj.get("fp64_valu_flop_per_launch") is None:
                continue
            flop = float(tj["fp64_valu_flop_per_launch"])
            # the profile's launch may differ in size from the live one: scale by the time, i.e. use the PROFILE's own
            # duration for its own flops (both from the same launches)
            us = float(tj["avg_launch_us"])
            tf = flop / (us * 1e-6) / 1e12
            # (named for what it is: a figure of the COMMITTED profile -- its flops over its own launch time -- not of this run)
            return {"bound": "valu_f64", "profile_achieved": round(tf, 3), "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "profile_frac": round(tf / FP64_VALU_PEAK_TFLOPS, 5), "fp64_valu_flop_per_launch": round(flop),
                    "profile_avg_launch_us": round(us, 2), "live_avg_launch_us": round(avg_launch_us, 2),
                    "valu_busy_frac": tj.get("valu_busy_frac"), "source": os.path.relpath(f, ROOT)}
        except Exception:
            pass
    return None


def solve_once(s_, mode):
    s_.reset_trajectory()
    if mode == "ilqr":
        s_.reset_stats()  # a bare iLQR::Solve accumulates iterations_total across calls (quirk Q11)
        s_.solve_ilqr()
    else:
        s_.solve()


def measure_other_config(A, P, S, torch, ci, steps, warmup, device_id):
    """A few steps of another BASELINE config after the headline (same step definition, one GPU): its own handle,
    created after the headline's was destroyed (so it gets its chains of sweeps like a fresh process would)."""
    cfg = CONFIGS[ci]
    B = cfg["batch"]
    make = lambda n_, m_, N_, b_, d_: A.BatchSolver(n_, m_, N_, b_, d_, device_id=device_id)  # noqa: E731
    s_ = getattr(P, cfg["factory"])(make, batch=B, dtype=getattr(A, cfg["dtype"]), seed=P.SEED_BASE + cfg["seed"],
                                     shard=S.shard_range(B, 1, 0))
    s_.set_options(profiler_enable=0)
    packed = torch.empty((B, 4), dtype=torch.float64, device=f"cuda:{device_id}")
    for _ in range(warmup):
        solve_once(s_, cfg["mode"])
        S.pack_and_gather(s_, packed, packed, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        solve_once(s_, cfg["mode"])
        S.pack_and_gather(s_, packed, packed, None)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    res = packed.cpu().numpy()
    solved = int((res[:, 3].astype(int) == 0).sum())
    s_.set_options(profiler_enable=1)
    solve_once(s_, cfg["mode"])
    tm = s_.get_timing()
    rl = dominant_kernel_roofline(cfg, tm, ci)
    s_.close()
    return {
        "workload": cfg["name"], "value": round(solved * steps / el, 1), "unit": "trajectories/s",
        "ms_per_step": round(1e3 * el / steps, 3), "steps": steps, "warmup": warmup,
        "solved_fraction": round(solved / B, 5),
        "dtype": "f64" if cfg["dtype"] == "F64" else "f64 arithmetic, f32 records (ALTRO_F32)",
        "dominant_kernel": rl["kernel"], "frac": rl["frac"], "achieved_gbs": rl["achieved"], "compute": rl.get("compute"),
        "avg_launch_us": rl["avg_launch_us"], "launches": rl["launches"],
        "all_kernels_frac": rl["all_kernels_frac"], "kernel_wall_ms": rl["kernel_wall_ms"],
        "concurrent_chains": rl["concurrent_chains"], "sweeps": tm["sweeps"], "tail_iterations": tm["fused_sweeps"],
        "segment_columns": rl["segment_columns"],
    }


def measure_device_loop(A, P, S, torch, device_id, batch=1536, steps=5, warmup=1):
    """SECONDARY, never `value` (round 6): the device-side sweep loop against the host-paced chains of sweeps on the largest
    batch the loop takes by default -- config 2's problem with `batch` instances (the slots of the resident workgroups: 256
    CUs x 2 x 3) -- same step definition as the headline; per mode: ms per step, kernel launches per solve, CPU time of the
    rank per wall second, and the two modes' result records compared bit for bit."""
    import numpy as np
    cfg = CONFIGS[2]
    rows, recs = {}, {}
    for mode, env in (("loop", None), ("sweeps", "0")):
        if env is None:
            os.environ.pop("ALTRO_HIP_SWEEP_LOOP", None)
        else:
            os.environ["ALTRO_HIP_SWEEP_LOOP"] = env  # (read when the engine uploads the problem)
        try:
            make = lambda n_, m_, N_, b_, d_: A.BatchSolver(n_, m_, N_, b_, d_, device_id=device_id)  # noqa: E731
            s_ = getattr(P, cfg["factory"])(make, batch=batch, dtype=getattr(A, cfg["dtype"]), seed=P.SEED_BASE + cfg["seed"],
                                             shard=S.shard_range(batch, 1, 0))
            s_.set_options(profiler_enable=0)
            packed = torch.empty((batch, 4), dtype=torch.float64, device=f"cuda:{device_id}")
            for _ in range(warmup):
                solve_once(s_, cfg["mode"])
                S.pack_and_gather(s_, packed, packed, None)
            torch.cuda.synchronize()
            c0, t0 = time.process_time(), time.perf_counter()
            for _ in range(steps):
                solve_once(s_, cfg["mode"])
                S.pack_and_gather(s_, packed, packed, None)
            torch.cuda.synchronize()
            el, cpu = time.perf_counter() - t0, time.process_time() - c0
            recs[mode] = packed.cpu().numpy().copy()
            s_.set_options(profiler_enable=1)
            solve_once(s_, cfg["mode"])
            tm = s_.get_timing()
            solved = int((recs[mode][:, 3].astype(int) == 0).sum())
            rows[mode] = {"ms_per_step": round(1e3 * el / steps, 3), "value": round(solved * steps / el, 1),
                          "host_cpu_cores": round(cpu / max(el, 1e-9), 3), "kernel_launches_per_solve": tm["launches"],
                          "sweep_launches": tm["sweep_launches"], "loop_ms": round(tm.get("loop_ms", 0.0), 3),
                          "loop_instance_iterations": tm.get("loop_instance_iterations", 0), "persistent_ms": round(tm["fused_ms"], 3),
                          "handed_to_persistent_kernel": tm.get("loop_handover", 0)}
            s_.close()
        finally:
            os.environ.pop("ALTRO_HIP_SWEEP_LOOP", None)
    return {"workload": f"BASELINE configs[2]'s problem, batch {batch} (fits the loop's slots)", "unit": "trajectories/s", "steps": steps,
            "loop": rows.get("loop"), "sweeps": rows.get("sweeps"),
            "records_identical": bool(np.array_equal(recs.get("loop"), recs.get("sweeps"))),
            "note": "NOT the headline: the batch the headline is quoted on (4096) exceeds the loop's slots and runs the host-paced "
                    "sweeps (profiles/r06_experiments.txt #1)"}


def measure_latency(A, P, device_id, with_cpu=True):
    """The MPC use case (reference docs/Overview.dox:48-54, test/augmented_lagrangian/auglag_test.cpp:353-380): wall time
    of ONE altro_solve_al call for a batch of 1 / 8 / 64 -- cold (the reference problem's initial guess) and warm-started
    (reset_duals = false, initial_penalty = 0, previous solution kept, the initial state moved a little) -- next to the
    oracle on one host core for the batch of 1."""
    import numpy as np
    lib = load_oracle() if with_cpu else None
    out = {}
    for name, factory in (("kTurn90", "batch_turn90"), ("kThreeObstacles", "batch_three_obstacles")):
        rows = {}
        for B in (1, 8, 64):
            make = lambda n_, m_, N_, b_, d_: A.BatchSolver(n_, m_, N_, b_, d_, device_id=device_id)  # noqa: E731
            g = getattr(P, factory)(make, batch=B, dtype=A.F64)
            g.set_options(profiler_enable=0)

            def timed(s_, fn, reps=5):
                best = 1e30
                for _ in range(reps):
                    fn(s_)
                    t0 = time.perf_counter()
                    s_.solve()
                    best = min(best, time.perf_counter() - t0)
                return best
            cold = timed(g, lambda s_: s_.reset_trajectory())
            st = g.get_stats()
            # warm start: keep duals, penalties and the solution; the measured state differs a little from the plan
            g.set_options(reset_duals=0, initial_penalty=0.0)
            Xs, Us = g.get_trajectory()
            x0 = np.ascontiguousarray(Xs[:, 0, :])
            k = [0]

            def nudge(s_):
                k[0] += 1
                s_.set_initial_state(x0 + 1e-3 * ((k[0] % 3) - 1))
            warm = timed(g, nudge)
            sw = g.get_stats()
            row = {"cold_ms": round(1e3 * cold, 3), "cold_iterations_max": int(st["iterations_total"].max()),
                   "warm_ms": round(1e3 * warm, 3), "warm_iterations_max": int(sw["iterations_total"].max())}
            g.close()
            if lib is not None and B == 1:
                omake = lambda n_, m_, N_, b_, d_: A.BatchSolver(n_, m_, N_, b_, d_, _lib=lib, _prefix="oracle_")  # noqa: E731
                o = getattr(P, factory)(omake, batch=1, dtype=A.F64)
                lib.oracle_set_threads(o._h, ctypes.c_int(1))
                lib.oracle_prepare(o._h)
                ocold = timed(o, lambda s_: s_.reset_trajectory())
                o.set_options(reset_duals=0, initial_penalty=0.0)
                Xo, _ = o.get_trajectory()
                xo = np.ascontiguousarray(Xo[:, 0, :])
                k[0] = 0

                def onudge(s_):
                    k[0] += 1
                    s_.set_initial_state(xo + 1e-3 * ((k[0] % 3) - 1))
                owarm = timed(o, onudge)
                row.update({"cpu_1thread_cold_ms": round(1e3 * ocold, 3), "cpu_1thread_warm_ms": round(1e3 * owarm, 3)})
            rows[f"batch_{B}"] = row
        out[name] = rows
    out["note"] = ("best of 5 blocking altro_solve_al calls through the C-ABI, problem resident on the device; "
                   "cpu_1thread_* = the oracle (port) on one host core for the same single instance")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="index into BASELINE.json configs")
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the all-gather also with ONE rank: the 1-GPU test of the "
                         "multi-GPU path (tests/test_rccl_world1_gpu.py); the line gains a `dist_check` key")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the extra key with BASELINE configs 1, 3, 4")
    ap.add_argument("--no-latency", action="store_true", help="skip the extra key with batch-of-1/8/64 solve latencies")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="EXTRA measurement, not the headline: keep this many solver handles in flight (asynchronous "
                         "solves on their own streams), so the latency-bound tail of one batch overlaps the next batch")
    ap.add_argument("--launch-check", action="store_true",
                    help="no GPU: every rank reports its rank / device / block of the global batch over gloo and rank 0 prints "
                         "one JSON line (the CPU-side test of the N-rank launch)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: the records travel through host tensors -- lets N ranks SHARE one GPU (--share-devices), the "
                         "1-GPU test of the world-size-N path; nccl (= RCCL) is the product path")
    ap.add_argument("--share-devices", action="store_true",
                    help="device = local_rank %% visible devices (several ranks on one GPU; tests only, needs --dist-backend gloo)")
    ap.add_argument("--no-fast-forward", action="store_true", help="skip the secondary `fast_forward` key")
    ap.add_argument("--no-pipeline2", action="store_true", help="skip the secondary `pipeline_2` key (two handles in flight)")
    ap.add_argument("--no-device-loop", action="store_true", help="skip the secondary `device_loop` key (k_sweep_loop against the sweeps, batch 1536)")
    ap.add_argument("--no-numa-bind", action="store_true",
                    help="multi-rank runs bind each rank's host threads to the NUMA node of its GPU; this switches it off")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly: launch the ranks ourselves (the driver's own torch.distributed.run line sets WORLD_SIZE)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE)")
    if args.launch_check:
        return launch_check(args, rank, local_rank, world)

    import numpy as np
    import torch
    import torch.distributed as dist

    A = graft.load_package()
    P = importlib.import_module("altro_cpp_amd.problems")
    S = importlib.import_module("altro_cpp_amd.sharding")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if args.share_devices:
        if args.dist_backend != "gloo":
            raise SystemExit("--share-devices needs --dist-backend gloo (RCCL refuses two ranks on one device)")
        local_rank = local_rank % torch.cuda.device_count()
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible): --gpus N needs N "
                         f"GPUs on this node (tests: --dist-backend gloo --share-devices lets the ranks share one)")
    torch.cuda.set_device(local_rank)
    numa = bind_to_gpu_numa_node(torch, local_rank) if (world > 1 and not args.no_numa_bind) else None

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
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
        world = dist.get_world_size()  # (what the process group really has, not what the environment claims)
    dev = f"cuda:{local_rank}"
    packed = torch.empty((B, 4), dtype=torch.float64, device=dev)
    host_gather = use_dist and args.dist_backend == "gloo"
    gathered = torch.empty((world * B, 4), dtype=torch.float64, device="cpu" if host_gather else dev) if use_dist else packed
    packed_host = torch.empty((B, 4), dtype=torch.float64).pin_memory() if host_gather else None

    def solve(s_):
        solve_once(s_, cfg["mode"])

    def step():
        solve(solver)
        if host_gather:  # (tests: N ranks on one GPU -- the records go through the host)
            solver.pack_results_device(packed.data_ptr())
            packed_host.copy_(packed)
            dist.all_gather_into_tensor(gathered, packed_host)
        else:
            S.pack_and_gather(solver, packed, gathered, dist, force_collective=args.force_dist)  # RCCL over xGMI: 32 B per instance

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
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if args.pipeline > 1:
        drain()
    barrier()
    t0 = time.perf_counter()
    c0 = time.process_time()
    for _ in range(args.steps):
        step()
    if args.pipeline > 1:
        drain()
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0  # this rank's own K steps, before it waits for the others (load imbalance shows here)
    barrier()
    elapsed = time.perf_counter() - t0
    host_cpu_cores = (time.process_time() - c0) / max(elapsed, 1e-9)  # CPU time of ALL threads of this rank per wall second
    rank_elapsed = [own_elapsed]
    cdev = "cpu" if host_gather else dev
    if use_dist:
        mine = torch.tensor([[elapsed, host_cpu_cores, own_elapsed]], dtype=torch.float64, device=cdev)  # (one row per rank)
        allr = torch.empty((world, 3), dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(allr, mine)
        allr = allr.cpu().numpy()
        elapsed = float(allr[:, 0].max())  # MAX over ranks of the barrier-to-barrier time
        rank_elapsed = [float(v) for v in allr[:, 2]]
        host_cpu_cores = float(allr[:, 1].max())

    res = gathered.cpu().numpy()
    # ---- dist_check at EVERY world size: the block of the gathered records that belongs to this rank is this rank's own
    #      altro_get_stats of the last timed solve, field for field; all ranks agree (all_reduce MIN) ----
    dist_check = None
    if use_dist and args.pipeline == 1:
        want = S.result_records(solver.get_stats())
        ok_local = bool(np.array_equal(res[rank * B:(rank + 1) * B], want))
        flag = torch.tensor([1.0 if ok_local else 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        dist_check = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                      "env_world_size": int(os.environ.get("WORLD_SIZE", "1")),
                      "records_match_get_stats": bool(flag.item() == 1.0),
                      "gather_is_separate_buffer": bool(gathered.data_ptr() != packed.data_ptr()),
                      "rank_ms_per_step": [round(1e3 * e / args.steps, 3) for e in rank_elapsed],
                      # (load imbalance is the only scaling loss of this path: a SCALE record diagnoses it at a glance)
                      "rank_ms_min": round(1e3 * min(rank_elapsed) / args.steps, 3),
                      "rank_ms_max": round(1e3 * max(rank_elapsed) / args.steps, 3),
                      "rank_ms_mean": round(1e3 * sum(rank_elapsed) / len(rank_elapsed) / args.steps, 3),
                      "numa": numa}
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
        roofline = dominant_kernel_roofline(cfg, tm, args.config)
        # ---- CPU baseline: the oracle (a port, see oracle/altro_oracle.cpp) on the host cores --------
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            cpu = cpu_baseline(A, P, cfg, B, seed)
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
        if args.force_dist and world == 1 and dist_check is not None:
            # the optional second collective (whole trajectories on every rank), outside the timed region
            Nk, nn, mm = solver.N, solver.n, solver.m
            xp = torch.empty((B, Nk + 1, nn), dtype=torch.float64, device=dev)
            up = torch.empty((B, Nk, mm), dtype=torch.float64, device=dev)
            xg, ug = torch.empty_like(xp), torch.empty_like(up)
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            Xall, Uall = S.pack_and_gather_trajectories(solver, xp, up, xg, ug, dist, force_collective=True)
            torch.cuda.synchronize()
            g_ms = 1e3 * (time.perf_counter() - g0)
            Xh, Uh = solver.get_trajectory()
            dist_check.update({"trajectories_match_get_trajectory": bool(np.array_equal(Xall.cpu().numpy(), Xh) and
                                                                         np.array_equal(Uall.cpu().numpy(), Uh)),
                               "trajectory_gather_ms": round(g_ms, 3),
                               "trajectory_bytes": int(xp.numel() * 8 + up.numel() * 8)})
        others, latency = None, None
        if world == 1 and args.pipeline == 1 and (not args.no_other_configs or not args.no_latency):
            solver.close()  # (the other workloads get the device to themselves, like the headline had it)
            if not args.no_other_configs:
                others = {}
                for ci in sorted(CONFIGS):
                    if ci != args.config:
                        others[f"configs[{ci}]"] = measure_other_config(A, P, S, torch, ci, 3, 1, local_rank)
            if not args.no_latency:
                latency = measure_latency(A, P, local_rank, with_cpu=not args.no_cpu_baseline)
        # ---- SECONDARY, never `value`: the same step with ALTRO_HIP_FAST_FORWARD_STALLS -- the stragglers' bit-identical
        #      repetitions of a rejected line search are counted and logged instead of recomputed (results bit-identical,
        #      tests/test_fused_gpu.py), i.e. work the reference performs is skipped ----
        fast_forward = None
        if (world == 1 and args.pipeline == 1 and cfg["mode"] == "al" and not args.no_fast_forward
                and not os.environ.get("ALTRO_HIP_FAST_FORWARD_STALLS")):
            if others is None and latency is None:
                solver.close()
            os.environ["ALTRO_HIP_FAST_FORWARD_STALLS"] = "1"  # (read when the engine is created)
            try:
                sf = new_solver()
                for _ in range(max(1, args.warmup)):
                    solve(sf)
                    S.pack_and_gather(sf, packed, packed, None)
                torch.cuda.synchronize()
                f0 = time.perf_counter()
                fsteps = max(1, min(args.steps, 5))
                for _ in range(fsteps):
                    solve(sf)
                    S.pack_and_gather(sf, packed, packed, None)
                torch.cuda.synchronize()
                fel = time.perf_counter() - f0
                fres = packed.cpu().numpy()
                fsolved = int((fres[:, 3].astype(int) == 0).sum())
                fast_forward = {"ms_per_step": round(1e3 * fel / fsteps, 3), "value": round(fsolved * fsteps / fel, 1),
                                "unit": "trajectories/s", "steps": fsteps,
                                "records_identical_to_headline": bool(np.array_equal(fres, res[:B])),
                                "note": "NOT the headline: ALTRO_HIP_FAST_FORWARD_STALLS=1 skips recomputing the stragglers' "
                                        "bit-identical rejected iterations (the reference recomputes them)"}
                sf.close()
            finally:
                del os.environ["ALTRO_HIP_FAST_FORWARD_STALLS"]
        # ---- SECONDARY, never `value`: TWO handles in flight (asynchronous solves on their own streams): the latency-bound
        #      tail of one batch -- ~100 stragglers on as many CUs -- overlaps the throughput-bound sweeps of the next.
        #      Every step is still one complete solve of the headline's batch; the records of BOTH handles are compared
        #      with the headline's ----
        pipeline_2 = None
        if world == 1 and args.pipeline == 1 and cfg["mode"] == "al" and not args.no_pipeline2:
            if others is None and latency is None and fast_forward is None:
                solver.close()
            pool = [new_solver(), new_solver()]
            pend = [False, False]
            same = [None, None]

            def pstep(i, check=False):
                s_ = pool[i]
                if pend[i]:
                    s_.wait()
                    s_.pack_results_device(packed.data_ptr())
                    if check:
                        same[i] = bool(np.array_equal(packed.cpu().numpy(), res[:B]))
                s_.reset_trajectory()
                s_.solve_async()
                pend[i] = True

            def pdrain(check=False):
                for i, s_ in enumerate(pool):
                    if pend[i]:
                        s_.wait()
                        s_.pack_results_device(packed.data_ptr())
                        if check:
                            same[i] = bool(np.array_equal(packed.cpu().numpy(), res[:B]))
                        pend[i] = False

            for i in range(2 * max(1, args.warmup)):
                pstep(i % 2)
            pdrain()
            torch.cuda.synchronize()
            psteps = 2 * max(1, min(args.steps, 10) // 2)
            p0 = time.perf_counter()
            for i in range(psteps):
                pstep(i % 2)
            pdrain()
            torch.cuda.synchronize()
            pel = time.perf_counter() - p0
            pstep(0), pstep(1)  # (untimed: one more solve per handle whose records are read back and compared)
            pdrain(check=True)
            pipeline_2 = {"ms_per_step": round(1e3 * pel / psteps, 3), "value": round(solved * psteps / pel, 1),
                          "unit": "trajectories/s", "steps": psteps, "handles_in_flight": 2,
                          "records_identical_to_headline": bool(same[0] and same[1]),
                          "note": "NOT the headline: two solver handles in flight (altro_solve_al_async), each step one complete "
                                  "solve of the headline's batch; the tail of one solve overlaps the sweeps of the next"}
            for s_ in pool:
                s_.close()
        device_loop = None
        if world == 1 and args.pipeline == 1 and args.config == 2 and not args.no_device_loop and not os.environ.get("ALTRO_HIP_SWEEP_LOOP"):
            if others is None and latency is None and fast_forward is None and pipeline_2 is None:
                solver.close()
            try:
                device_loop = measure_device_loop(A, P, S, torch, local_rank)
            except Exception as e:  # (never fail the line for a secondary key)
                device_loop = {"error": str(e)[:300]}
        # ---- the achievable HBM ceiling of this device beside the data-sheet peak ----
        try:
            copy_gbs = measure_hbm_copy_gbs(torch, dev)
            roofline["peak_measured_copy"] = round(copy_gbs, 1)
            roofline["frac_of_measured_copy"] = round(roofline["achieved"] / copy_gbs, 5)
        except Exception as e:  # (never fail the line for the side measurement)
            roofline["peak_measured_copy"] = None
            roofline["peak_measured_copy_error"] = str(e)[:200]
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
                "parallelism": f"instance-sharded x{world} (no data-path collective; "
                               f"{'RCCL' if args.dist_backend == 'nccl' else 'gloo (host)'} all_gather of result records)",
                "host_cpu_cores_per_rank": round(host_cpu_cores, 3),
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
            "other_configs": others,
            "latency": latency,
            "fast_forward": fast_forward,
            "pipeline_2": pipeline_2,
            "device_loop": device_loop,
            **({"dist_check": dist_check} if dist_check is not None else {}),
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
