"""libaltro_group.so (include/altro_group.h): the single-process multi-GPU exchange for C / C++ callers -- shard
arithmetic on CPU, the RCCL all-gather with a group of ONE device on the GPU box."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "altro-cpp_amd", "csrc", "libaltro_group.so")
HEADER = os.path.join(ROOT, "include", "altro_group.h")


def _lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.dirname(LIB), "libaltro_group.so"])
    lib = ctypes.CDLL(LIB)
    lib.altro_group_last_error.restype = ctypes.c_char_p
    lib.altro_group_last_error.argtypes = [ctypes.c_void_p]
    lib.altro_group_part_ms.restype = ctypes.c_double
    lib.altro_group_gather_ms.restype = ctypes.c_double
    lib.altro_group_trajectory_gather_ms.restype = ctypes.c_double
    return lib


def test_group_library_exports_its_header():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(altro_group_[a-z0-9_]+)\s*\(", src)))
    lib = _lib()
    assert len(names) >= 12 and not [n for n in names if not hasattr(lib, n)]


def test_shard_range_is_the_block_split_of_the_python_side(A):
    import importlib
    S = importlib.import_module("altro_cpp_amd.sharding")
    lib = _lib()
    lo, hi = ctypes.c_int(), ctypes.c_int()
    for total in (1, 7, 8, 4096, 32768, 32771):
        for parts in (1, 2, 3, 4, 8):
            edges = []
            for part in range(parts):
                lib.altro_group_shard_range(total, parts, part, ctypes.byref(lo), ctypes.byref(hi))
                assert (lo.value, hi.value) == S.shard_range(total, parts, part)
                edges.append((lo.value, hi.value))
            assert edges[0][0] == 0 and edges[-1][1] == total and all(a[1] == b[0] for a, b in zip(edges, edges[1:]))


def test_group_create_without_a_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _lib()
    g = ctypes.c_void_p()
    dev = (ctypes.c_int * 1)(0)
    assert lib.altro_group_create(dev, 1, ctypes.byref(g)) != 0
    assert b"HIP device" in lib.altro_group_last_error(None)


@pytest.mark.gpu
def test_group_of_one_device_gathers_the_solver_statistics(A, P, hip_make):
    """ncclCommInitAll + ncclAllGather really run (one device), beside the solver's own streams: the gathered records are
    the handle's statistics, bit for bit, and the solve through the group equals a plain solve."""
    lib = _lib()
    B = 300
    ref = P.batch_turn90(hip_make, batch=B)
    ref.solve()
    want = np.stack([ref.get_stats()[f].astype(np.float64) for f in ("cost", "violation", "iterations_total", "status")], axis=1)
    s = P.batch_turn90(hip_make, batch=B)
    s.num_constraints()  # device state (and the solver's streams) first, then RCCL's
    g = ctypes.c_void_p()
    dev = (ctypes.c_int * 1)(0)
    assert lib.altro_group_create(dev, 1, ctypes.byref(g)) == 0, lib.altro_group_last_error(None)
    try:
        assert lib.altro_group_size(g) == 1
        assert lib.altro_group_solve_al(g) != 0 and b"no handle attached" in lib.altro_group_last_error(g)
        assert lib.altro_group_attach(g, 0, s._h, B) == 0
        for _ in range(2):  # twice: buffers and communicator are reused
            s.reset_trajectory()
            assert lib.altro_group_solve_al(g) == 0, lib.altro_group_last_error(g)
            out = np.zeros((B, 4))
            assert lib.altro_group_get_results(g, 0, out.ctypes.data_as(ctypes.c_void_p), B) == 0
            assert np.array_equal(out, want)
            assert lib.altro_group_total(g) == B and lib.altro_group_part_ms(g, 0) > 0 and lib.altro_group_gather_ms(g) > 0
        assert lib.altro_group_get_results(g, 0, out.ctypes.data_as(ctypes.c_void_p), B - 1) != 0
        # the optional second collective: whole trajectories (two more ncclAllGather calls), equal to get_trajectory
        X, U = np.zeros((B, 101, 3)), np.zeros((B, 100, 2))
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        assert lib.altro_group_get_trajectories(g, 0, vp(X), vp(U), B) != 0 and b"no trajectories gathered" in lib.altro_group_last_error(g)
        for _ in range(2):
            assert lib.altro_group_gather_trajectories(g, 3, 2, 100) == 0, lib.altro_group_last_error(g)
            assert lib.altro_group_get_trajectories(g, 0, vp(X), vp(U), B) == 0, lib.altro_group_last_error(g)
            Xr, Ur = s.get_trajectory()
            assert np.array_equal(X, Xr) and np.array_equal(U, Ur)
        assert lib.altro_group_trajectory_gather_ms(g) > 0
    finally:
        lib.altro_group_destroy(g)


@pytest.mark.gpu
def test_perf_driver_shards_through_the_facade_group():
    """perf/benchmark_unicycle --gpus 1: altro::BatchGroup (include/altro/group.hpp) around the facade solvers."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "perf"), "benchmark_unicycle"])
    r = subprocess.run([os.path.join(ROOT, "perf", "benchmark_unicycle"), "2", "512", "--gpus", "1"], capture_output=True,
                       text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0
    lines = [l for l in r.stdout.splitlines() if "GPU(s) x 512 instances" in l]
    assert len(lines) == 2 and all("records match the per-solver statistics" in l for l in lines)
    assert "trajectories match the per-solver ones" in r.stdout


@pytest.mark.gpu
def test_group_reads_the_handles_dimensions_and_rejects_mismatches(A, P, hip_make):
    """the gather buffers are sized from what the HANDLE says (altro_get_desc), so a caller-supplied batch or
    (n, m, N) that disagrees is refused instead of letting the pack kernels write past the group's buffers."""
    lib = _lib()
    core = A.load_library()
    s = P.batch_turn90(hip_make, batch=96)
    d = A.Desc()
    assert core.altro_get_desc(s._h, ctypes.byref(d)) == 0
    assert (d.n, d.m, d.N, d.batch, d.device_id) == (3, 2, 100, 96, 0)
    g = ctypes.c_void_p()
    dev = (ctypes.c_int * 1)(0)
    assert lib.altro_group_create(dev, 1, ctypes.byref(g)) == 0, lib.altro_group_last_error(None)
    try:
        assert lib.altro_group_attach(g, 0, s._h, 64) != 0 and b"created with batch 96" in lib.altro_group_last_error(g)
        assert lib.altro_group_attach(g, 0, s._h, 96) == 0
        s.solve()
        assert lib.altro_group_gather(g) == 0, lib.altro_group_last_error(g)
        # trajectories: the dimensions must be the handle's
        assert lib.altro_group_gather_trajectories(g, 3, 2, 50) != 0 and b"has (3, 2, 100)" in lib.altro_group_last_error(g)
        assert lib.altro_group_gather_trajectories(g, 3, 2, 100) == 0, lib.altro_group_last_error(g)
    finally:
        lib.altro_group_destroy(g)


@pytest.mark.gpu
def test_group_solve_does_not_spin_a_core_per_handle(A, P, hip_make):
    """while altro_group_solve_al waits for a 4096-instance solve (8 ms: batched sweeps, then the
    persistent tail) the process -- the caller's polling thread plus the handle's worker thread -- uses well under one
    core: the sweep loop naps once the solve has run 0.5 ms, the tail is awaited on a blocking-sync event, the group
    polls its parts with a sleeping poll."""
    import time
    lib = _lib()
    s = P.batch_turn90(hip_make, batch=4096, seed=P.SEED_BASE + 3)
    s.num_constraints()
    g = ctypes.c_void_p()
    dev = (ctypes.c_int * 1)(0)
    assert lib.altro_group_create(dev, 1, ctypes.byref(g)) == 0, lib.altro_group_last_error(None)
    try:
        assert lib.altro_group_attach(g, 0, s._h, 4096) == 0
        for _ in range(2):  # warm-up
            s.reset_trajectory()
            assert lib.altro_group_solve_al(g) == 0, lib.altro_group_last_error(g)
        w0, c0 = time.perf_counter(), time.process_time()
        reps = 10
        for _ in range(reps):
            s.reset_trajectory()
            assert lib.altro_group_solve_al(g) == 0, lib.altro_group_last_error(g)
        wall, cpu = time.perf_counter() - w0, time.process_time() - c0
        print(f"group solve of 4096 instances: {1e3 * wall / reps:.2f} ms per solve, {cpu / wall:.2f} cores")
        assert cpu / wall < 1.0
        assert s.get_timing()["host_naps"] > 0
    finally:
        lib.altro_group_destroy(g)
