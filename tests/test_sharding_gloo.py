"""Multi-process path (world_size 2, gloo, CPU): instance sharding + the result-record all_gather.

Each rank solves its shard of the seeded config-3 batch with the CPU oracle (tests may use it; the
product needs a GPU) and the gathered records must equal a single-process solve of the whole batch
bit for bit -- the template is the reference's serial-vs-threaded equality test
(test/examples/example_unicycle_test.py:155-166)."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes, importlib, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
S = importlib.import_module("altro_cpp_amd.sharding")
lib = ctypes.CDLL(os.path.join(sys.argv[1], "oracle", "_build", "liboracle.so"))
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
total = int(sys.argv[2])
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d, _lib=lib, _prefix="oracle_")
# global seeded batch; each rank builds only its block -- exactly what bench.py does
shard = S.shard_range(total, world, rank)
s = P.batch_turn90(make, batch=total, shard=shard)
s.solve()
if total % world == 0:
    # equal shards: the result exchange of bench.py's step(), same function, gloo instead of RCCL and host
    # memory instead of HBM (the oracle's pack_results_device writes to a host pointer)
    b = shard[1] - shard[0]
    packed = torch.empty((b, 4), dtype=torch.float64)
    gathered = torch.empty((world * b, 4), dtype=torch.float64)
    allrec = S.pack_and_gather(s, packed, gathered, dist).numpy()
    assert np.array_equal(packed.numpy(), S.result_records(s.get_stats()))
    # the optional second collective: whole trajectories on every rank
    xp, up = torch.empty((b, 101, 3), dtype=torch.float64), torch.empty((b, 100, 2), dtype=torch.float64)
    xg, ug = torch.empty((world * b, 101, 3), dtype=torch.float64), torch.empty((world * b, 100, 2), dtype=torch.float64)
    Xall, Uall = S.pack_and_gather_trajectories(s, xp, up, xg, ug, dist)
    Xl, Ul = s.get_trajectory()
    assert np.array_equal(xp.numpy(), Xl) and np.array_equal(up.numpy(), Ul)
    if rank == 0:
        np.savez(sys.argv[3] + ".traj.npz", X=Xall.numpy(), U=Uall.numpy())
else:
    allrec = S.gather_variable(S.result_records(s.get_stats()), dist)
if rank == 0:
    np.save(sys.argv[3], allrec)
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


import pytest


@pytest.mark.parametrize("total", [12, 13])  # 12: bench.py's pack_and_gather path; 13: uneven shards
def test_two_rank_gloo_matches_single_process(tmp_path, A, P, oracle_make, total):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "gathered.npy"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(script), ROOT, str(total), str(out)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    gathered = np.load(out)
    import importlib
    S = importlib.import_module("altro_cpp_amd.sharding")
    s = P.batch_turn90(oracle_make, batch=total)
    s.solve()
    ref = S.result_records(s.get_stats())
    assert gathered.shape == ref.shape
    assert (gathered == ref).all()  # bitwise: instances are independent of how they are sharded
    if total % 2 == 0:  # the trajectory all-gather of the equal-shard path
        t = np.load(str(out) + ".traj.npz")
        Xr, Ur = s.get_trajectory()
        assert np.array_equal(t["X"], Xr) and np.array_equal(t["U"], Ur)


def test_shard_range_partitions(A):
    import importlib
    S = importlib.import_module("altro_cpp_amd.sharding")
    for total in (1, 7, 8, 4096, 32768):
        for world in (1, 2, 3, 8):
            spans = [S.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
