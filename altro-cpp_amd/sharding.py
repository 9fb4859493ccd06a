"""Instance sharding across ranks (one process per GPU) and the result gather.

The batch of independent problem instances is the only parallel axis: the
reference has no batch and no distributed code at all, so there is nothing to translate.  Each rank
owns a contiguous block of instances, solves it with its own handle on its own device, and the only
collective is ONE all_gather of the 32-byte per-instance result record {cost, violation,
iterations_total, status} after the solve (RCCL over xGMI on GPUs: backend "nccl"; gloo on CPU for
the tests).  No all-reduce, no per-iteration communication.  A caller that wants the whole solution on
every device asks for the optional second collective, ``pack_and_gather_trajectories``.

``pack_and_gather`` is the one code path of that exchange: bench.py calls it on GPU tensors with RCCL,
tests/test_sharding_gloo.py calls the very same function on CPU tensors with gloo.
"""
import numpy as np

RECORD_FIELDS = ("cost", "violation", "iterations_total", "status")


def shard_range(total, world, rank):
    """Contiguous block split of `total` instances: ranks [0, total % world) get one extra."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def result_records(stats):
    """[B][4] float64 records {cost, violation, iterations_total, status} from a get_stats() array."""
    out = np.empty((len(stats), 4), dtype=np.float64)
    for i, f in enumerate(RECORD_FIELDS):
        out[:, i] = stats[f]
    return out


def pack_and_gather(solver, packed, gathered=None, dist=None, force_collective=False):
    """The result exchange of one step.

    The solver writes its [b][4] fp64 result records straight into ``packed`` -- memory of the rank's
    own device (a torch tensor on cuda:<local_rank>; host memory when the CPU oracle stands in for the
    device in the tests) -- and, with more than one rank, ONE ``all_gather_into_tensor`` assembles
    ``gathered`` = [world * b][4] on every rank.  Equal shard sizes (weak scaling); uneven shards go
    through ``gather_variable``.  Returns the tensor that holds the global records.  ``force_collective`` runs the
    all-gather also in a group of one rank (the 1-GPU test of the RCCL path, tests/test_rccl_world1_gpu.py)."""
    solver.pack_results_device(packed.data_ptr())
    if dist is not None and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_collective):
        dist.all_gather_into_tensor(gathered, packed)
        return gathered
    return packed


def pack_and_gather_trajectories(solver, x_packed, u_packed, x_gathered=None, u_gathered=None, dist=None,
                                 force_collective=False):
    """The optional SECOND collective of SURVEY.md section 8(e): every rank receives the trajectories of all shards.

    The solver writes X[b][N+1][n] and U[b][N][m] (fp64, the layout of ``get_trajectory``) straight into ``x_packed`` /
    ``u_packed`` -- memory of the rank's own device -- and, with more than one rank, two ``all_gather_into_tensor``
    calls assemble [world * b][...] on every rank (C4: 32 768 x 503 doubles = 132 MB per GPU; at one xGMI link of
    ~153 GB/s per neighbour that is ~1 ms in a ring over several links, against 41 ms of solve).  Equal shard sizes.
    Returns the two tensors that hold the global trajectories."""
    solver.pack_trajectory_device(x_packed.data_ptr(), u_packed.data_ptr())
    if dist is not None and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_collective):
        dist.all_gather_into_tensor(x_gathered, x_packed)
        dist.all_gather_into_tensor(u_gathered, u_packed)
        return x_gathered, u_gathered
    return x_packed, u_packed


def gather_variable(local_np, dist):
    """Gather record blocks of different sizes (uneven shards): pad to the max, gather, trim."""
    import torch

    world = dist.get_world_size()
    n = torch.tensor([local_np.shape[0]], dtype=torch.int64)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = np.zeros((mx, local_np.shape[1]), dtype=np.float64)
    pad[: local_np.shape[0]] = local_np
    t = torch.from_numpy(pad)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.concatenate([o.numpy()[:s] for o, s in zip(out, sizes)], axis=0)
