"""Instance sharding across ranks (one process per GPU) and the result gather.

The batch of independent problem instances is the only parallel axis: the
reference has no batch and no distributed code at all, so there is nothing to translate.  Each rank
owns a contiguous block of instances, solves it with its own handle on its own device, and the only
collective is ONE all_gather of the 32-byte per-instance result record {cost, violation,
iterations_total, status} after the solve (RCCL over xGMI on GPUs: backend "nccl"; gloo on CPU for
the tests).  No all-reduce, no per-iteration communication.
"""
import numpy as np


def shard_range(total, world, rank):
    """Contiguous block split of `total` instances: ranks [0, total % world) get one extra."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def result_records(stats):
    """[B][4] float64 records {cost, violation, iterations_total, status} from a get_stats() array."""
    out = np.empty((len(stats), 4), dtype=np.float64)
    out[:, 0] = stats["cost"]
    out[:, 1] = stats["violation"]
    out[:, 2] = stats["iterations_total"]
    out[:, 3] = stats["status"]
    return out


def gather_records(local, dist=None, device=None):
    """all_gather of equally-sized [b][4] record blocks -> [world*b][4] (torch tensors).

    `local` is a torch tensor on the communication device (cuda for nccl/RCCL, cpu for gloo)."""
    import torch

    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    out = torch.empty((world * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


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
