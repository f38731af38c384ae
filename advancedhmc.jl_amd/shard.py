"""Multi-GPU sharding of the chain axis.

Chains are independent — in vectorised mode they do not even share step size or mass matrix — so
the N chains are cut into contiguous blocks, one per rank (= one process per GPU).  Nothing is
exchanged while sampling; the Philox stream of a chain depends on its GLOBAL index only
(`PhiloxRNG(seed, chain_offset=first global chain of the shard)`), so a sharded run reproduces
the single-device run chain for chain.  The only collectives are at the end: a SUM of counters
and an all-gather of per-dimension moments (RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

import numpy as np


def chain_shard(n_total: int, rank: int, world: int):
    """Contiguous block of chains owned by `rank`: (offset, count); the first n_total % world
    ranks own one extra chain."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    base, extra = divmod(int(n_total), int(world))
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def pooled_moments(sum_theta: np.ndarray, sumsq_theta: np.ndarray, n_draws: int):
    """Per-dimension (Σθ, Σθ², n) of one shard from the engine's per-chain accumulators"""
    return np.stack([sum_theta.sum(axis=1), sumsq_theta.sum(axis=1)]), int(n_draws)


def gather_moments(dist, local: "torch.Tensor", n_local: int, device):
    """All-gather the (2, D) moment sums of every rank and pool them.  Returns (mean, var, n)."""
    import torch

    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    n = torch.tensor([float(n_local)], dtype=torch.float64, device=device)
    if world > 1:
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        tot = torch.stack(parts).sum(dim=0)
    else:
        tot = local
    n_tot = float(n.item())
    mean = tot[0] / n_tot
    var = tot[1] / n_tot - mean * mean
    return mean.cpu().numpy(), var.cpu().numpy(), int(n_tot)


class EngineComm:
    """The RCCL communicator of a sharded run, owned by the engine's context: rank 0 makes the unique id
    (ahmc_comm_unique_id), the 128 bytes travel through the process group the launcher already set up
    (torch.distributed — plumbing), every rank joins with ahmc_comm_init; from then on the gathers are C-ABI calls
    (ahmc_gather_moments, ahmc_gather_state) that run ncclAllReduce / ncclAllGather on the context's stream.
    Without a process group the world is one rank and no communicator is made."""

    def __init__(self, engine, dist, device):
        self.engine = engine
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        self.how = "single rank (no collective)"
        if dist is not None and dist.is_initialized():
            import torch

            rank = dist.get_rank()
            uid = torch.zeros(128, dtype=torch.uint8, device=device)
            if rank == 0:
                uid = torch.frombuffer(bytearray(engine.comm_unique_id()), dtype=torch.uint8).to(device)
            dist.broadcast(uid, src=0)
            engine.comm_init(bytes(uid.cpu().numpy().tobytes()), self.world, rank)
            self.how = f"ahmc_gather_moments: ncclAllReduce of 2*D+3 doubles over {self.world} rank(s), communicator made by ahmc_comm_init"

    def gather_moments(self):
        g = self.engine.gather_moments()
        g["how"] = self.how
        return g

    def close(self):
        pass  # the context owns the communicator and destroys it with itself
