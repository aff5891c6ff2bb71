"""Multi-GPU helpers: worlds shard by env index, one process per GPU, no data-path collective.

The only cross-rank traffic is an optional all-reduce of a few episode statistics (sum of
returns, collisions, occupied landmarks, episode count), issued once per episode outside the step
loop through ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on GPUs, ``gloo`` in CPU
tests).  It is latency-bound (<= 64 bytes), so it stays off the stepping stream's critical path.
"""
import torch
import torch.distributed as dist


def shard_range(global_envs, rank, world_size):
    """Contiguous env-index range [begin, end) of ``rank``: sizes differ by at most one and the
    ranges tile [0, global_envs) in rank order."""
    if not (0 <= rank < world_size) or global_envs < 0:
        raise ValueError("bad shard request: %d envs, rank %d of %d" % (global_envs, rank, world_size))
    base, extra = divmod(global_envs, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class EpisodeStats:
    """Running sums [return of agent 0, collisions, occupied landmarks, episodes] as one float64
    tensor on the rank's device; ``reduce()`` sums it over all ranks."""

    FIELDS = ("return_agent0", "collisions", "occupied", "episodes")

    def __init__(self, device):
        self.sums = torch.zeros(len(self.FIELDS), dtype=torch.float64, device=device)

    def add_episode(self, rew, info_i=None):
        """rew [T, B, A] (or [B, A]); info_i [T, B, A+1] simple_spread info ints (optional)."""
        A = rew.shape[-1]
        self.sums[0] += rew[..., 0].double().sum()
        if info_i is not None:
            self.sums[1] += info_i[..., :A].double().sum()
            self.sums[2] += info_i[-1, :, A].double().sum() if info_i.dim() == 3 else info_i[:, A].double().sum()
        self.sums[3] += rew.shape[-2]

    def reduce(self, group=None):
        """Global sums as a dict (identical on every rank)."""
        total = self.sums.clone()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            if dist.get_backend(group) == "gloo":
                total = total.cpu()         # gloo reduces host memory
            dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        return dict(zip(self.FIELDS, total.tolist()))
