"""The N > 1 path on CPU: env-index sharding and the statistics all-reduce over gloo, 2 ranks."""
import os
import socket

import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def test_shard_ranges_tile_the_batch():
    from mpe_hip.dist import shard_range
    for total in (0, 1, 7, 65536, 1048576 + 3):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mpe_hip.dist import EpisodeStats, shard_range
    b, e = shard_range(10, rank, world)
    stats = EpisodeStats("cpu")
    T, A = 3, 3
    # rewards / info derived from the GLOBAL env index so the expected global sums are known
    idx = torch.arange(b, e, dtype=torch.float64)
    rew = -idx.view(1, -1, 1).expand(T, e - b, A).contiguous()
    info = torch.ones(T, e - b, A + 1, dtype=torch.int32)
    stats.add_episode(rew, info)
    total = stats.reduce()
    out.put((rank, total))
    dist.barrier()
    dist.destroy_process_group()


def test_stats_allreduce_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = {"return_agent0": -3.0 * sum(range(10)), "collisions": 3.0 * 10 * 3, "occupied": 10.0, "episodes": 10.0}
    assert results[0] == want and results[1] == want
