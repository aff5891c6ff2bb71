"""bench.py's N > 1 path on the card: two rank processes sharing GPU 0 (gloo between them, since RCCL wants one
device per rank), started by bench.py itself and by torch.distributed.run the way the driver starts them.  Each rank
steps its own shard of the env index; rank 0 prints the one line."""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--gpus", "2", "--share-device", "--steps", "20", "--warmup", "5", "--envs", "4096", "--target-steps", "500",
        "--no-cpu-baseline", "--no-step-mode"]


def _run(cmd):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _check(d):
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["envs_per_gpu"] == 4096 and d["config"]["global_envs"] == 8192
    assert d["steps"] == 20 and d["warmup"] == 5 and d["timed_steps"] == 500 and d["value"] > 0
    # whole-job rate: both ranks' envs over the slower rank's time
    assert abs(d["value"] - 8192 * 3 * d["timed_steps"] / d["elapsed_s"]) <= 1e-6 * d["value"]
    # what torch.distributed saw: two ranks, and ONE seed with each rank's shard keyed by its place in the env index
    assert d["ranks"]["world_size"] == 2 and d["ranks"]["backend"] == "gloo"
    assert [(r["rank"], r["env_base"], r["envs"]) for r in d["ranks"]["ranks"]] == [(0, 0, 4096), (1, 4096, 4096)]
    assert len({r["seed"] for r in d["ranks"]["ranks"]}) == 1 and d["config"]["reset_seed"] == d["ranks"]["ranks"][0]["seed"]
    assert d["configs"] is None and d["sustained"] is None      # one-GPU records
    assert d["cpu_baseline"] is None            # a one-GPU figure
    assert d["mean_episode_return_agent0"] < 0  # the statistics all-reduce ran over both shards


def test_self_launched_ranks_share_the_card():
    _check(_run([sys.executable, "bench.py"] + ARGS))


def test_ranks_under_torch_distributed_run():
    _check(_run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", "29617", "bench.py"] + ARGS))
