"""bench.py's host logic without a GPU: the steps / warm-up arithmetic for every request the driver may
make, the byte model, and the self-launched N-rank path (two fresh rank processes over gloo)."""
import json
import os
import subprocess
import sys

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("steps", [1, 7, 20, 24, 25, 26, 30, 100, 999, 2500, 25000, 25001, 100000])
@pytest.mark.parametrize("warmup", [0, 5, 25, 250])
def test_timing_plan_is_whole_and_never_empty(steps, warmup):
    p = bench.timing_plan(steps, warmup)
    assert p["steps_requested"] == steps and p["repeats"] >= 1
    assert p["timed_steps"] == p["repeats"] * steps                       # exactly the requested block, repeated
    assert p["timed_steps"] >= min(bench.TARGET_STEPS, max(steps, bench.TARGET_STEPS))
    assert p["full_launches"] * bench.EPISODE + p["tail_steps"] == p["timed_steps"]
    assert p["full_launches"] >= 4 * bench.EVENT_GROUP                       # enough launches for HIP event groups
    if steps < bench.TARGET_STEPS:
        assert p["tail_steps"] == 0                                        # short requests time whole episodes only
        assert p["timed_steps"] < bench.TARGET_STEPS + bench.EPISODE * steps   # and not needlessly many
    else:
        assert p["repeats"] == 1 and p["tail_steps"] == steps % bench.EPISODE
    assert p["warmup_steps"] >= max(warmup, 2 * bench.EPISODE) and p["warmup_steps"] % bench.EPISODE == 0
    assert p["warmup_steps"] < max(warmup, 2 * bench.EPISODE) + bench.EPISODE


def test_timing_plan_refuses_nonsense():
    with pytest.raises(ValueError):
        bench.timing_plan(0, 5)
    with pytest.raises(ValueError):
        bench.timing_plan(20, -1)


def test_the_drivers_own_request():
    # python3 bench.py --gpus 1 --steps 20 --warmup 5
    p = bench.timing_plan(20, 5)
    assert p == {"steps_requested": 20, "repeats": 1250, "timed_steps": 25000, "full_launches": 1000, "tail_steps": 0,
                 "warmup_requested": 5, "warmup_steps": 50}


def test_algorithmic_bytes_match_the_survey_figures():
    # simple_spread 3/3 float32: 313 B per env-step as single steps; an episode launch reads no state
    assert bench.algorithmic_bytes(3, 3, 3, 42, 4, 1) == 313
    assert abs(bench.algorithmic_bytes(3, 3, 3, 42, 4, 25, reads_state=False) - 194.92) < 1e-9
    # ... and with the reset state's observation written once per episode: + 168 / 25
    assert abs(bench.algorithmic_bytes(3, 3, 3, 42, 4, 25, reads_state=False, reset_obs=True) - 201.64) < 1e-9
    assert bench.algorithmic_bytes(4, 2, 4, 62, 4, 1) == 425                 # simple_tag
    assert bench.algorithmic_bytes(64, 64, 64, 64 * 258, 4, 1) == 69121     # simple_spread 64/64
    # consensus: a single step also reads the ten-slot history (240 values) and writes the newest slot (24 values) ...
    assert bench.algorithmic_bytes(6, 0, 6, 480, 4, 1, hist_rows=240) == 2161 + 960 + 96
    # mpe_step_auto: + step counter read and written, running return read and written, episode number read; per
    # 25-step episode the finished return and the episode number written once
    assert abs(bench.algorithmic_bytes(3, 3, 3, 42, 4, 1, auto_period=25) - (313 + 8 + 4 + 24 + 16 / 25)) < 1e-9
    # ... an episode launch reads no state at all
    assert abs(bench.algorithmic_bytes(6, 0, 6, 480, 4, 25, reads_state=False, reset_obs=True, hist_rows=240) - 2049.64) < 1e-9


def test_stored_traffic_is_refused_for_another_build(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "source_hash", lambda: "aaaa")
    (prof / "traffic.json").write_text(json.dumps({"k": {"hbm_bytes_per_launch": 5.0, "source_hash": "aaaa"},
                                                   "old": {"hbm_bytes_per_launch": 7.0, "source_hash": "bbbb"},
                                                   "bare": {"hbm_bytes_per_launch": 9.0}}))
    assert bench.stored_traffic("k") == 5.0
    assert bench.stored_traffic("old") is None and bench.stored_traffic("bare") is None
    assert bench.stored_traffic("missing") is None


def test_stored_valu_is_refused_for_another_build(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "source_hash", lambda: "aaaa")
    fig = {"valu_insts_per_wave": 6500.0, "valu_issue_fraction_of_wave_lifetime": 0.25, "cycles_per_valu_inst_while_issuing": 4.3}
    (prof / "r01_valu.json").write_text(json.dumps({"k": dict(fig, source_hash="bbbb")}))
    assert bench.stored_valu("k") is None
    (prof / "r02_valu_x.json").write_text(json.dumps({"k": dict(fig, source_hash="aaaa")}))
    got = bench.stored_valu("k")
    assert got["insts_per_wave"] == 6500.0 and got["source"] == "profiles/r02_valu_x.json"
    assert bench.stored_valu("missing") is None


def test_source_hash_is_stable():
    assert bench.source_hash() == bench.source_hash() and len(bench.source_hash()) == 16


def _bench(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, timeout=300, env=e)


def test_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: two rank processes over gloo, one line from rank 0."""
    r = _bench("--gpus", "2", "--steps", "20", "--warmup", "5", "--envs", "1000", "--dry", "--backend", "gloo")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                              # rank 1 prints nothing
    out = json.loads(lines[0])
    assert out["dry"] is True and out["n_gpus"] == 2
    assert out["shards"] == [[0, 1000], [1000, 2000]]   # env-index shards tile the global batch in rank order
    assert out["stats"]["episodes"] == 2000.0 and out["stats"]["return_agent0"] == -float(sum(range(2000)))
    assert out["plan"]["timed_steps"] == 25000
    # one seed on every rank; a rank's draws are keyed by its shard's place in the global env index
    assert out["ranks"]["world_size"] == 2 and out["ranks"]["backend"] == "gloo"
    assert [(r["rank"], r["env_base"], r["envs"]) for r in out["ranks"]["ranks"]] == [(0, 0, 1000), (1, 1000, 1000)]
    assert len({r["seed"] for r in out["ranks"]["ranks"]}) == 1


def test_a_failing_rank_ends_the_launch():
    # --envs -1 makes shard_range raise in every rank: the launcher must come back with a non-zero code, not hang
    r = _bench("--gpus", "2", "--dry", "--backend", "gloo", "--envs", "-1")
    assert r.returncode != 0


def test_world_size_mismatch_is_refused():
    r = _bench("--gpus", "2", "--dry", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
