"""Episode launches at ragged and odd batch sizes against mpe_reset + single steps, scenario by scenario: the
full-tile instantiation, the guarded one for the tail, and the all-guarded case (a batch that is no multiple of 4)
must produce the same bits as stepping (same kernels' arithmetic, different launch shapes, action fetches and
flush paths)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multiagent-particle-envs_amd"))

pytestmark = pytest.mark.gpu

SCENARIOS = ["simple_spread", "simple_tag", "simple_world_comm", "simple_reference", "simple_crypto", "simple_speaker_listener", "collector"]


@pytest.mark.parametrize("B", [3, 65, 132, 196])
@pytest.mark.parametrize("scenario", SCENARIOS)
def test_episode_launch_equals_reset_and_steps(scenario, B):
    import mpe_hip
    T = 25          # a whole episode: simple_tag / simple_reference fetch all of its actions at once, the others refill
    e1 = mpe_hip.BatchedEnv(scenario, B, dtype=torch.float32, device="cuda:0", info=False)
    e2 = mpe_hip.BatchedEnv(scenario, B, dtype=torch.float32, device="cuda:0", info=False)
    n_act = e1.layout.n_act
    a = torch.randint(0, 5, (T, B, n_act), dtype=torch.int32, device="cuda:0", generator=torch.Generator("cuda:0").manual_seed(B + n_act))
    obs0 = e1.reset(seed=13)
    out = e2.rollout(a, reset_seed=13, obs0=True)
    assert torch.equal(out["obs0"].reshape(obs0.shape), obs0)
    for t in range(T):
        obs, rew, done, _ = e1.step(a[t])
        assert torch.equal(obs, out["obs"][t]), (t, "obs")
        assert torch.equal(rew, out["rew"][t]), (t, "rew")
        assert torch.equal(done, out["done"][t]), (t, "done")
    for k in ("pos", "vel", "comm", "aux"):
        x, y = getattr(e1, k, None), getattr(e2, k, None)
        if x is not None:
            assert torch.equal(x, y), k


LONG = ["simple_spread", "simple_tag", "simple_world_comm", "simple_reference", "simple_predator_prey", "collector", "consensus", "spread_9_4"]


@pytest.mark.parametrize("T", [1, 29, 71])
@pytest.mark.parametrize("scenario", LONG)
def test_long_and_short_rollouts_equal_stepping(scenario, T):
    """Launches of 1, 29 and 71 steps: shorter than, just beyond and several times the actions a lane holds between two
    fetches (28 steps of simple_spread, 33 of simple_tag, 6 of simple_world_comm and simple_predator_prey, 32 of
    consensus, every step of the wave-per-world kernel), ending in the middle of a fetch, on a batch with a ragged
    tail.  State, every step's outputs and (consensus) the history ring must be the bits single steps give."""
    import mpe_hip
    B = 200
    kw = dict(n_agents=9, n_landmarks=4) if scenario == "spread_9_4" else {}
    name = "simple_spread" if scenario == "spread_9_4" else scenario
    for dtype in (torch.float32, torch.float64):
        e1 = mpe_hip.BatchedEnv(name, B, dtype=dtype, device="cuda:0", info=False, **kw)
        e2 = mpe_hip.BatchedEnv(name, B, dtype=dtype, device="cuda:0", info=False, **kw)
        gen = torch.Generator("cuda:0").manual_seed(100 * T + B)
        a = torch.randint(0, 5, (T, B, e1.n_act), dtype=torch.int32, device="cuda:0", generator=gen)
        for col, n in enumerate(e1.action_dims):
            if n != 5:
                a[:, :, col] = torch.randint(0, n, (T, B), dtype=torch.int32, device="cuda:0", generator=gen)
        e1.reset(seed=5)
        e2.reset(seed=5)
        out = e2.rollout(a)
        for t in range(T):
            obs, rew, done, _ = e1.step(a[t])
            assert torch.equal(obs, out["obs"][t]), (scenario, dtype, t, "obs")
            assert torch.equal(rew, out["rew"][t]), (scenario, dtype, t, "rew")
            assert torch.equal(done, out["done"][t]), (scenario, dtype, t, "done")
        for k in ("pos", "vel", "aux"):
            x, y = getattr(e1, k, None), getattr(e2, k, None)
            if x is not None:
                assert torch.equal(x, y), (scenario, dtype, k)
        if e1.comm is not None:
            if scenario == "consensus":
                for x, y in zip(e1.get_history(), e2.get_history()):
                    assert (x == y).all(), (scenario, dtype, "history")
            else:
                assert torch.equal(e1.comm, e2.comm), (scenario, dtype, "comm")


@pytest.mark.parametrize("scenario,B", [("simple_spread", 8192 * 64 + 3 * 64 + 5), ("simple_tag", 8192 * 64 + 64)])
def test_launch_of_many_rounds_deals_tiles_unevenly_to_the_xcds(scenario, B):
    """From 8 192 tiles on, an episode launch gives the even XCDs 9 tiles for every 7 of the odd ones (workgroup index ->
    tile is then a permutation, with idle workgroups at the end of every share): every env must be stepped exactly once,
    the done flags' shared 128-byte lines included, with an odd number of tiles and a ragged tail."""
    import mpe_hip
    T = 3
    e1 = mpe_hip.BatchedEnv(scenario, B, dtype=torch.float32, device="cuda:0", info=False)
    e2 = mpe_hip.BatchedEnv(scenario, B, dtype=torch.float32, device="cuda:0", info=False)
    a = torch.randint(0, 5, (T, B, e1.layout.n_act), dtype=torch.int32, device="cuda:0", generator=torch.Generator("cuda:0").manual_seed(B))
    obs0 = e1.reset(seed=17)
    out = e2.rollout(a, reset_seed=17, obs0=True)
    out["done"].fill_(7)
    out = e2.rollout(a, out=out)      # a second launch from the state the first one left, into flags that are not zero
    assert not bool(out["done"].any())
    e3 = mpe_hip.BatchedEnv(scenario, B, dtype=torch.float32, device="cuda:0", info=False)
    first = e3.rollout(a, reset_seed=17, obs0=True)
    assert torch.equal(first["obs0"].reshape(obs0.shape), obs0)
    for t in range(T):
        obs, rew, done, _ = e1.step(a[t])
        assert torch.equal(obs, first["obs"][t]) and torch.equal(rew, first["rew"][t]) and torch.equal(done, first["done"][t]), t
    for t in range(T):
        obs, rew, done, _ = e1.step(a[t])
        assert torch.equal(obs, out["obs"][t]) and torch.equal(rew, out["rew"][t]), t
    assert torch.equal(e1.pos, e2.pos) and torch.equal(e1.vel, e2.vel)
