"""Sharding by env index leaves the results unchanged: two batches of 4 096 envs with env_base 0 and 4 096 are, bit for
bit, the two halves of one batch of 8 192 envs under the same seed -- the worlds an on-device reset draws, the
respawn streams of the scenarios that have one, the time-limit resets of mpe_step_auto and the device's noise
generator are all keyed by the env's index in the whole job (mpe_set_env_base), as the reference seeds env k of a
vectorised run from its global position (multiagent/backup/mappo.py:97,113)."""
import dataclasses

import numpy as np
import pytest

import mpe_oracle as orc

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

HALF = 4096
# one scenario per kernel family (+ the two whose post_step draws respawns from a per-world stream)
CASES = [("simple_spread", {}), ("simple_world_comm", {}), ("simple_predator_prey", {}), ("collector", {}),
         ("simple_spread", dict(n_agents=9, n_landmarks=5)), ("consensus", {})]


def _envs(scenario, kw, dtype=None, **more):
    import mpe_hip
    dt = dtype or torch.float32
    whole = mpe_hip.BatchedEnv(scenario, 2 * HALF, dtype=dt, device="cuda:0", **more, **kw)
    parts = [mpe_hip.BatchedEnv(scenario, HALF, dtype=dt, device="cuda:0", env_base=k * HALF, **more, **kw) for k in range(2)]
    return whole, parts


def _actions(env, T, seed):
    gen = torch.Generator("cuda:0").manual_seed(seed)
    a = torch.randint(0, 5, (T, env.n_envs, env.n_act), dtype=torch.int32, device="cuda:0", generator=gen)
    for col, n in enumerate(env.action_dims):
        if n != 5:
            a[:, :, col] = torch.randint(0, n, (T, env.n_envs), dtype=torch.int32, device="cuda:0", generator=gen)
    return a


def _same_state(whole, parts):
    w = whole.get_state()
    for k, p in enumerate(parts):
        got = p.get_state()
        for x, y in zip(w, got):
            if x is not None:
                assert np.array_equal(x[k * HALF:(k + 1) * HALF], y, equal_nan=True)
        if whole.aux is not None:
            assert torch.equal(whole.aux[:, k * HALF:(k + 1) * HALF], p.aux)


@pytest.mark.parametrize("scenario,kw", CASES)
def test_episode_launch_of_two_shards_is_the_unsharded_launch(scenario, kw):
    whole, parts = _envs(scenario, kw)
    T = 25
    a = _actions(whole, T, 3)
    out = whole.rollout(a, reset_seed=77, obs0=True)
    for k, p in enumerate(parts):
        sl = slice(k * HALF, (k + 1) * HALF)
        o = p.rollout(a[:, sl].contiguous(), reset_seed=77, obs0=True)
        for key in ("obs0", "obs", "rew", "done"):
            assert torch.equal(o[key], out[key][:, sl]), (scenario, key, k)
    _same_state(whole, parts)
    # the second shard's worlds are not the first shard's
    assert not torch.equal(out["obs0"][0, :HALF], out["obs0"][0, HALF:])


@pytest.mark.parametrize("scenario,kw", CASES)
def test_reset_and_step_auto_across_a_time_limit(scenario, kw):
    H, T = 5, 12
    whole, parts = _envs(scenario, kw, max_episode_steps=H, seed=31)
    a = _actions(whole, T, 4)
    first = whole.reset(seed=31).clone()
    for k, p in enumerate(parts):
        assert torch.equal(p.reset(seed=31), first[k * HALF:(k + 1) * HALF]), (scenario, k)
    for t in range(T):
        obs, rew, done, _ = whole.step(a[t])
        for k, p in enumerate(parts):
            sl = slice(k * HALF, (k + 1) * HALF)
            o, r, d, _ = p.step(a[t, sl].contiguous())
            assert torch.equal(o, obs[sl]) and torch.equal(r, rew[sl]) and torch.equal(d, done[sl]), (scenario, t, k)
        assert bool(done.all()) == ((t + 1) % H == 0)
    _same_state(whole, parts)
    for k, p in enumerate(parts):
        sl = slice(k * HALF, (k + 1) * HALF)
        assert torch.equal(p.last_return, whole.last_return[sl]) and torch.equal(p.episode_count, whole.episode_count[sl])


@pytest.mark.parametrize("scenario,kw,levels", [("simple_tag", {}, [0.4, 0.0, 0.2, 0.7]),
                                                  ("simple_spread", dict(n_agents=9, n_landmarks=5), [0.3] * 9),
                                                  ("consensus", {}, [0.5] * 6)])
def test_device_noise_of_two_shards_is_the_unsharded_stream(scenario, kw, levels):
    import mpe_hip
    spec = dataclasses.replace(mpe_hip.make_spec(scenario, **kw), u_noise=np.asarray(levels, float))
    whole, parts = _envs(spec, {}, dtype=torch.float64)
    a = _actions(whole, 6, 9)
    for e in [whole] + parts:
        e.reset(seed=2, observe=False)
        e.set_noise(seed=55)
    for t in range(2):
        obs, _, _, _ = whole.step(a[t])
        for k, p in enumerate(parts):
            sl = slice(k * HALF, (k + 1) * HALF)
            assert torch.equal(p.step(a[t, sl].contiguous())[0], obs[sl]), (scenario, t, k)
    out = whole.rollout(a[2:])
    for k, p in enumerate(parts):
        sl = slice(k * HALF, (k + 1) * HALF)
        assert torch.equal(p.rollout(a[2:, sl].contiguous())["obs"], out["obs"][:, sl]), (scenario, k)
    _same_state(whole, parts)
    # and the stream is the documented one: world b of the second shard draws as world HALF + b
    if scenario == "simple_tag":
        ospec = orc.make_spec(scenario)
        ospec.u_noise = np.asarray(levels, float)
        nu_whole, _ = orc.device_noise(ospec, 55, 0, 2 * HALF)
        nu_part, _ = orc.device_noise(ospec, 55, 0, HALF, env_base=HALF)
        assert np.array_equal(nu_whole[HALF:], nu_part)


def test_the_oracle_restates_the_offset_reset_stream():
    import mpe_hip
    env = mpe_hip.BatchedEnv("simple_tag", 300, dtype=torch.float64, device="cuda:0", env_base=123456789)
    env.reset(seed=5, observe=False)
    pos, _, _ = env.get_state()
    b = np.arange(300, dtype=np.uint64) + np.uint64(123456789)
    want = np.stack([orc.device_reset_draw(5, b, 0, e, 1.0 if e < 4 else 0.9) for e in range(6)], axis=1)
    assert np.array_equal(pos, want)
    with pytest.raises(mpe_hip.MpeError):
        mpe_hip.BatchedEnv("simple_tag", 8, device="cuda:0", env_base=1 << 50)
