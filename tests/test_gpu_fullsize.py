"""Full BASELINE.json batch sizes on the GPU, checked through size-independent properties:

* a sample of the big batch, replayed through the numpy oracle from the same initial states and
  actions, matches (float64 within 1e-9 relative, float32 within 1e-4) after a 25-step episode;
* batch-composition independence: the first envs of the big batch behave bit-identically in a small
  batch fed the same states and actions (env-index sharding across GPUs relies on this);
* one-launch episodes equal step-by-step launches; landmarks never move; done stays 0;
  simple_spread rewards are never positive and collision counts stay within [0, A-1].
"""
import numpy as np
import pytest

import mpe_oracle as orc
from conftest import within_fp32_bar

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

CASES = [  # scenario, kwargs, envs per GPU at the BASELINE.json configuration
    ("simple_spread", {}, 65536),
    ("simple_tag", {}, 131072),            # 1 M envs over 8 GPUs
    ("simple_world_comm", {}, 262144),
    ("simple_spread", dict(n_agents=64, n_landmarks=64), 2048),   # 16 384 envs over 8 GPUs
]
T = 25


def _actions(env, scn, gen):
    a = torch.randint(0, 5, (T, env.n_envs, env.n_act), generator=gen, device="cuda:0", dtype=torch.int32)
    if scn == "simple_world_comm":
        a[:, :, 1] = torch.randint(0, 4, (T, env.n_envs), generator=gen, device="cuda:0", dtype=torch.int32)
    return a


def _close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b) <= atol + rtol * np.abs(b)


@pytest.mark.parametrize("scn,kw,B", CASES)
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_full_batch_episode(scn, kw, B, dtype):
    import mpe_hip
    dt = getattr(torch, dtype)
    big = mpe_hip.BatchedEnv(scn, B, dtype=dt, info=True, **kw)
    A, L = big.n_agents, big.n_landmarks
    spec = orc.make_spec(scn, **({"A": A, "L": L} if scn == "simple_spread" else {}))
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(5)
    actions = _actions(big, scn, gen)
    big.reset(seed=11, observe=False)
    pos0, vel0, _ = big.get_state()
    out = big.rollout(actions)
    pos1, vel1, comm1 = big.get_state()

    # invariants over the whole batch
    assert not out["done"].any()
    assert np.array_equal(pos1[:, A:], pos0[:, A:])                      # landmarks are immovable
    assert np.abs(pos0[:, :A]).max() <= 1.0 and np.abs(vel0).max() == 0
    if scn == "simple_spread":
        assert float(out["rew"].max()) <= 0.0
        coll = out["info_i"][..., :A]
        assert int(coll.min()) >= 0 and int(coll.max()) <= A - 1
    assert torch.isfinite(out["obs"][-1]).all() and torch.isfinite(out["rew"]).all()

    # a sample against the oracle, free-running for the whole episode
    n = 256 if A <= 6 else 4
    idx = np.linspace(0, B - 1, n).astype(np.int64)
    p, v = pos0[idx].astype(np.float64), vel0[idx].astype(np.float64)
    act = actions[:, torch.as_tensor(idx, device="cuda:0")].cpu().numpy()
    for t in range(T):
        p, v, c, o_obs, o_rew, _ = orc.env_step(spec, p, v, act[t])
    if dtype == "float64":
        assert _close(pos1[idx], p, 1e-9, 1e-12).all() and _close(vel1[idx], v, 1e-9, 1e-12).all()
        tol = 1e-6 if scn == "simple_spread" else 1e-9                   # spread obs are float32-rounded
        assert _close(out["obs"][-1].cpu().numpy()[idx], np.concatenate(o_obs, 1), tol, 1e-12).all()
        assert _close(out["rew"][-1].cpu().numpy()[idx], o_rew, 1e-9, 1e-12).all()
    elif A <= 6:
        assert within_fp32_bar(pos1[idx], p, "full size %s:pos" % scn) and within_fp32_bar(vel1[idx], v, "full size %s:vel" % scn)
    else:
        # 64 overlapping agents: float32 trajectories decorrelate within an episode (chaotic contact
        # dynamics), so float32 is held to the oracle over the first step only
        small = mpe_hip.BatchedEnv(scn, n, dtype=dt, **kw)
        small.set_state(pos0[idx], vel0[idx])
        small.step(actions[0][torch.as_tensor(idx, device="cuda:0")].contiguous())
        p1, v1, *_ = orc.env_step(spec, pos0[idx].astype(np.float64), vel0[idx].astype(np.float64), act[0])
        sp, sv, _ = small.get_state()
        assert within_fp32_bar(sp, p1, "full size spread64 single step:pos") and within_fp32_bar(sv, v1, "full size spread64 single step:vel")

    # batch-composition independence + episode launch == step launches, bit for bit
    m = 1000 if A <= 6 else 37
    small = mpe_hip.BatchedEnv(scn, m, dtype=dt, info=True, **kw)
    small.set_state(pos0[:m], vel0[:m])
    for t in range(T):
        obs, rew, done, (ii, ff) = small.step(actions[t, :m].contiguous())
    assert torch.equal(obs, out["obs"][-1, :m]) and torch.equal(rew, out["rew"][-1, :m])
    assert torch.equal(ii, out["info_i"][-1, :m])
    sp, sv, sc = small.get_state()
    assert np.array_equal(sp, pos1[:m]) and np.array_equal(sv, vel1[:m])
    if sc is not None:
        assert np.array_equal(sc, comm1[:m])


@pytest.mark.parametrize("scn,kw,B", CASES)
def test_benchmark_launch_at_full_size_equals_its_parts(scn, kw, B):
    """What bench.py times -- one launch per episode: on-device reset, the reset state's observation, 25 steps --
    against mpe_reset + mpe_observe + mpe_rollout on a second batch, bit for bit, at the BASELINE batch sizes."""
    import mpe_hip
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(9)
    e1 = mpe_hip.BatchedEnv(scn, B, dtype=torch.float32, **kw)
    actions = _actions(e1, scn, gen)
    first = e1.reset(seed=1234).clone()
    o1 = e1.rollout(actions)
    want = {k: o1[k].clone() for k in ("obs", "rew", "done")}
    pos, vel = e1.pos.clone(), e1.vel.clone()
    del e1, o1
    e2 = mpe_hip.BatchedEnv(scn, B, dtype=torch.float32, **kw)
    o2 = e2.rollout(actions, reset_seed=1234, obs0=True)
    assert torch.equal(o2["obs0"][0], first)
    for k in want:
        assert torch.equal(o2[k], want[k]), k
    assert torch.equal(e2.pos, pos) and torch.equal(e2.vel, vel)


def test_reset_is_reproducible_and_independent_of_batch_size():
    import mpe_hip
    a = mpe_hip.BatchedEnv("simple_tag", 131072, dtype=torch.float32)
    b = mpe_hip.BatchedEnv("simple_tag", 5000, dtype=torch.float32)
    a.reset(seed=3, observe=False)
    b.reset(seed=3, observe=False)
    pa, _, _ = a.get_state()
    pb, _, _ = b.get_state()
    assert np.array_equal(pa[:5000], pb)
    a2 = mpe_hip.BatchedEnv("simple_tag", 131072, dtype=torch.float32)
    a2.reset(seed=3, observe=False)
    assert torch.equal(a.pos, a2.pos)
    a2.reset(seed=4, observe=False)
    assert not torch.equal(a.pos, a2.pos)


@pytest.mark.parametrize("scenario,dtype", [("simple_spread", "float32"), ("simple_spread", "float64"), ("simple_tag", "float32"),
                                            ("simple_world_comm", "float32"), ("simple_reference", "float64"),
                                            ("simple_predator_prey", "float32"), ("collector", "float64"),
                                            ("consensus", "float32"), ("consensus", "float64"),
                                            ("spread5", "float32"), ("spread5", "float64")])
def test_multi_episode_launch_equals_single_episode_launches(scenario, dtype):
    """mpe_episodes(K) == K mpe_episode launches with consecutive episode numbers, bit for bit; B is not a
    multiple of the wave size for half of the cases."""
    import mpe_hip
    dt = getattr(torch, dtype)
    B = 4096 if dtype == "float32" else 1000
    K, T = 3, 25
    kw = dict(info=True)
    if scenario == "spread5":                   # a wave-per-world size of simple_spread
        scenario, kw, B = "simple_spread", dict(info=True, n_agents=5, n_landmarks=5), B // 8
    if scenario == "consensus":                 # no benchmark outputs
        kw = {}
    e1 = mpe_hip.BatchedEnv(scenario, B, dtype=dt, device="cuda:0", **kw)
    e2 = mpe_hip.BatchedEnv(scenario, B, dtype=dt, device="cuda:0", **kw)
    gen = torch.Generator("cuda:0").manual_seed(7)
    a = torch.stack([torch.randint(0, n, (K * T, B), dtype=torch.int32, device="cuda:0", generator=gen) for n in e1.action_dims], dim=2).contiguous()
    e1._episode = e2._episode = 5
    out = e1.rollout(a, reset_seed=13, episodes=K, obs0=True)
    assert e1._episode == 5 + K and out["obs0"].shape[0] == K
    for k in range(K):
        o = e2.rollout(a[k * T:(k + 1) * T].contiguous(), reset_seed=13, obs0=True)
        for key in ("obs", "rew", "done", "info_i", "info_f"):
            if o[key] is not None:
                assert torch.equal(out[key][k * T:(k + 1) * T], o[key]), (scenario, k, key)
        assert torch.equal(out["obs0"][k], o["obs0"][0]), (scenario, k)
        # the reset observation is what mpe_reset + mpe_observe give for that episode
        e3 = mpe_hip.BatchedEnv(scenario, B, dtype=dt, device="cuda:0", **kw)
        e3._episode = 5 + k
        assert torch.equal(e3.reset(seed=13), o["obs0"][0]), (scenario, k)
    assert torch.equal(e1.pos, e2.pos) and torch.equal(e1.vel, e2.vel)
    if e1.aux is not None:
        assert torch.equal(e1.aux, e2.aux)
    if e1.comm is not None:
        assert torch.equal(e1.comm, e2.comm)
