"""Trainer-side glue on the GPU: the advantage scan against the fixture captured from the reference's
own Runner.run (bit-exact: float32 in, float64 scan, float32 out) and the vectorised-env adapter."""
import numpy as np
import pytest

import mpe_oracle as orc
from conftest import load_golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _t(x, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(x), device="cuda:0") if dtype is None else torch.as_tensor(np.ascontiguousarray(x), device="cuda:0").to(dtype)


def test_gae_reference_mode_matches_the_reference_runner_bit_for_bit():
    import mpe_hip
    g = load_golden("gae")
    adv, ret = mpe_hip.gae(_t(g["rewards"]), _t(g["values"]), _t(g["dones"], torch.uint8), _t(g["last_values"]), _t(g["last_dones"], torch.uint8),
                           gamma=float(g["gamma"]), lam=float(g["lam"]), reference=True)
    assert np.array_equal(adv.cpu().numpy(), g["advs"]) and np.array_equal(ret.cpu().numpy(), g["returns"])


@pytest.mark.parametrize("A", [1, 3, 7, 12])
def test_gae_both_modes_against_the_oracle(A):
    import mpe_hip
    rng = np.random.RandomState(A)
    T, B = 33, 1000
    r = rng.uniform(-2, 1, (T, B, A)).astype(np.float32)
    v = rng.uniform(-3, 3, (T, B, A)).astype(np.float32)
    lv = rng.uniform(-3, 3, (B, A)).astype(np.float32)
    d = rng.uniform(size=(T, B)) < 0.1
    ld = rng.uniform(size=B) < 0.5
    a_ref, r_ref = orc.gae_reference(r, v, d, lv, ld, 0.97, 0.9)
    adv, ret = mpe_hip.gae(_t(r), _t(v), _t(d, torch.uint8), _t(lv), _t(ld, torch.uint8), gamma=0.97, lam=0.9, reference=True)
    assert np.array_equal(adv.cpu().numpy(), a_ref) and np.array_equal(ret.cpu().numpy(), r_ref)
    a_std, r_std = orc.gae_standard(r, v, d, lv, 0.97, 0.9)
    adv, ret = mpe_hip.gae(_t(r), _t(v), _t(d, torch.uint8), _t(lv), gamma=0.97, lam=0.9)
    assert np.array_equal(adv.cpu().numpy(), a_std) and np.array_equal(ret.cpu().numpy(), r_std)


def test_gae_refuses_bad_arguments():
    import mpe_hip
    r = torch.zeros(4, 8, 3, device="cuda:0")
    with pytest.raises(ValueError):
        mpe_hip.gae(r.double(), r.double(), torch.zeros(4, 8, device="cuda:0"), torch.zeros(8, 3, device="cuda:0").double())
    with pytest.raises(ValueError):
        mpe_hip.gae(r, r, torch.zeros(5, 8, device="cuda:0"), torch.zeros(8, 3, device="cuda:0"))
    with pytest.raises(mpe_hip.MpeError):
        mpe_hip.gae(r, r.clone(), torch.zeros(4, 8, device="cuda:0"), torch.zeros(8, 3, device="cuda:0"), out=(r, torch.empty_like(r)))


def test_vec_env_rollout_feeds_the_advantage_scan():
    """The runner's loop (runner_old.py:44-56) on the adapter: nsteps of step(), dones at the time limit with
    the finished returns reported, then advantages over the collected batch."""
    import mpe_hip
    B, T, H = 512, 20, 8
    venv = mpe_hip.BatchedVecEnv("simple_spread", B, max_episode_steps=H, seed=3)
    assert venv.num_envs == B and [s.n for s in venv.action_space] == [5, 5, 5] and [s.shape for s in venv.observation_space] == [(14,)] * 3
    obs = venv.reset()
    assert tuple(obs.shape) == (B, 42) and tuple(venv.obs_n[1].shape) == (B, 14)
    rews, dones, rets = [], [], torch.zeros(B, 3, device="cuda:0")
    gen = torch.Generator("cuda:0").manual_seed(0)
    for t in range(T):
        a = torch.randint(0, 5, (B, 3), dtype=torch.int32, device="cuda:0", generator=gen)
        obs, rew, done, infos = venv.step(a)
        rews.append(rew.clone())
        dones.append(done.clone())
        rets += rew
        if (t + 1) % H == 0:
            assert done.all() and torch.allclose(infos["last_return"], rets, rtol=1e-5, atol=1e-4)
            rets.zero_()
        else:
            assert not done.any()
    rewards = torch.stack(rews)
    values = torch.zeros_like(rewards)
    adv, ret = mpe_hip.gae(rewards, values, torch.stack(dones), torch.zeros(B, 3, device="cuda:0"), gamma=1.0, lam=1.0)
    # with zero values, gamma = lam = 1: the advantage is the reward-to-go within the episode
    want = torch.zeros_like(rewards)
    acc = torch.zeros(B, 3, device="cuda:0", dtype=torch.float64)
    for t in reversed(range(T)):
        acc = rewards[t].double() + acc * (~dones[t]).double()[:, None]
        want[t] = acc.float()
    assert torch.allclose(adv, want, rtol=1e-6, atol=1e-5) and torch.equal(adv, ret)
