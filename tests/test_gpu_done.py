"""The batch's done_callback (mpe_set_done_bound: an agent beyond the arena bound; MultiAgentEnv combines the agents'
values with any(done_n), multiagent/environment.py:98,110-113,143-146) against the reference's own step() driven with
that callback (tests/golden/done.npz), and mpe_step_auto ending episodes on it, in all three kernel families."""
import numpy as np
import pytest

import mpe_oracle as orc
from conftest import load_golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _env(scenario, B, dtype=torch.float64, **kw):
    import mpe_hip
    return mpe_hip.BatchedEnv(scenario, B, dtype=dtype, device="cuda:0", **kw)


@pytest.mark.parametrize("copies", [1, 4])          # 32 envs: the guarded kernels; 128 envs: the full-tile ones
def test_done_flags_are_the_references(copies):
    g = load_golden("done")
    T, B0, A = g["actions"].shape
    B = B0 * copies
    rep = lambda x, axis=0: np.concatenate([x] * copies, axis=axis)
    bound = float(g["bound"])
    step_env, roll_env, epi_env = (_env("simple_tag", B, done_bound=bound) for _ in range(3))
    plain = _env("simple_tag", B)
    for e in (step_env, roll_env, plain):
        e.set_state(rep(g["pos0"]), rep(g["vel0"]))
    actions = torch.as_tensor(rep(g["actions"], 1)).to("cuda:0")
    want = rep(g["done"], 1)
    assert want.any() and not want.all()
    for t in range(T):
        _, _, done, _ = step_env.step(actions[t])
        assert np.array_equal(done.cpu().numpy(), want[t]), t
        assert np.allclose(step_env.get_state()[0][:, :A], rep(g["pos"][t]), rtol=1e-9, atol=1e-12), t
        assert not plain.step(actions[t])[2].any()                    # without a bound: no done callback, never done
    out = roll_env.rollout(actions)
    assert np.array_equal(out["done"].cpu().numpy(), want)
    assert torch.equal(roll_env.pos, step_env.pos)                    # the flag changes nothing else
    # an episode launch draws its own worlds: its flags against the oracle's restatement of the callback
    out = epi_env.rollout(actions, reset_seed=5)
    check = _env("simple_tag", B)
    check.reset(seed=5, observe=False)
    spec = orc.make_spec("simple_tag")
    for t in range(T):
        check.step(actions[t])
        assert np.array_equal(out["done"][t].cpu().numpy().astype(bool), orc.done_outside(spec, check.get_state()[0], bound)), t


def _random_actions(env, gen):
    a = torch.randint(0, 5, (env.n_envs, env.n_act), dtype=torch.int32, device="cuda:0", generator=gen)
    for col, n in enumerate(env.action_dims):
        if n != 5:
            a[:, col] = torch.randint(0, n, (env.n_envs,), dtype=torch.int32, device="cuda:0", generator=gen)
    return a


@pytest.mark.parametrize("scenario,kw", [("simple_tag", {}), ("simple_world_comm", {}), ("simple_spread", dict(n_agents=9, n_landmarks=5)),
                                         ("consensus", {})])
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_step_auto_ends_episodes_on_done(scenario, kw, dtype):
    """mpe_step_auto with a done bound and a time limit against plain mpe_step with the bookkeeping done here: an env
    whose agent leaves the bound is done, its return is closed and it is reset with the draws of its next episode
    number, exactly as an env that reaches the limit."""
    dt = getattr(torch, dtype)
    B, H, bound = 300, 9, 1.05
    auto = _env(scenario, B, dt, max_episode_steps=H, seed=13, done_bound=bound, **kw)
    ref = _env(scenario, B, dt, done_bound=bound, **kw)
    assert torch.equal(auto.reset(seed=13), ref.reset(seed=13))
    gen = torch.Generator("cuda:0").manual_seed(23)
    count, episode = np.zeros(B, np.int64), np.zeros(B, np.int64)
    ret = torch.zeros(B, auto.n_agents, dtype=dt, device="cuda:0")
    by_bound = by_limit = 0
    for t in range(40):
        a = _random_actions(auto, gen)
        oa, ra, da, _ = auto.step(a)
        orr, rr, dr, _ = ref.step(a)
        assert torch.equal(ra, rr), t
        ret += rr
        count += 1
        outside = dr.cpu().numpy().astype(bool)
        fin = outside | (count >= H)
        by_bound += int((outside & (count < H)).sum())
        by_limit += int((~outside & (count >= H)).sum())
        assert np.array_equal(da.cpu().numpy().astype(bool), fin), t
        if fin.any():
            f = torch.as_tensor(fin, device="cuda:0")
            assert torch.equal(auto.last_return[f], ret[f]), t
            ret[f] = 0
            orr = orr.clone()
            for ep in np.unique(episode[fin] + 1):
                ref._episode = int(ep)
                sel = fin & (episode + 1 == ep)
                fresh = ref.reset(seed=13, mask=torch.as_tensor(sel.astype(np.uint8)))
                orr[torch.as_tensor(sel, device="cuda:0")] = fresh[torch.as_tensor(sel, device="cuda:0")]
            episode[fin] += 1
            count[fin] = 0
        assert torch.equal(oa, orr), t
        assert torch.equal(auto.ep_return, ret), t
        a_state, r_state = auto.get_state(), ref.get_state()
        if scenario == "consensus":             # the rings may differ in where they start, not in what they hold
            a_state, r_state = a_state[:2] + auto.get_history(), r_state[:2] + ref.get_history()
        for x, y in zip(a_state, r_state):
            assert (x is None and y is None) or np.array_equal(x, y), t
        assert np.array_equal(auto.step_count.cpu().numpy(), count) and np.array_equal(auto.episode_count.cpu().numpy(), episode), t
    assert by_bound > 20 and by_limit > 20, (by_bound, by_limit)


def test_the_bound_can_be_switched_off_and_is_part_of_the_checkpoint():
    import mpe_hip
    env = _env("simple_tag", 64, done_bound=float("inf"))            # +inf: no done callback
    env.reset(seed=1, observe=False)
    env.pos.mul_(50.0)
    assert not env.step(torch.zeros(64, 4, dtype=torch.int32))[2].any()
    with pytest.raises(ValueError, match="fingerprint"):
        _env("simple_tag", 64, done_bound=1.0).load_state_dict(env.state_dict())
    with pytest.raises(mpe_hip.MpeError):
        mpe_hip.BatchedEnv(mpe_hip.physics_spec(2, 0, 2, [0.1, 0.1], [1, 1], [1, 1], [5, 5], [np.inf, np.inf], [True, True]), 8,
                           device="cuda:0", done_bound=1.0)
