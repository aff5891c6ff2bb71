"""consensus (new_core physics, walls, 10-step history observations, reward shared over six agents)
on the GPU against the reference fixture and the oracle.  Bars as in tests/test_gpu_parity.py."""
import numpy as np
import pytest

import mpe_oracle as orc
from conftest import load_golden, within_fp32_bar

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
CASE = "consensus"


def _close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= atol + rtol * np.abs(b))


def _ok(dtype, got, ref, label, f32_factor=1.0, f64=(1e-9, 1e-12)):
    """float64: within f64 = (rtol, atol) of the reference; float32: the stated bar |got - ref| <= f32_factor * 1e-4 *
    max(1, |ref|), its observed margin recorded under `label`."""
    if dtype == "float64":
        return _close(got, ref, *f64)
    return within_fp32_bar(got, ref, label, f32_factor)


def _env(B, dtype, **kw):
    import mpe_hip
    return mpe_hip.BatchedEnv(CASE, B, dtype=dtype, device="cuda:0", **kw)


def test_layout_and_refusals():
    import mpe_hip
    g = load_golden(CASE)
    env = _env(8, torch.float64)
    assert env.obs_dims == list(g["obs_dims"]) and env.n_act == 6 and env.layout.n_hist_rows == 240 and env.layout.n_aux == 1
    assert tuple(env.comm.shape) == (240, 8)
    with pytest.raises(ValueError):
        _env(8, torch.float32, info=True)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_trajectory_matches_reference(dtype):
    g = load_golden(CASE)
    T, B, A = g["actions"].shape
    env = _env(B, getattr(torch, dtype))
    env.set_state(g["pos0"], g["vel0"])
    env.set_history(g["his_pos0"], g["his_vel0"])
    env.observe(score=True)
    assert _ok(dtype, env.obs.cpu().numpy(), g["obs0"], "consensus:obs")
    assert _ok(dtype, env.rew.cpu().numpy(), orc.consensus_reward(orc.make_spec(CASE), g["pos0"]), "consensus:rew own", 1.0, (1e-9, 2e-12))
    for t in range(T):
        obs, rew, done, _ = env.step(torch.as_tensor(g["actions"][t]))
        pos, vel, _ = env.get_state()
        assert _ok(dtype, pos, g["pos"][t], "consensus:pos"), (t, np.abs(pos - g["pos"][t]).max())
        assert _ok(dtype, vel, g["vel"][t], "consensus:vel"), t
        assert _ok(dtype, obs.cpu().numpy(), g["obs"][t], "consensus:obs", 1.0, (1e-9, 2e-12)), (t, np.abs(obs.cpu().numpy() - g["obs"][t]).max())
        # the shared reward sums six agents' terms, each with a wall penalty of slope 2 * 10 beyond 0.9
        assert _ok(dtype, rew.cpu().numpy(), g["rew"][t], "consensus:rew shared", 1.0, (1e-9, 2e-11)), t
        assert not done.any()
    hp, hv = env.get_history()
    assert _ok(dtype, hp, g["pos"][T - 10:].transpose(1, 2, 0, 3), "consensus:pos") and _ok(dtype, hv, g["vel"][T - 10:].transpose(1, 2, 0, 3), "consensus:vel")


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_rollout_and_episode_equal_stepping(dtype):
    dt = getattr(torch, dtype)
    B, T = 1000, 25
    e1, e2, e3 = _env(B, dt), _env(B, dt), _env(B, dt)
    a = torch.randint(0, 5, (T, B, 6), dtype=torch.int32, device="cuda:0", generator=torch.Generator("cuda:0").manual_seed(5))
    e1.reset(seed=11, observe=False)
    e2.reset(seed=11, observe=False)
    out2 = e2.rollout(a)
    out3 = e3.rollout(a, reset_seed=11)
    for t in range(T):
        obs, rew, done, _ = e1.step(a[t])
        assert torch.equal(obs, out2["obs"][t]) and torch.equal(rew, out2["rew"][t]) and torch.equal(done, out2["done"][t]), t
    for k in ("obs", "rew", "done"):
        assert torch.equal(out2[k], out3[k]), k
    for k in ("pos", "vel", "comm", "aux"):
        assert torch.equal(getattr(e1, k), getattr(e2, k)) and torch.equal(getattr(e1, k), getattr(e3, k)), k


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_step_auto_truncates_resets_and_keeps_returns(dtype):
    """mpe_step_auto against plain mpe_step with the bookkeeping done here: running and finished returns, the
    done flag on the step that reaches the limit, and a reset (history ring included) equal to mpe_reset's."""
    dt = getattr(torch, dtype)
    B, H = 130, 7
    auto, ref = _env(B, dt, max_episode_steps=H, seed=21), _env(B, dt)
    assert torch.equal(auto.reset(seed=21), ref.reset(seed=21))
    gen = torch.Generator("cuda:0").manual_seed(3)
    ret = torch.zeros(B, 6, dtype=dt, device="cuda:0")
    for t in range(2 * H + 3):
        a = torch.randint(0, 5, (B, 6), dtype=torch.int32, device="cuda:0", generator=gen)
        oa, ra, da, _ = auto.step(a)
        orr, rr, _, _ = ref.step(a)
        assert torch.equal(ra, rr), t
        ret += rr
        if (t + 1) % H == 0:
            assert da.all() and torch.equal(auto.last_return, ret) and not auto.ep_return.any() and not auto.step_count.any()
            ret.zero_()
            assert torch.equal(oa, ref.reset(seed=21)), t           # the next episode's draws, observed
            assert (auto.episode_count == (t + 1) // H).all()
        else:
            assert not da.any() and torch.equal(oa, orr) and torch.equal(auto.ep_return, ret), t
        for k in ("pos", "vel", "comm"):
            assert torch.equal(getattr(auto, k), getattr(ref, k)), (t, k)
        # ring heads differ after a reset (0 against wherever the stepping left it) only in position, not in content
        assert torch.equal(torch.as_tensor(auto.get_history()[0]), torch.as_tensor(ref.get_history()[0]))


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_step_auto_with_staggered_step_counts_and_coinciding_heads(dtype):
    """A masked reset ten steps into 25-step episodes: the reset worlds' ring head is 0 and so is everyone else's
    (10 % 10), so a workgroup's worlds agree on the head at launch start, but they reach their time limits at
    different steps and the heads part inside an mpe_step_auto launch.  Checked against plain mpe_step with the
    bookkeeping done here, ring content and ring heads included."""
    dt = getattr(torch, dtype)
    B, H = 200, 25
    auto, ref = _env(B, dt, max_episode_steps=H, seed=21), _env(B, dt)
    assert torch.equal(auto.reset(seed=21), ref.reset(seed=21))
    gen = torch.Generator("cuda:0").manual_seed(17)
    count = np.zeros(B, dtype=np.int64)
    episode = np.zeros(B, dtype=np.int64)
    mask = torch.zeros(B, dtype=torch.uint8)
    mask[[0, 3, 4, 31, 32, 40, 97, 98, 99, 199]] = 1
    m = mask.numpy().astype(bool)
    heads_parted = False
    for t in range(45):
        if t == 10:
            assert not auto.aux.any()               # every head is back at slot 0
            oa = auto.reset(seed=21, mask=mask)
            orr = ref.reset(seed=21, mask=mask)
            assert torch.equal(oa, orr)
            count[m] = 0
            episode[m] = 1                          # the first reset drew episode 0, this one episode 1
            assert not auto.aux.any() and not ref.aux.any()
        a = torch.randint(0, 5, (B, 6), dtype=torch.int32, device="cuda:0", generator=gen)
        oa, ra, da, _ = auto.step(a)
        orr, rr, _, _ = ref.step(a)
        assert torch.equal(ra, rr), t
        count += 1
        fin = count >= H
        assert np.array_equal(da.cpu().numpy().astype(bool), fin), t
        if fin.any():
            # the worlds that reached their limit start their next episode: the draws of mpe_reset(seed, episode + 1)
            for ep in np.unique(episode[fin] + 1):
                sel = fin & (episode + 1 == ep)
                ref._episode = int(ep)
                orr = ref.reset(seed=21, mask=torch.as_tensor(sel.astype(np.uint8)))
            episode[fin] += 1
            count[fin] = 0
        assert torch.equal(oa, orr), t
        for k in ("pos", "vel"):
            assert torch.equal(getattr(auto, k), getattr(ref, k)), (t, k)
        ha, hr = auto.get_history(), ref.get_history()          # read through each env's own ring heads
        assert np.array_equal(ha[0], hr[0]) and np.array_equal(ha[1], hr[1]), t
        assert np.array_equal(auto.aux[0].cpu().numpy(), count % 10), t
        assert np.array_equal(auto.step_count.cpu().numpy(), count) and np.array_equal(auto.episode_count.cpu().numpy(), episode), t
        heads_parted |= len(set(auto.aux[0].cpu().numpy().tolist())) == 2
    assert heads_parted


def test_large_batch_fp64_against_oracle_with_masked_reset():
    """Device reset, free-running steps; half of the worlds are reset again mid-way, so the worlds of a
    wave no longer share their ring head."""
    spec = orc.make_spec(CASE)
    B, T = 1500, 24
    env = _env(B, torch.float64)
    env.reset(seed=5, observe=False)
    pos, vel, _ = env.get_state()
    want = np.stack([orc.device_reset_draw(5, np.arange(B, dtype=np.uint64), 0, i, 1.0) for i in range(6)], axis=1)
    assert np.array_equal(pos, want) and not vel.any()
    hp, hv = env.get_history()
    assert np.array_equal(hp, np.repeat(pos[:, :, None, :], 10, axis=2)) and not hv.any()
    rng = np.random.RandomState(1)
    for t in range(T):
        if t == 7:
            mask = (np.arange(B) % 2 == 0)
            env.reset(seed=6, mask=torch.as_tensor(mask.astype(np.uint8)), observe=False)
            p2, v2, _ = env.get_state()
            assert np.array_equal(p2[~mask], pos[~mask]) and not np.array_equal(p2[mask], pos[mask])
            pos[mask], vel[mask] = p2[mask], 0.0
            hp[mask], hv[mask] = np.repeat(p2[mask][:, :, None, :], 10, axis=2), 0.0
        a = rng.randint(0, 5, (B, 6))
        obs, rew, done, _ = env.step(torch.as_tensor(a))
        pos, vel, hp, hv, o_obs, o_rew, _ = orc.consensus_env_step(spec, pos, vel, hp, hv, a)
        gp, gv, _ = env.get_state()
        assert _close(gp, pos, 1e-9, 1e-12) and _close(gv, vel, 1e-9, 1e-12), t
        assert _close(obs.cpu().numpy(), np.concatenate(o_obs, axis=1), 1e-9, 1e-12), t
        assert _close(rew.cpu().numpy(), o_rew, 1e-9, 1e-12), t
    ghp, ghv = env.get_history()
    assert _close(ghp, hp, 1e-9, 1e-12) and _close(ghv, hv, 1e-9, 1e-12)
    assert len(set(env.aux[0].cpu().numpy().tolist())) == 2


@pytest.mark.parametrize("n_masked", [40, 7])
@pytest.mark.parametrize("T", [4, 13, 25])
def test_rollout_after_a_partial_reset_equals_stepping(n_masked, T):
    """Some workgroups' worlds agree on their ring head, one holds both kinds: the whole-line and the per-thread
    form of the ring read, the slot stores (only the slots the launch does not overwrite) and the state
    write-back must leave the same state as single steps, ring included."""
    B = 200
    e1, e2 = _env(B, torch.float32), _env(B, torch.float32)
    gen = torch.Generator("cuda:0").manual_seed(8)
    for e in (e1, e2):
        e.reset(seed=3, observe=False)
    for t in range(3):
        a = torch.randint(0, 5, (B, 6), dtype=torch.int32, device="cuda:0", generator=gen)
        e1.step(a)
        e2.step(a)
    mask = torch.zeros(B, dtype=torch.uint8)
    mask[:n_masked] = 1
    for e in (e1, e2):
        e.reset(seed=4, mask=mask, observe=False)
    assert len(set(e1.aux[0].cpu().numpy().tolist())) == 2
    a = torch.randint(0, 5, (T, B, 6), dtype=torch.int32, device="cuda:0", generator=gen)
    out = e2.rollout(a)
    for t in range(T):
        obs, rew, done, _ = e1.step(a[t])
        assert torch.equal(obs, out["obs"][t]) and torch.equal(rew, out["rew"][t]) and torch.equal(done, out["done"][t]), t
    for k in ("pos", "vel", "aux"):
        assert torch.equal(getattr(e1, k), getattr(e2, k)), k
    h1, h2 = e1.get_history(), e2.get_history()
    assert np.array_equal(h1[0], h2[0]) and np.array_equal(h1[1], h2[1])


@pytest.mark.parametrize("B", [1, 5, 31, 32, 33, 64, 97, 127, 128, 129, 132, 260, 65536 + 3 * 32 + 7])
def test_ragged_batches_rollout_equals_stepping(B):
    """Workgroups of 32 worlds with fewer than 32 valid ones, batches that are no multiple of 4 (byte-wise done
    flags) and blocks of 128 done flags cut short: an episode launch (with its reset observation) against
    mpe_reset + single steps.  The largest batch is beyond 2 048 workgroups, where a launch of several steps deals
    its workgroups 9 : 7 between even and odd XCDs: every world must still be stepped exactly once."""
    T = 12 if B < 1000 else 3
    e1, e2 = _env(B, torch.float32), _env(B, torch.float32)
    a = torch.randint(0, 5, (T, B, 6), dtype=torch.int32, device="cuda:0", generator=torch.Generator("cuda:0").manual_seed(B))
    obs0 = e1.reset(seed=9)
    out = e2.rollout(a, reset_seed=9, obs0=True)
    assert torch.equal(out["obs0"].reshape(obs0.shape), obs0)
    for t in range(T):
        obs, rew, done, _ = e1.step(a[t])
        assert torch.equal(obs, out["obs"][t]) and torch.equal(rew, out["rew"][t]) and torch.equal(done, out["done"][t]), t
    for k in ("pos", "vel", "aux"):
        assert torch.equal(getattr(e1, k), getattr(e2, k)), k
    h1, h2 = e1.get_history(), e2.get_history()
    assert np.array_equal(h1[0], h2[0]) and np.array_equal(h1[1], h2[1])


def test_drop_in_env_replays_the_reference():
    import mpe_hip
    g = load_golden(CASE)
    env = mpe_hip.make_env(CASE, benchmark=True)
    assert env.shared_reward and env.n == 6
    assert env._scenario is not None          # the step is the fused kernel (plus mpe_observe for the agents' own rewards)
    env.seed(0)
    env.reset()
    b = 20
    for i, a in enumerate(env.world.agents):
        a.state.p_pos, a.state.p_vel = g["pos0"][b, i].copy(), g["vel0"][b, i].copy()
        a.state.p_his_pos, a.state.p_his_vel = g["his_pos0"][b, i].copy(), g["his_vel0"][b, i].copy()
    assert _close(np.concatenate([env._get_obs(a) for a in env.agents]), g["obs0"][b], 1e-9, 1e-12)
    for t in range(25):
        obs_n, rew_n, done, info = env.step([int(x) for x in g["actions"][t, b]])
        assert obs_n[0].dtype == np.float32 and _close(np.concatenate(obs_n), g["obs"][t, b], 1e-9, 1e-12), t
        assert _close(rew_n, g["rew"][t, b], 1e-9, 1e-12) and done is False
        assert _close([info["r%d" % i] for i in range(6)], g["rew_own"][t, b], 1e-9, 1e-12)
        assert info["n"] == [(0, 0, 0, 0)] * 6
