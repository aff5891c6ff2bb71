"""simple, simple_adversary and simple_push on the GPU against the reference fixtures and the oracle.

Same bars as tests/test_gpu_parity.py: float64 within 1e-9 relative (1e-12 absolute floor) of the
reference after 25 free-running steps, float32 within 1e-4, the goal landmark index bit-exact.
"""
import numpy as np
import pytest

import mpe_oracle as orc
from conftest import CLASSIC_CASES, load_golden, within_fp32_bar

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= atol + rtol * np.abs(b))


def _env(case, B, dtype, **kw):
    import mpe_hip
    return mpe_hip.BatchedEnv(case, B, dtype=dtype, device="cuda:0", info=(case == "simple_adversary"), **kw)


def _inject(env, g, case):
    env.set_state(g["pos0"], g["vel0"], goal=None if case == "simple" else g["goal"])


@pytest.mark.parametrize("case", CLASSIC_CASES)
def test_fp64_trajectory_matches_reference(case):
    g = load_golden(case)
    T, B, A = g["actions"].shape
    env = _env(case, B, torch.float64)
    assert env.obs_dims == list(g["obs_dims"])
    _inject(env, g, case)
    for t in range(T):
        obs, rew, done, (info_i, info_f) = env.step(torch.as_tensor(g["actions"][t]))
        pos, vel, _ = env.get_state()
        assert _close(pos[:, :A], g["pos"][t], 1e-9, 1e-12), (case, t, np.abs(pos[:, :A] - g["pos"][t]).max())
        assert _close(vel, g["vel"][t], 1e-9, 1e-12), (case, t)
        assert _close(obs.cpu().numpy(), g["obs"][t], 1e-9, 1e-12), (case, t)
        assert _close(rew.cpu().numpy(), g["rew"][t], 1e-9, 1e-12), (case, t)
        assert not done.any()
        if case == "simple_adversary":
            assert np.array_equal(info_i.cpu().numpy()[:, 0], g["goal"])
            assert _close(info_f.cpu().numpy(), g["info_sq"][t], 1e-9, 1e-12)


@pytest.mark.parametrize("case", CLASSIC_CASES)
def test_fp32_trajectory_within_1e4(case):
    g = load_golden(case)
    T, B, A = g["actions"].shape
    env = _env(case, B, torch.float32)
    _inject(env, g, case)
    for t in range(T):          # every step of the free-running episode, |got - ref| <= 1e-4 * max(1, |ref|)
        obs, rew, done, _ = env.step(torch.as_tensor(g["actions"][t]))
        pos, vel, _ = env.get_state()
        assert within_fp32_bar(pos[:, :A], g["pos"][t], case + ":pos"), (t, np.abs(pos[:, :A] - g["pos"][t]).max())
        assert within_fp32_bar(vel, g["vel"][t], case + ":vel"), t
        assert within_fp32_bar(obs.cpu().numpy(), g["obs"][t], case + ":obs"), t
        assert within_fp32_bar(rew.cpu().numpy(), g["rew"][t], case + ":rew"), t


@pytest.mark.parametrize("case", CLASSIC_CASES)
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_rollout_and_episode_equal_stepping(case, dtype):
    """mpe_rollout == repeated mpe_step, and mpe_episode == mpe_reset + mpe_rollout, bit for bit;
    B is not a multiple of the wave size."""
    dt = getattr(torch, dtype)
    B, T = 1000, 25
    e1, e2, e3 = _env(case, B, dt), _env(case, B, dt), _env(case, B, dt)
    a = torch.randint(0, 5, (T, B, e1.n_act), dtype=torch.int32, device="cuda:0", generator=torch.Generator("cuda:0").manual_seed(5))
    e1.reset(seed=11, observe=False)
    e2.reset(seed=11, observe=False)
    out2 = e2.rollout(a)
    out3 = e3.rollout(a, reset_seed=11)
    for t in range(T):
        obs, rew, done, _ = e1.step(a[t])
        assert torch.equal(obs, out2["obs"][t]) and torch.equal(rew, out2["rew"][t]) and torch.equal(done, out2["done"][t])
    for k in ("obs", "rew", "done"):
        assert torch.equal(out2[k], out3[k]), k
    assert torch.equal(e1.pos, e2.pos) and torch.equal(e1.pos, e3.pos) and torch.equal(e1.vel, e3.vel)
    if e1.aux is not None:
        assert torch.equal(e1.aux, e3.aux)


@pytest.mark.parametrize("case", CLASSIC_CASES)
def test_large_batch_fp64_against_oracle(case):
    """On-device reset (goal included), 25 free-running steps, oracle on the same initial state."""
    spec = orc.make_spec(case)
    B, T = 777, 25
    env = _env(case, B, torch.float64)
    env.reset(seed=5, observe=False)
    pos, vel, _ = env.get_state()
    goal = None if env.goal is None else env.goal.cpu().numpy().astype(np.int64)
    if goal is not None:
        assert goal.min() == 0 and goal.max() == spec.L - 1 and 0.4 < goal.mean() < 0.6
    assert np.abs(pos).max() <= 1.0 and not vel.any()
    rng = np.random.RandomState(1)
    for t in range(T):
        a = rng.randint(0, 5, (B, spec.n_act))
        obs, rew, done, _ = env.step(torch.as_tensor(a))
        pos, vel, comm, o_obs, o_rew, _ = orc.env_step(spec, pos, vel, a, goal)
        gp, gv, _ = env.get_state()
        assert _close(gp, pos, 1e-9, 1e-12) and _close(gv, vel, 1e-9, 1e-12), (case, t)
        assert _close(obs.cpu().numpy(), np.concatenate(o_obs, axis=1), 1e-9, 1e-12), (case, t)
        assert _close(rew.cpu().numpy(), o_rew, 1e-9, 1e-12), (case, t)


@pytest.mark.parametrize("case", ["simple_adversary", "simple_push"])
def test_step_auto_redraws_the_goal(case):
    """Time-limit auto-reset draws a new goal with the new episode, the same one mpe_reset draws."""
    B, H = 300, 4
    env = _env(case, B, torch.float64, max_episode_steps=H, seed=21)
    ref = _env(case, B, torch.float64)
    env.reset(seed=21, observe=False)
    goal0 = env.goal.clone()
    a = torch.zeros(B, env.n_act, dtype=torch.int32, device="cuda:0")
    for t in range(H):
        obs, rew, done, _ = env.step(a)
    assert done.all()
    ref._episode = 1
    ref.reset(seed=21)
    assert torch.equal(env.goal, ref.goal) and torch.equal(env.pos, ref.pos)
    assert torch.equal(obs, ref.obs)
    assert not torch.equal(env.goal, goal0)


def test_missing_goal_buffer_and_info_are_refused():
    import mpe_hip
    from mpe_hip import _lib
    env = _env("simple_push", 64, torch.float32)
    lib = _lib.load()
    a = torch.zeros(64, 2, dtype=torch.int32, device="cuda")
    rc = lib.mpe_step(env._h, a.data_ptr(), env.pos.data_ptr(), env.vel.data_ptr(), None, None, env.obs.data_ptr(),
                      env.rew.data_ptr(), env.done.data_ptr(), None, None, None)
    assert rc == -1 and b"aux" in lib.mpe_last_error(env._h)
    with pytest.raises(ValueError):
        mpe_hip.BatchedEnv("simple_push", 8, info=True)


@pytest.mark.parametrize("case", CLASSIC_CASES)
def test_drop_in_env_replays_the_reference(case):
    """make_env(name) -> MultiAgentEnv with the reference's API, one env of the fixture."""
    from multiagent.environment import MultiAgentEnv  # noqa: F401  (alias package resolves)
    import mpe_hip
    g = load_golden(case)
    env = mpe_hip.make_env(case, benchmark=(case == "simple_adversary"))
    env.seed(0)
    b = 5
    A = env.n
    env.reset()
    w = env.world
    for i, e in enumerate(w.entities):
        e.state.p_pos = g["pos0"][b, i].copy()
        if i < A:
            e.state.p_vel = g["vel0"][b, i].copy()
    if case != "simple":
        for ag in w.agents:
            ag.goal_a = w.landmarks[int(g["goal"][b])]
    off = np.concatenate([[0], np.cumsum(g["obs_dims"])])
    for t in range(25):
        obs_n, rew_n, done, info = env.step([int(x) for x in g["actions"][t, b]])
        for i in range(A):
            assert _close(obs_n[i], g["obs"][t, b, off[i]:off[i + 1]], 1e-9, 1e-12), (t, i)
        assert _close(rew_n, g["rew"][t, b], 1e-9, 1e-12) and done is False
        if case == "simple_adversary":
            flat = np.concatenate([np.atleast_1d(x) for x in info["n"]])
            assert _close(flat, g["info_sq"][t, b], 1e-9, 1e-12)
