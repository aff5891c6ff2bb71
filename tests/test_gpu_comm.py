"""simple_speaker_listener, simple_reference and simple_crypto on the GPU against the reference
fixtures and the oracle: immovable agents, several speakers, shared rewards.

Same bars as tests/test_gpu_parity.py: float64 within 1e-9 relative (1e-12 absolute floor) of the
reference after 25 free-running steps, float32 within 1e-4; comm state and landmark indices bit-exact.
"""
import numpy as np
import pytest

import mpe_oracle as orc
from conftest import COMM_CASES, load_golden, within_fp32_bar

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= atol + rtol * np.abs(b))


def _env(case, B, dtype, info=True, **kw):
    import mpe_hip
    return mpe_hip.BatchedEnv(case, B, dtype=dtype, device="cuda:0", info=info, **kw)


def _actions(spec, rng, shape):
    """Random int actions in the reference's column order: move 0..4 if movable, word 0..dim_c-1 if speaking."""
    cols = []
    for i in range(spec.A):
        if spec.movable[i]:
            cols.append(rng.randint(0, 5, shape))
        if not spec.silent[i]:
            cols.append(rng.randint(0, spec.dim_c, shape))
    return np.stack(cols, axis=-1).astype(np.int32)


@pytest.mark.parametrize("case", COMM_CASES)
def test_layout(case):
    g = load_golden(case)
    spec = orc.make_spec(case)
    env = _env(case, 8, torch.float64)
    assert env.obs_dims == list(g["obs_dims"]) and env.n_act == g["actions"].shape[2] == spec.n_act
    assert env.layout.n_speakers == int((~spec.silent).sum()) and env.layout.n_aux == np.atleast_2d(g["goal"].T).shape[0]


@pytest.mark.parametrize("case", COMM_CASES)
def test_fp64_trajectory_matches_reference(case):
    g = load_golden(case)
    T, B, _ = g["actions"].shape
    A = g["pos"].shape[2]
    env = _env(case, B, torch.float64)
    env.set_state(g["pos0"], g["vel0"], goal=g["goal"])
    # the callbacks on the reset state (comm all zero): what MultiAgentEnv.reset returns and reward() gives
    env.observe(score=True)
    assert _close(env.obs.cpu().numpy(), g["obs0"], 1e-9, 1e-12)
    assert _close(env.rew.cpu().numpy(), g["rew0"], 1e-9, 1e-12)
    for t in range(T):
        obs, rew, done, (info_i, info_f) = env.step(torch.as_tensor(g["actions"][t]))
        pos, vel, comm = env.get_state()
        assert _close(pos[:, :A], g["pos"][t], 1e-9, 1e-12), (case, t, np.abs(pos[:, :A] - g["pos"][t]).max())
        assert _close(vel, g["vel"][t], 1e-9, 1e-12), (case, t)
        assert np.array_equal(comm, g["comm"][t]), (case, t)
        assert _close(obs.cpu().numpy(), g["obs"][t], 1e-9, 1e-12), (case, t)
        assert _close(rew.cpu().numpy(), g["rew"][t], 1e-9, 1e-12), (case, t)
        assert not done.any()
        assert np.array_equal(info_i.cpu().numpy(), np.atleast_2d(g["goal"].T).T.reshape(B, -1))
        if case == "simple_crypto":
            assert np.array_equal(info_f.cpu().numpy().reshape(B, A, -1), g["info_c_goal"][t])
        else:
            assert _close(info_f.cpu().numpy(), g["rew_own"][t], 1e-9, 1e-12)


@pytest.mark.parametrize("case", COMM_CASES)
def test_fp32_trajectory_within_1e4(case):
    g = load_golden(case)
    T, B, _ = g["actions"].shape
    A = g["pos"].shape[2]
    env = _env(case, B, torch.float32, info=False)
    env.set_state(g["pos0"], g["vel0"], goal=g["goal"])
    for t in range(T):          # every step of the free-running episode, |got - ref| <= 1e-4 * max(1, |ref|)
        obs, rew, done, _ = env.step(torch.as_tensor(g["actions"][t]))
        pos, vel, comm = env.get_state()
        assert within_fp32_bar(pos[:, :A], g["pos"][t], case + ":pos"), (t, np.abs(pos[:, :A] - g["pos"][t]).max())
        assert within_fp32_bar(vel, g["vel"][t], case + ":vel"), t
        assert np.array_equal(comm, g["comm"][t]), t
        assert within_fp32_bar(obs.cpu().numpy(), g["obs"][t], case + ":obs"), t
        assert within_fp32_bar(rew.cpu().numpy(), g["rew"][t], case + ":rew"), t


@pytest.mark.parametrize("case", COMM_CASES)
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_rollout_and_episode_equal_stepping(case, dtype):
    """mpe_rollout == repeated mpe_step, and mpe_episode == mpe_reset + mpe_rollout, bit for bit;
    B is not a multiple of the wave size."""
    dt = getattr(torch, dtype)
    spec = orc.make_spec(case)
    B, T = 1000, 25
    e1, e2, e3 = _env(case, B, dt), _env(case, B, dt), _env(case, B, dt)
    a = torch.as_tensor(_actions(spec, np.random.RandomState(5), (T, B)), device="cuda:0")
    e1.reset(seed=11, observe=False)
    e2.reset(seed=11, observe=False)
    out2 = e2.rollout(a)
    out3 = e3.rollout(a, reset_seed=11)
    for t in range(T):
        obs, rew, done, (ii, ff) = e1.step(a[t])
        assert torch.equal(obs, out2["obs"][t]) and torch.equal(rew, out2["rew"][t]) and torch.equal(done, out2["done"][t])
        assert torch.equal(ii, out2["info_i"][t]) and torch.equal(ff, out2["info_f"][t])
    for k in ("obs", "rew", "done", "info_i", "info_f"):
        assert torch.equal(out2[k], out3[k]), k
    assert torch.equal(e1.pos, e2.pos) and torch.equal(e1.pos, e3.pos) and torch.equal(e1.vel, e3.vel)
    assert torch.equal(e1.comm, e2.comm) and torch.equal(e1.comm, e3.comm) and torch.equal(e1.aux, e3.aux)


@pytest.mark.parametrize("case", COMM_CASES)
def test_large_batch_fp64_against_oracle(case):
    """On-device reset (landmark draws included), 25 free-running steps, oracle on the same initial state."""
    spec = orc.make_spec(case)
    B, T = 777, 25
    env = _env(case, B, torch.float64)
    env.reset(seed=5, observe=False)
    pos, vel, comm = env.get_state()
    goal = env.goal.cpu().numpy().astype(np.int64)
    assert goal.min() == 0 and goal.max() == spec.L - 1
    if goal.ndim == 2:      # the two draws are independent
        assert 0.3 < np.mean(goal[:, 0] == goal[:, 1]) < (0.7 if spec.L == 2 else 0.45)
    assert np.abs(pos).max() <= 1.0 and not vel.any() and not comm.any()
    rng = np.random.RandomState(1)
    for t in range(T):
        a = _actions(spec, rng, (B,))
        obs, rew, done, (info_i, info_f) = env.step(torch.as_tensor(a))
        pos, vel, ocomm, o_obs, o_rew, _ = orc.env_step(spec, pos, vel, a, goal)
        gp, gv, gc = env.get_state()
        assert _close(gp, pos, 1e-9, 1e-12) and _close(gv, vel, 1e-9, 1e-12), (case, t)
        assert np.array_equal(gc, ocomm[:, ~spec.silent].reshape(B, -1))
        assert _close(obs.cpu().numpy(), np.concatenate(o_obs, axis=1), 1e-9, 1e-12), (case, t)
        assert _close(rew.cpu().numpy(), o_rew, 1e-9, 1e-12), (case, t)
        if spec.shared:
            assert _close(info_f.cpu().numpy(), orc.reward(spec, pos, goal, ocomm), 1e-9, 1e-12)


@pytest.mark.parametrize("case", COMM_CASES)
def test_step_auto_redraws_and_clears_comm(case):
    """Time-limit auto-reset draws new landmark indices, the ones mpe_reset draws, and zeroes the comm
    state the first observation of the new episode shows."""
    spec = orc.make_spec(case)
    B, H = 300, 4
    env = _env(case, B, torch.float64, max_episode_steps=H, seed=21)
    ref = _env(case, B, torch.float64)
    env.reset(seed=21, observe=False)
    goal0 = env.goal.clone()
    a = torch.as_tensor(_actions(spec, np.random.RandomState(2), (B,)), device="cuda:0")
    total = torch.zeros(B, env.n_agents, dtype=torch.float64, device="cuda:0")
    for t in range(H):
        obs, rew, done, _ = env.step(a)
        total += rew
    assert done.all() and torch.allclose(env.last_return, total, rtol=1e-12, atol=1e-12)
    ref._episode = 1
    ref.reset(seed=21)
    assert torch.equal(env.goal, ref.goal) and torch.equal(env.pos, ref.pos)
    assert torch.equal(obs, ref.obs)
    assert not torch.equal(env.goal, goal0)


def test_out_of_range_words_select_nothing():
    """A comm int outside 0..dim_c-1 leaves the utterance all zero (the reference raises IndexError)."""
    env = _env("simple_reference", 64, torch.float32, info=False)
    env.reset(seed=1, observe=False)
    a = torch.tensor([[0, 10, 0, 15]] * 32 + [[0, 99, 0, 9]] * 32, dtype=torch.int32)
    env.step(a)
    comm = env.get_state()[2]
    assert not comm[:32].any() and not comm[32:, :10].any()
    assert np.array_equal(comm[32:, 10:], np.tile(np.eye(10)[9], (32, 1)))


def test_collaborative_flag_must_match_the_scenario():
    import mpe_hip
    spec = mpe_hip.make_spec("simple_reference")
    spec.collaborative = False
    with pytest.raises(mpe_hip.MpeError):
        mpe_hip.BatchedEnv(spec, 8)
    spec = mpe_hip.make_spec("simple_spread")
    spec.collaborative = True
    with pytest.raises(mpe_hip.MpeError):
        mpe_hip.BatchedEnv(spec, 8)


@pytest.mark.parametrize("case", COMM_CASES)
def test_drop_in_env_replays_the_reference(case):
    """make_env(name) -> MultiAgentEnv with the reference's API, one env of the fixture.  simple_reference's
    agents take (move, comm) pairs, the form the reference's Tuple space describes."""
    import mpe_hip
    g = load_golden(case)
    env = mpe_hip.make_env(case, benchmark=(case == "simple_crypto"))
    assert env.shared_reward == (case != "simple_crypto")
    env.seed(0)
    b = 7
    A = env.n
    env.reset()
    w = env.world
    for i, e in enumerate(w.entities):
        e.state.p_pos = g["pos0"][b, i].copy()
        if i < A:
            e.state.p_vel = g["vel0"][b, i].copy()
    goal = np.atleast_1d(g["goal"][b])
    if case == "simple_speaker_listener":
        w.agents[0].goal_b = w.landmarks[int(goal[0])]
    elif case == "simple_reference":
        for i in range(2):
            w.agents[i].goal_b = w.landmarks[int(goal[i])]
    else:
        for ag in w.agents:
            ag.goal_a = w.landmarks[int(goal[0])]
        w.agents[2].key = w.landmarks[int(goal[1])].color
    off = np.concatenate([[0], np.cumsum(g["obs_dims"])])
    obs_n = [env._get_obs(a) for a in env.agents]
    assert _close(np.concatenate(obs_n), g["obs0"][b], 1e-9, 1e-12)
    assert _close([env._get_reward(a) for a in env.agents], g["rew0"][b], 1e-9, 1e-12)
    for t in range(25):
        row = [int(x) for x in g["actions"][t, b]]
        act = [(row[0], row[1]), (row[2], row[3])] if case == "simple_reference" else row
        obs_n, rew_n, done, info = env.step(act)
        for i in range(A):
            assert _close(obs_n[i], g["obs"][t, b, off[i]:off[i + 1]], 1e-9, 1e-12), (t, i)
        assert _close(rew_n, g["rew"][t, b], 1e-9, 1e-12) and done is False
        assert _close([info["r%d" % i] for i in range(A)], g["rew_own"][t, b], 1e-9, 1e-12)
        assert np.array_equal(np.concatenate([a.state.c for a in w.agents if not a.silent]), g["comm"][t, b])
        if case == "simple_crypto":
            flat = np.array([np.concatenate(x) for x in info["n"]])
            assert np.array_equal(flat, g["info_c_goal"][t, b])


def test_physics_only_world_keeps_immovable_agents():
    """A user scenario with an immovable agent on the physics-only step: World.step leaves it alone
    (multiagent/core.py:195) while the others integrate as usual."""
    import mpe_hip
    from mpe_hip import Agent, Landmark, World
    w = World()
    w.dim_c = 2
    for i in range(3):
        a = Agent()
        a.silent, a.collide, a.size = True, True, 0.1
        a.state.p_pos, a.state.p_vel, a.state.c = np.array([0.15 * i, 0.0]), np.array([0.3, -0.2]), np.zeros(2)
        a.action.u, a.action.c = np.array([1.0, 0.5]), np.zeros(2)
        w.agents.append(a)
    w.agents[1].movable = False
    lm = Landmark()
    lm.state.p_pos, lm.state.p_vel = np.array([0.5, 0.5]), np.zeros(2)
    w.landmarks.append(lm)
    spec = orc.Spec("physics", 3, 1, 2, np.array([0.1, 0.1, 0.1, 0.05]), np.array([True, False, True, False]), np.ones(4, bool), np.ones(4),
                    sens=np.full(3, 5.0), max_speed=np.full(3, np.inf), silent=np.ones(3, bool), adversary=np.zeros(3, bool), leader=np.zeros(3, bool))
    pos = np.array([[e.state.p_pos for e in w.entities]])
    vel = np.array([[a.state.p_vel for a in w.agents]])
    u = np.array([[a.action.u for a in w.agents]])
    for _ in range(5):
        w.step()
        pos, vel, _ = orc.world_step(spec, pos, vel, u, np.zeros((1, 3, 2)))
    got = np.array([e.state.p_pos for e in w.entities])
    assert _close(got, pos[0], 1e-9, 1e-12)
    assert np.array_equal(got[1], [0.15, 0.0]) and np.array_equal(w.agents[1].state.p_vel, [0.3, -0.2])
