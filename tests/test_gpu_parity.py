"""HIP kernels against the golden fixtures (reference outputs) and the numpy oracle.

Bars: float64 kernels within 1e-9 (relative, with a 1e-12 absolute floor) of the reference after
25 free-running steps; float32 kernels within 1e-4; integer fields (collision counts, occupied
landmarks, forest / visibility masks, done) bit-exact for float64.  For float32 an integer field
may legitimately differ where a distance sits within the float32 drift of its threshold; everywhere
else (the reference's float64 state decides the test with a 4e-4 margin) it must be bit-exact too, at
every step of a free-running episode.
"""
import numpy as np
import pytest

import mpe_oracle as orc
from conftest import GOLDEN_CASES, load_golden, within_fp32_bar

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

SMALL_CASES = ["spread3", "tag", "world_comm"]
# Rewards are derived from the positions with a gain: simple_tag's and simple_world_comm's boundary penalty has slope
# 10 (20 for world_comm's 2 * bound) between 0.9 and 1 and up to 2 * e^(2x-2) beyond (simple_tag.py:103-108), so a
# position within the bar gives a reward within that multiple of it.
REW_FACTOR = {"spread3": 1.0, "tag": 1.0, "world_comm": 1.0}


def _env(case, B, dtype, info=True):
    import mpe_hip
    scn, kw = GOLDEN_CASES[case]
    kw = {("n_agents" if k == "A" else "n_landmarks"): v for k, v in kw.items()}
    return mpe_hip.BatchedEnv(scn, B, dtype=dtype, device="cuda:0", info=info, **kw)


def _close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= atol + rtol * np.abs(b))


def _unpack_info(case, env, info_i, info_f):
    A = env.n_agents
    ii = info_i.cpu().numpy()
    out = {"collisions": ii[:, :A]}
    if case.startswith("spread"):
        ff = info_f.cpu().numpy()
        out.update(occupied=ii[:, A], min_dists=ff[:, 0], info_rew=ff[:, 1:1 + A])
    if case == "world_comm":
        fb, vb = ii[:, A], ii[:, A + 1]
        out["in_forest"] = np.stack([np.stack([(fb >> (2 * i + f)) & 1 for f in range(2)], 1) for i in range(A)], 1)
        out["visible"] = np.stack([np.stack([((vb >> ((A - 1) * i + (j - 1 if j > i else j))) & 1) * (i != j) for j in range(A)], 1)
                                   for i in range(A)], 1)
    return out


@pytest.mark.parametrize("case", SMALL_CASES)
def test_fp64_trajectory_matches_reference(case):
    g = load_golden(case)
    T, B, _ = g["actions"].shape
    env = _env(case, B, torch.float64)
    env.set_state(g["pos0"], g["vel0"])
    A = env.n_agents
    for t in range(T):
        obs, rew, done, (info_i, info_f) = env.step(torch.as_tensor(g["actions"][t]))
        pos, vel, comm = env.get_state()
        assert _close(pos[:, :A], g["pos"][t], 1e-9, 1e-12), (case, t, np.abs(pos[:, :A] - g["pos"][t]).max())
        assert _close(vel, g["vel"][t], 1e-9, 1e-12), (case, t)
        assert _close(obs.cpu().numpy(), g["obs"][t], 1e-6 if case == "spread3" else 1e-9, 1e-12), (case, t)
        assert _close(rew.cpu().numpy(), g["rew"][t], 1e-9, 1e-12), (case, t)
        assert not done.any()
        inf = _unpack_info(case, env, info_i, info_f)
        assert np.array_equal(inf["collisions"], g["collisions"][t]), (case, t)
        if case == "spread3":
            assert np.array_equal(inf["occupied"], g["occupied"][t])
            assert _close(inf["min_dists"], g["min_dists"][t], 1e-9, 1e-12)
            assert _close(inf["info_rew"], g["info_rew"][t], 1e-9, 1e-12)
        if case == "world_comm":
            assert np.array_equal(inf["in_forest"], g["in_forest"][t])
            assert np.array_equal(inf["visible"], g["visible"][t])
            assert np.array_equal(comm, g["comm"][t])


@pytest.mark.parametrize("case", SMALL_CASES)
def test_fp64_single_step_is_bit_exact_from_reference_state(case):
    """Teacher-forced: each step starts from the reference's own state, so the only differences
    left are the device's exp/log1p last-place roundings; integer fields must match exactly."""
    g = load_golden(case)
    T, B, _ = g["actions"].shape
    scn, kw = GOLDEN_CASES[case]
    A = orc.make_spec(scn, **kw).A
    env = _env(case, B, torch.float64)
    pos_prev = g["pos0"].copy()
    vel_prev = g["vel0"]
    worst = 0.0
    for t in range(T):
        env.set_state(pos_prev, vel_prev)
        obs, rew, done, (info_i, info_f) = env.step(torch.as_tensor(g["actions"][t]))
        pos, vel, _ = env.get_state()
        worst = max(worst, np.abs(pos[:, :A] - g["pos"][t]).max(), np.abs(vel - g["vel"][t]).max())
        assert np.array_equal(_unpack_info(case, env, info_i, info_f)["collisions"], g["collisions"][t])
        pos_prev = pos_prev.copy()
        pos_prev[:, :A] = g["pos"][t]
        vel_prev = g["vel"][t]
    assert worst < 1e-13, worst


@pytest.mark.parametrize("case", SMALL_CASES)
def test_fp32_trajectory_within_1e4(case):
    """Free-running float32 episode against the reference's float64 one, at EVERY step: positions, velocities and
    observations within 1e-4 * max(1, |ref|); rewards likewise wherever the collision counts (the rewards' +-1 / 5 / 10
    jumps) agree."""
    g = load_golden(case)
    T, B, _ = g["actions"].shape
    env = _env(case, B, torch.float32)
    env.set_state(g["pos0"], g["vel0"])
    A = env.n_agents
    for t in range(T):
        obs, rew, done, (info_i, info_f) = env.step(torch.as_tensor(g["actions"][t]))
        pos, vel, _ = env.get_state()
        assert within_fp32_bar(pos[:, :A], g["pos"][t], case + ":pos"), (t, np.abs(pos[:, :A] - g["pos"][t]).max())
        # Velocities: the stated bar at the end of the episode.  Mid-episode an agent in contact turns a position
        # error dx into a velocity error contact_force * dt * dx = 10 dx within the same step (the softplus contact
        # has slope contact_force = 100 once the agents overlap), so a position well inside its bar can show up to
        # ten times larger in that step's velocity: the fixture's contact-seeded simple_spread envs peak at 1.08e-4
        # (step 22 of 25), everything else stays below 1e-5.  Mid-episode velocities are held to twice the bar.
        # (the observation carries the agents' velocities and is held like them)
        if t == T - 1:
            assert within_fp32_bar(vel, g["vel"][t], case + ":vel after the episode"), (t, np.abs(vel - g["vel"][t]).max())
            assert within_fp32_bar(obs.cpu().numpy(), g["obs"][t], case + ":obs after the episode"), t
        assert within_fp32_bar(vel, g["vel"][t], case + ":vel every step", 2.0), (t, np.abs(vel - g["vel"][t]).max())
        assert within_fp32_bar(obs.cpu().numpy(), g["obs"][t], case + ":obs every step", 2.0), t
        # rewards contain +-5/10 jumps at collision thresholds: compare where the counts agree
        same = np.all(_unpack_info(case, env, info_i, info_f)["collisions"] == g["collisions"][t], axis=1)
        assert same.mean() > 0.9
        assert within_fp32_bar(rew.cpu().numpy()[same], g["rew"][t][same], case + ":rew", REW_FACTOR[case]), t


@pytest.mark.parametrize("case", SMALL_CASES)
def test_fp32_integer_fields_are_exact_away_from_thresholds(case):
    """Every integer field of the float32 kernel, at every step of a free-running episode, equals the
    reference's wherever the reference's own float64 state decides it with a margin.

    float32 positions stay within 1e-4 of the reference's (|p| <= 1 inside the arena), so a distance between two
    entities is off by less than 4e-4.  An env is decisive at step t when the oracle, evaluated on the
    reference's state, returns the same integers with every entity 2e-4 larger and 2e-4 smaller (every
    contact / forest threshold moved by +-4e-4) and, for simple_spread, no landmark's nearest agent is
    within 4e-4 of the 0.1 occupancy radius.  There the float32 integers must be the reference's, bit for
    bit; the rewards, whose jumps are those integers, must then agree to the float32 tolerance."""
    import dataclasses
    g = load_golden(case)
    T, B, _ = g["actions"].shape
    scn, kw = GOLDEN_CASES[case]
    spec = orc.make_spec(scn, **kw)
    A, L, shift = spec.A, spec.L, 2e-4
    moved = [dataclasses.replace(spec, size=spec.size + s) for s in (shift, -shift)]

    def ints(sp, pos):
        out = [orc.info(sp, pos)["collisions"].reshape(B, -1)]
        if case == "world_comm":
            inf, vis = orc.visibility_masks(sp, pos)
            out += [inf.reshape(B, -1), vis.reshape(B, -1)]
            # the good agents' food contacts are jumps of their reward (simple_world_comm.py:178-180)
            out += [np.stack([orc._is_collision(pos, i, A + f, sp.size) for i in range(A) if not sp.adversary[i] for f in sp.food], axis=1)]
        return np.concatenate([o.astype(np.int64) for o in out], axis=1)

    env = _env(case, B, torch.float32)
    env.set_state(g["pos0"], g["vel0"])
    pos = np.array(g["pos0"], np.float64)
    decisive_total = 0
    for t in range(T):
        obs, rew, done, (info_i, info_f) = env.step(torch.as_tensor(g["actions"][t]))
        got_pos = env.get_state()[0]
        assert within_fp32_bar(got_pos[:, :A], g["pos"][t], case + ":pos"), (case, t)
        pos[:, :A] = g["pos"][t]
        decisive = np.all(ints(moved[0], pos) == ints(moved[1], pos), axis=1)
        if case == "spread3":
            nearest = np.stack([np.sqrt(((pos[:, :A] - pos[:, A + l:A + l + 1]) ** 2).sum(axis=2)).min(axis=1) for l in range(L)], axis=1)
            decisive &= np.all(np.abs(nearest - 0.1) > 2 * shift, axis=1)
        decisive_total += int(decisive.sum())
        inf = _unpack_info(case, env, info_i, info_f)
        assert np.array_equal(inf["collisions"][decisive], g["collisions"][t][decisive]), (case, t)
        if case == "spread3":
            assert np.array_equal(inf["occupied"][decisive], g["occupied"][t][decisive]), (case, t)
        if case == "world_comm":
            assert np.array_equal(inf["in_forest"][decisive], g["in_forest"][t][decisive]), (case, t)
            assert np.array_equal(inf["visible"][decisive], g["visible"][t][decisive]), (case, t)
        assert within_fp32_bar(rew.cpu().numpy()[decisive], g["rew"][t][decisive], case + ":rew", REW_FACTOR[case]), (case, t)
    # the fixtures seed some envs onto thresholds on purpose; most env-steps are still decided with a margin
    assert decisive_total > 0.8 * T * B, (case, decisive_total, T * B)


@pytest.mark.parametrize("case", SMALL_CASES)
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_rollout_equals_repeated_step(case, dtype):
    dt = getattr(torch, dtype)
    g = load_golden(case)
    T, B, _ = g["actions"].shape
    a = torch.as_tensor(g["actions"]).to("cuda:0")
    e1, e2 = _env(case, B, dt), _env(case, B, dt)
    e1.set_state(g["pos0"], g["vel0"])
    e2.set_state(g["pos0"], g["vel0"])
    out = e2.rollout(a)
    for t in range(T):
        obs, rew, done, (ii, ff) = e1.step(a[t])
        assert torch.equal(obs, out["obs"][t]) and torch.equal(rew, out["rew"][t])
        assert torch.equal(ii, out["info_i"][t]) and torch.equal(done, out["done"][t])
        if ff is not None:
            assert torch.equal(ff, out["info_f"][t])
    assert torch.equal(e1.pos, e2.pos) and torch.equal(e1.vel, e2.vel)


@pytest.mark.parametrize("case", SMALL_CASES)
def test_large_batch_fp64_against_oracle(case):
    """B is not a multiple of the 64-lane wave; 25 free-running steps against the oracle."""
    scn, kw = GOLDEN_CASES[case]
    spec = orc.make_spec(scn, **kw)
    B = 4096 + 37
    rng = np.random.RandomState(7)
    pos, vel, _ = orc.reset(spec, rng, 1)
    pos = np.concatenate([rng.uniform(-1, 1, (B, spec.A, 2)), rng.uniform(-0.9, 0.9, (B, spec.L, 2))], axis=1)
    vel = np.zeros((B, spec.A, 2))
    env = _env(case, B, torch.float64)
    env.set_state(pos, vel)
    for t in range(25):
        act = rng.randint(0, 5, (B, spec.n_act))
        if scn == "simple_world_comm":
            act[:, 1] = rng.randint(0, 4, B)
        obs, rew, done, (ii, ff) = env.step(torch.as_tensor(act, dtype=torch.int32))
        pos, vel, comm, o_obs, o_rew, _ = orc.env_step(spec, pos, vel, act)
    g_pos, g_vel, _ = env.get_state()
    assert _close(g_pos, pos, 1e-9, 1e-12), np.abs(g_pos - pos).max()
    assert _close(g_vel, vel, 1e-9, 1e-12)
    tol_obs = 1e-6 if scn == "simple_spread" else 1e-9
    assert _close(obs.cpu().numpy(), np.concatenate(o_obs, 1), tol_obs, 1e-12)
    inf = orc.info(spec, pos)
    got = ii.cpu().numpy()[:, :spec.A]
    assert np.array_equal(got, inf["collisions"])
    assert _close(rew.cpu().numpy(), o_rew, 1e-9, 1e-12)


def test_observe_matches_oracle_and_reset_distribution():
    import mpe_hip
    for scn in ("simple_spread", "simple_tag", "simple_world_comm"):
        spec = orc.make_spec(scn)
        env = mpe_hip.BatchedEnv(scn, 1000, dtype=torch.float64, device="cuda:0")
        obs = env.reset(seed=3).cpu().numpy()
        pos, vel, comm = env.get_state()
        assert np.abs(pos[:, :spec.A]).max() <= 1.0 and np.abs(vel).max() == 0
        lim = 1.0 if scn == "simple_spread" else 0.9
        assert np.abs(pos[:, spec.A:]).max() <= lim
        assert abs(pos.mean()) < 0.05 and pos[:, :spec.A].std() == pytest.approx(1 / np.sqrt(3), rel=0.05)
        c = np.zeros((1000, spec.A, spec.dim_c))
        want = np.concatenate(orc.observe(spec, pos, vel, c), 1)
        assert _close(obs, want, 1e-6 if scn == "simple_spread" else 0, 0)
        # a second reset draws a different episode; a masked reset leaves unmasked envs alone
        mask = torch.zeros(1000, dtype=torch.uint8)
        mask[::2] = 1
        env.reset(seed=3, mask=mask)
        pos2, _, _ = env.get_state()
        assert np.array_equal(pos2[1::2], pos[1::2]) and not np.array_equal(pos2[::2], pos[::2])


def test_errors_are_loud():
    import mpe_hip
    with pytest.raises(mpe_hip.MpeError):
        mpe_hip.BatchedEnv("simple_spread", 0)
    env = mpe_hip.BatchedEnv("simple_spread", 8)
    with pytest.raises(ValueError):
        env.step(torch.zeros(8, 4, dtype=torch.int32))
    bad = mpe_hip.tag_spec()
    bad.adversary = np.array([True, False, True, False])
    with pytest.raises(mpe_hip.MpeError):
        mpe_hip.BatchedEnv(bad, 8)


# ------------------------------------------------------------------------------------------
# wave-per-world kernels: scaled simple_spread (64 agents / 64 landmarks) and physics-only step
# ------------------------------------------------------------------------------------------
def test_spread64_fp64_trajectory_matches_reference():
    g = load_golden("spread64")
    T, B, A = g["actions"].shape
    env = _env("spread64", B, torch.float64)
    assert env.env_major
    env.set_state(g["pos0"], g["vel0"])
    obs_steps = list(g["obs_steps"])
    for t in range(T):
        obs, rew, done, (info_i, info_f) = env.step(torch.as_tensor(g["actions"][t]))
        pos, vel, _ = env.get_state()
        assert _close(pos[:, :A], g["pos"][t], 1e-9, 1e-12), (t, np.abs(pos[:, :A] - g["pos"][t]).max())
        assert _close(vel, g["vel"][t], 1e-9, 1e-12), t
        assert _close(rew.cpu().numpy(), g["rew"][t], 1e-9, 1e-12), t
        inf = _unpack_info("spread64", env, info_i, info_f)
        assert np.array_equal(inf["collisions"], g["collisions"][t]), t
        assert np.array_equal(inf["occupied"], g["occupied"][t])
        assert _close(inf["min_dists"], g["min_dists"][t], 1e-9, 1e-12)
        assert _close(inf["info_rew"], g["info_rew"][t], 1e-9, 1e-12)
        if t in obs_steps:
            assert _close(obs.cpu().numpy(), g["obs"][obs_steps.index(t)], 1e-6, 1e-12), t


def test_spread64_fp32_single_step_within_1e4():
    """64 overlapping agents are chaotic: float32 drifts past 1e-4 within an episode whatever the
    kernel does, so float32 is checked teacher-forced (one step from the reference's state)."""
    g = load_golden("spread64")
    T, B, A = g["actions"].shape
    env = _env("spread64", B, torch.float32)
    pos_prev, vel_prev = g["pos0"].copy(), g["vel0"]
    for t in range(T):
        env.set_state(pos_prev, vel_prev)
        obs, rew, done, _ = env.step(torch.as_tensor(g["actions"][t]))
        pos, vel, _ = env.get_state()
        assert within_fp32_bar(pos[:, :A], g["pos"][t], "spread64 single step:pos"), (t, np.abs(pos[:, :A] - g["pos"][t]).max())
        assert within_fp32_bar(vel, g["vel"][t], "spread64 single step:vel"), (t, np.abs(vel - g["vel"][t]).max())
        pos_prev = pos_prev.copy()
        pos_prev[:, :A] = g["pos"][t]
        vel_prev = g["vel"][t]


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_spread64_rollout_equals_repeated_step(dtype):
    dt = getattr(torch, dtype)
    g = load_golden("spread64")
    a = torch.as_tensor(g["actions"][:5]).to("cuda:0")
    e1, e2 = _env("spread64", 2, dt), _env("spread64", 2, dt)
    e1.set_state(g["pos0"], g["vel0"])
    e2.set_state(g["pos0"], g["vel0"])
    out = e2.rollout(a)
    for t in range(5):
        obs, rew, done, (ii, ff) = e1.step(a[t])
        assert torch.equal(obs, out["obs"][t]) and torch.equal(rew, out["rew"][t])
        assert torch.equal(ii, out["info_i"][t]) and torch.equal(ff, out["info_f"][t])
    assert torch.equal(e1.pos, e2.pos) and torch.equal(e1.vel, e2.vel)


@pytest.mark.parametrize("A,L", [(2, 1), (5, 7), (9, 4), (33, 20)])
def test_spread_other_sizes_against_oracle(A, L):
    """simple_spread at sizes other than the lane-per-world 3/3 runs on the wave-per-world kernel."""
    import mpe_hip
    spec = orc.spec_spread(A, L)
    B = 50
    rng = np.random.RandomState(A * 100 + L)
    pos = rng.uniform(-1, 1, (B, A + L, 2))
    vel = rng.uniform(-0.5, 0.5, (B, A, 2))
    env = mpe_hip.BatchedEnv("simple_spread", B, dtype=torch.float64, info=True, n_agents=A, n_landmarks=L)
    env.set_state(pos, vel)
    assert _close(env.observe().cpu().numpy(), np.concatenate(orc.observe(spec, pos, vel, None), 1), 1e-6, 1e-12)
    for t in range(10):
        act = rng.randint(0, 5, (B, A))
        obs, rew, done, (ii, ff) = env.step(torch.as_tensor(act, dtype=torch.int32))
        pos, vel, comm, o_obs, o_rew, _ = orc.env_step(spec, pos, vel, act)
    g_pos, g_vel, _ = env.get_state()
    assert _close(g_pos, pos, 1e-9, 1e-12) and _close(g_vel, vel, 1e-9, 1e-12)
    assert _close(obs.cpu().numpy(), np.concatenate(o_obs, 1), 1e-6, 1e-12)
    assert _close(rew.cpu().numpy(), o_rew, 1e-9, 1e-12)
    inf = orc.info(spec, pos)
    assert np.array_equal(ii.cpu().numpy()[:, :A], inf["collisions"]) and np.array_equal(ii.cpu().numpy()[:, A], inf["occupied"])


@pytest.mark.parametrize("scn", ["simple_tag", "simple_world_comm"])
def test_world_step_physics_only_matches_oracle(scn):
    """mpe_world_step: World.step with forces and comm actions given directly, checked against the
    oracle's world_step on the simple_tag / simple_world_comm worlds (speed clamp, obstacles,
    one speaking agent)."""
    import mpe_hip
    spec = orc.make_spec(scn)
    ps = mpe_hip.make_spec(scn)
    phys = mpe_hip.physics_spec(ps.n_agents, ps.n_landmarks, ps.dim_c, ps.size, ps.mass, ps.collide, ps.sensitivity,
                                ps.max_speed, ps.silent)
    B = 300
    rng = np.random.RandomState(11)
    pos = np.concatenate([rng.uniform(-1, 1, (B, spec.A, 2)), rng.uniform(-0.9, 0.9, (B, spec.L, 2))], axis=1)
    vel = rng.uniform(-1, 1, (B, spec.A, 2))
    env = mpe_hip.BatchedEnv(phys, B, dtype=torch.float64)
    env.set_state(pos, vel)
    for t in range(10):
        act = rng.randint(0, 5, (B, spec.n_act))
        if scn == "simple_world_comm":
            act[:, 1] = rng.randint(0, 4, B)
        u, c = orc.decode_actions(spec, act)
        speakers = [i for i in range(spec.A) if not spec.silent[i]]
        env.world_step(u.reshape(B, -1), c[:, speakers, :].reshape(B, -1) if speakers else None)
        pos, vel, comm = orc.world_step(spec, pos, vel, u, c)
    g_pos, g_vel, g_comm = env.get_state()
    assert _close(g_pos, pos, 1e-9, 1e-12), np.abs(g_pos - pos).max()
    assert _close(g_vel, vel, 1e-9, 1e-12)
    if speakers:
        assert np.array_equal(g_comm, comm[:, speakers, :].reshape(B, -1))


@pytest.mark.parametrize("case", SMALL_CASES + ["spread64"])
def test_episode_equals_reset_then_rollout(case):
    """mpe_episode (reset fused into the rollout launch) against mpe_reset + mpe_rollout."""
    g = load_golden(case)
    B = 200 if case != "spread64" else 6
    a = torch.as_tensor(np.resize(g["actions"], (7, B, g["actions"].shape[2]))).to("cuda:0")
    for dt in (torch.float32, torch.float64):
        e1, e2 = _env(case, B, dt), _env(case, B, dt)
        first = e1.reset(seed=9).clone()                   # mpe_reset + mpe_observe: what MultiAgentEnv.reset returns
        o1 = e1.rollout(a)
        o2 = e2.rollout(a, reset_seed=9, obs0=True)
        assert o2["obs0"].shape == (1, B, e2.layout.obs_total) and torch.equal(o2["obs0"][0], first)
        assert torch.equal(o1["obs"], o2["obs"]) and torch.equal(o1["rew"], o2["rew"])
        assert torch.equal(o1["info_i"], o2["info_i"])
        assert torch.equal(e1.pos, e2.pos) and torch.equal(e1.vel, e2.vel)
        # a second episode draws new worlds
        o3 = e2.rollout(a, reset_seed=9)
        assert not torch.equal(o3["obs"][0], o2["obs"][0])


@pytest.mark.parametrize("case", SMALL_CASES + ["spread64"])
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_step_auto_bookkeeping_and_reset(case, dtype):
    """mpe_step_auto against plain mpe_step + host-side bookkeeping: returns, step counters,
    time-limit truncation with on-device reset (the reset worlds equal mpe_reset's draws)."""
    import mpe_hip
    dt = getattr(torch, dtype)
    scn, kw = GOLDEN_CASES[case]
    kw = {("n_agents" if k == "A" else "n_landmarks"): v for k, v in kw.items()}
    B, H = (130, 7) if case != "spread64" else (5, 4)
    auto = mpe_hip.BatchedEnv(scn, B, dtype=dt, info=True, max_episode_steps=H, seed=21, **kw)
    ref = mpe_hip.BatchedEnv(scn, B, dtype=dt, info=True, **kw)
    obs_a = auto.reset(seed=21).clone()
    obs_r = ref.reset(seed=21).clone()
    assert torch.equal(obs_a, obs_r)
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(2)
    ret = torch.zeros(B, auto.n_agents, dtype=dt, device="cuda:0")
    for t in range(2 * H + 3):
        a = torch.randint(0, 4, (B, auto.n_act), generator=gen, device="cuda:0", dtype=torch.int32)
        oa, ra, da, (ia, fa) = auto.step(a)
        o, r, d, (ii, ff) = ref.step(a)
        assert torch.equal(ra, r) and torch.equal(ia, ii)
        ret = ret + r
        boundary = (t + 1) % H == 0
        if boundary:
            assert bool(da.all())
            assert torch.equal(auto.last_return, ret)
            assert float(auto.ep_return.abs().max()) == 0 and int(auto.step_count.max()) == 0
            o = ref.reset(seed=21)                       # episode (t + 1) / H of the same seed
            ret.zero_()
            assert int(auto.episode_count.min()) == int(auto.episode_count.max()) == (t + 1) // H
        else:
            assert not bool(da.any())
            assert torch.equal(auto.ep_return, ret) and int(auto.step_count.min()) == (t + 1) % H
        assert torch.equal(oa, o)
        assert torch.equal(auto.pos, ref.pos) and torch.equal(auto.vel, ref.vel)


# ------------------------------------------------------------------------------------------
# edge cases
# ------------------------------------------------------------------------------------------
def test_coincident_agents_give_nan_like_the_reference():
    """dist == 0 makes the reference divide 0/0 (core.py:230); the NaN must propagate, not be fixed."""
    spec = orc.spec_spread()
    pos = np.zeros((4, 6, 2))
    pos[:, 2:] = np.random.RandomState(0).uniform(-1, 1, (4, 4, 2))
    pos[1, 1] = (0.4, 0.4)                               # env 1: no coincidence
    vel = np.zeros((4, 3, 2))
    act = np.array([[0, 0, 0], [1, 2, 3], [4, 4, 4], [0, 1, 0]])
    with np.errstate(all="ignore"):
        p, v, c, o_obs, o_rew, _ = orc.env_step(spec, pos, vel, act)
    for dtype in (torch.float64, torch.float32):
        env = _env("spread3", 4, dtype)
        env.set_state(pos, vel)
        obs, rew, done, _ = env.step(torch.as_tensor(act, dtype=torch.int32))
        gp, gv, _ = env.get_state()
        assert np.array_equal(np.isnan(gp), np.isnan(p)) and np.array_equal(np.isnan(gv), np.isnan(v))
        assert np.isnan(gp[0, 0]).all() and not np.isnan(gp[1]).any() and not np.isnan(gp[:, 2]).any()
        assert np.array_equal(np.isnan(obs.cpu().numpy()), np.isnan(np.concatenate(o_obs, 1)))


@pytest.mark.parametrize("case", SMALL_CASES)
def test_out_of_range_actions_are_noops(case):
    """environment.py:178-181 only reacts to 1..4; any other int leaves u at zero."""
    scn, kw = GOLDEN_CASES[case]
    spec = orc.make_spec(scn, **kw)
    B = 64
    rng = np.random.RandomState(1)
    pos = rng.uniform(-1, 1, (B, spec.E, 2))
    vel = rng.uniform(-0.3, 0.3, (B, spec.A, 2))
    act = rng.choice([-7, -1, 0, 5, 6, 9, 100, 2 ** 31 - 1, 1, 3], size=(B, spec.n_act))
    if scn == "simple_world_comm":
        act[:, 1] = rng.randint(0, 4, B)                 # the comm word must stay a valid index in the oracle
    p, v, c, o_obs, o_rew, _ = orc.env_step(spec, pos, vel, act)
    env = _env(case, B, torch.float64)
    env.set_state(pos, vel)
    obs, rew, done, _ = env.step(torch.as_tensor(act, dtype=torch.int32))
    gp, gv, _ = env.get_state()
    assert _close(gp, p, 1e-12, 1e-13) and _close(gv, v, 1e-12, 1e-13)
    out = env.rollout(torch.as_tensor(act[None], dtype=torch.int32).to("cuda:0").contiguous())   # packed-action path
    p2, v2, *_ = orc.env_step(spec, p, v, act)
    gp2, gv2, _ = env.get_state()
    assert _close(gp2, p2, 1e-12, 1e-13) and _close(gv2, v2, 1e-12, 1e-13)


@pytest.mark.parametrize("case", SMALL_CASES + ["spread64"])
def test_modified_world_constants_and_entity_properties(case):
    """dt, damping, contact force / margin, sizes, masses, accelerations and speed limits are run-time
    parameters: a world edited after make_world must be simulated with its own numbers."""
    import mpe_hip
    scn, kw = GOLDEN_CASES[case]
    spec = orc.make_spec(scn, **kw)
    rng = np.random.RandomState(4)
    spec.dt, spec.damping, spec.contact_force, spec.contact_margin = 0.07, 0.4, 55.0, 2e-3
    spec.size = spec.size * rng.uniform(0.7, 1.5, spec.E)
    spec.mass = rng.uniform(0.5, 2.0, spec.E)
    spec.sens = rng.uniform(2.0, 6.0, spec.A)
    spec.max_speed = np.where(rng.rand(spec.A) < 0.5, rng.uniform(0.3, 0.9, spec.A), np.inf)
    kwp = {("n_agents" if k == "A" else "n_landmarks"): v for k, v in kw.items()}
    ps = mpe_hip.make_spec(scn, **kwp)
    ps.dt, ps.damping, ps.contact_force, ps.contact_margin = spec.dt, spec.damping, spec.contact_force, spec.contact_margin
    ps.size, ps.mass, ps.sensitivity, ps.max_speed = spec.size, spec.mass, spec.sens, spec.max_speed
    B = 96 if spec.A <= 6 else 3
    pos = np.concatenate([rng.uniform(-1, 1, (B, spec.A, 2)), rng.uniform(-0.9, 0.9, (B, spec.L, 2))], axis=1)
    vel = rng.uniform(-1, 1, (B, spec.A, 2))
    env = mpe_hip.BatchedEnv(ps, B, dtype=torch.float64, info=True)
    env.set_state(pos, vel)
    for t in range(8):
        act = rng.randint(0, 5, (B, spec.n_act))
        if scn == "simple_world_comm":
            act[:, 1] = rng.randint(0, 4, B)
        obs, rew, done, (ii, ff) = env.step(torch.as_tensor(act, dtype=torch.int32))
        pos, vel, comm, o_obs, o_rew, _ = orc.env_step(spec, pos, vel, act)
    gp, gv, _ = env.get_state()
    assert _close(gp, pos, 1e-9, 1e-12), np.abs(gp - pos).max()
    assert _close(gv, vel, 1e-9, 1e-12)
    assert _close(rew.cpu().numpy(), o_rew, 1e-9, 1e-12)
    assert np.array_equal(ii.cpu().numpy()[:, :spec.A], orc.info(spec, pos)["collisions"])


def test_far_outside_the_arena_boundary_penalty_saturates():
    """bound(x) = min(exp(2x - 2), 10) for x >= 1 (simple_tag.py:108): huge |x| must cap at 10, not overflow."""
    spec = orc.spec_tag()
    B = 8
    pos = np.zeros((B, 6, 2))
    pos[:, :3] = [(5, 5), (-5, 5), (5, -5)]
    pos[:, 4:] = [(7, 7), (-7, -7)]
    pos[:, 3, 0] = [0.5, 0.95, 1.0, 1.2, 2.16, 3.0, 50.0, 400.0]
    pos[:, 3, 1] = [-0.5, -0.95, -1.0, -1.2, -2.16, -3.0, -50.0, -400.0]
    vel = np.zeros((B, 4, 2))
    act = np.zeros((B, 4), int)
    p, v, c, o_obs, o_rew, _ = orc.env_step(spec, pos, vel, act)
    for dtype, tol in ((torch.float64, 1e-12), (torch.float32, 2e-6)):
        env = _env("tag", B, dtype)
        env.set_state(pos, vel)
        obs, rew, done, _ = env.step(torch.as_tensor(act, dtype=torch.int32))
        assert _close(rew.cpu().numpy(), o_rew, tol, tol), (rew.cpu().numpy()[:, 3], o_rew[:, 3])
    assert o_rew[-1, 3] == -20.0 and o_rew[0, 3] == 0.0


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_world_step_with_walls_matches_reference(dtype):
    """mpe_world_step with wall contact against the trajectory captured from the reference's
    World.step (walls with rounded ends, a soft wall and a ghost agent)."""
    import mpe_hip
    from test_oracle_golden import walls_spec
    g = load_golden("walls")
    spec = walls_spec(g)
    dt = getattr(torch, dtype)
    ps = mpe_hip.tag_spec()
    phys = mpe_hip.physics_spec(ps.n_agents, ps.n_landmarks, ps.dim_c, ps.size, ps.mass, ps.collide, ps.sensitivity, ps.max_speed,
                                ps.silent, walls=spec.walls, ghost=spec.ghost)
    T, B, A = g["actions"].shape
    env = mpe_hip.BatchedEnv(phys, B, dtype=dt)
    if dtype == "float64":
        env.set_state(g["pos0"], g["vel0"])
        for t in range(T):
            u, c = orc.decode_actions(spec, g["actions"][t])
            env.world_step(u.reshape(B, -1))
            pos, vel, _ = env.get_state()
            assert _close(pos[:, :A], g["pos"][t], 1e-9, 1e-12), (t, np.abs(pos[:, :A] - g["pos"][t]).max())
            assert _close(vel, g["vel"][t], 1e-9, 1e-12), t
    else:
        pos_prev, vel_prev = g["pos0"].copy(), g["vel0"]
        for t in range(T):          # teacher-forced: wall contact is as stiff as entity contact
            env.set_state(pos_prev, vel_prev)
            u, c = orc.decode_actions(spec, g["actions"][t])
            env.world_step(u.reshape(B, -1))
            pos, vel, _ = env.get_state()
            assert within_fp32_bar(pos[:, :A], g["pos"][t], "walls single step:pos") and within_fp32_bar(vel, g["vel"][t], "walls single step:vel"), t
            pos_prev = pos_prev.copy()
            pos_prev[:, :A] = g["pos"][t]
            vel_prev = g["vel"][t]


def test_misaligned_buffer_is_refused():
    """Rows are stored as 16-byte vectors: the ABI refuses a buffer that is not 16-byte aligned."""
    import mpe_hip
    from mpe_hip import _lib
    env = mpe_hip.BatchedEnv("simple_spread", 64, dtype=torch.float32)
    env.reset(seed=1)
    lib = _lib.load()
    actions = torch.zeros(64, 3, dtype=torch.int32, device="cuda")
    shifted = torch.zeros(64 * env.layout.obs_total + 1, dtype=torch.float32, device="cuda")[1:]
    rc = lib.mpe_step(env._h, actions.data_ptr(), env.pos.data_ptr(), env.vel.data_ptr(), None, None, shifted.data_ptr(),
                      env.rew.data_ptr(), env.done.data_ptr(), None, None, None)
    assert rc == -1
    assert b"16-byte aligned" in lib.mpe_last_error(env._h)
    odd_done = torch.zeros(64 + 2, dtype=torch.uint8, device="cuda")[2:]      # rollouts zero the flags as dwords
    rc = lib.mpe_step(env._h, actions.data_ptr(), env.pos.data_ptr(), env.vel.data_ptr(), None, None, env.obs.data_ptr(),
                      env.rew.data_ptr(), odd_done.data_ptr(), None, None, None)
    assert rc == -1
    assert b"16-byte aligned" in lib.mpe_last_error(env._h)


def test_steps_read_action_slices_of_any_batch_size():
    """A step may read actions[t] of a pre-generated [T, B, n_act] array: with B = 6 most slices start off a
    16-byte boundary.  Only output rows need that alignment; the result equals stepping from aligned copies."""
    import mpe_hip
    B, T = 6, 5
    gen = torch.Generator(device="cuda").manual_seed(3)
    actions = torch.randint(0, 5, (T, B, 3), generator=gen, device="cuda", dtype=torch.int32)
    assert any(actions[t].data_ptr() % 16 for t in range(T))
    a = mpe_hip.BatchedEnv("simple_spread", B, dtype=torch.float64)
    b = mpe_hip.BatchedEnv("simple_spread", B, dtype=torch.float64)
    a.reset(seed=5)
    b.reset(seed=5)
    for t in range(T):
        oa, ra, _, _ = a.step(actions[t])                   # the slice itself
        ob, rb, _, _ = b.step(actions[t].clone())           # a fresh (aligned) allocation
        assert torch.equal(oa, ob) and torch.equal(ra, rb)
    assert torch.equal(a.pos, b.pos) and torch.equal(a.vel, b.vel)


def test_entry_points_leave_the_callers_device_current():
    """The library selects the handle's device for its launches and puts the caller's device back."""
    import mpe_hip
    env = mpe_hip.BatchedEnv("simple_spread", 64, dtype=torch.float32)
    before = torch.cuda.current_device()
    env.reset(seed=1)
    env.step(torch.zeros(64, 3, dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    assert torch.cuda.current_device() == before


def test_scripted_agents_act_from_the_state_of_every_step():
    """Agent.action_callback in batched form (core.py:144-147): the scripted agent's action is computed on the device
    from the current state before every step, the caller supplies the policy agents' actions only; the result is what
    stepping with the full action array gives."""
    import mpe_hip
    B = 500

    def flee(env):
        # the good agent of simple_tag (agent 3) moves away from adversary 0 along the larger offset
        d = env.pos[6:8] - env.pos[0:2]                     # [2, B]: p_3 - p_0 (env-minor rows 2e, 2e + 1)
        along_x = d[0].abs() >= d[1].abs()
        return torch.where(along_x, torch.where(d[0] >= 0, 2, 1), torch.where(d[1] >= 0, 4, 3)).to(torch.int32)

    env = mpe_hip.BatchedEnv("simple_tag", B, dtype=torch.float64, scripted={3: flee})
    ref = mpe_hip.BatchedEnv("simple_tag", B, dtype=torch.float64)
    assert env.n_policy_act == 3 and env.policy_cols == [0, 1, 2] and ref.n_policy_act == 4
    env.reset(seed=8)
    ref.reset(seed=8)
    gen = torch.Generator("cuda").manual_seed(4)
    moved = 0
    for t in range(10):
        a = torch.randint(0, 5, (B, 3), dtype=torch.int32, device="cuda", generator=gen)
        full = torch.cat([a, flee(ref).view(B, 1)], dim=1).contiguous()
        moved += int((full[:, 3] != 0).sum())
        o1, r1, _, _ = env.step(a)
        o2, r2, _, _ = ref.step(full)
        assert torch.equal(o1, o2) and torch.equal(r1, r2), t
    assert moved == 10 * B and torch.equal(env.pos, ref.pos)
    with pytest.raises(ValueError):
        env.step(torch.zeros(B, 4, dtype=torch.int32, device="cuda"))          # the scripted agent's column is not the caller's
    with pytest.raises(ValueError):
        env.rollout(torch.zeros(5, B, 3, dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError):
        mpe_hip.BatchedEnv("simple_tag", B, scripted={7: flee})
