"""simple_predator_prey (new_core physics, walls, post_step pick-up / deposit / respawn) on the GPU.

The reference draws respawn positions from numpy's global generator, the kernel from a counter-based
stream, so the fixture is replayed teacher-forced (every step starts from the reference's state) and
free-running only until a food respawns; free-running parity over whole episodes is checked against
the oracle, which can evaluate the device's stream.  Bars as in tests/test_gpu_parity.py: float64
1e-9 relative (1e-12 floor), float32 1e-4; flags and collision counts bit-exact.
"""
import numpy as np
import pytest

import mpe_oracle as orc
from conftest import decided_with_margin, load_golden, within_fp32_bar

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
CASE = "simple_predator_prey"


def _close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= atol + rtol * np.abs(b))


def _ok(dtype, got, ref, label, f32_factor=1.0, f64=(1e-9, 1e-12)):
    """float64: within f64 = (rtol, atol) of the reference; float32: the stated bar |got - ref| <= f32_factor * 1e-4 *
    max(1, |ref|), its observed margin recorded under `label`."""
    if dtype == "float64":
        return _close(got, ref, *f64)
    return within_fp32_bar(got, ref, label, f32_factor)


def _env(B, dtype, info=True, **kw):
    import mpe_hip
    return mpe_hip.BatchedEnv(CASE, B, dtype=dtype, device="cuda:0", info=info, **kw)


def _flags(alive, holding):
    return (alive.astype(np.int64) << np.arange(3)).sum(axis=1) | ((holding.astype(np.int64) << (3 + np.arange(3))).sum(axis=1))


def _aux(alive, holding, key=None, count=None):
    B = alive.shape[0]
    return np.stack([_flags(alive, holding), np.zeros(B, np.int64) if key is None else key, np.zeros(B, np.int64) if count is None else count], axis=1)


def _state_before(g, t):
    """The reference's world at the start of step t: (pos [B, E, 2], vel, alive, holding)."""
    A = g["pos"].shape[2]
    pos = g["pos0"].copy()
    if t == 0:
        return pos, g["vel0"], g["alive0"], g["holding0"]
    pos[:, :A] = g["pos"][t - 1]
    pos[:, A + 1:A + 4] = g["food_pos"][t - 1]
    return pos, g["vel"][t - 1], g["alive"][t - 1], g["holding"][t - 1]


def test_layout():
    g = load_golden(CASE)
    env = _env(8, torch.float64)
    assert env.obs_dims == list(g["obs_dims"]) and env.n_act == 7 and env.layout.n_aux == 3 and env.layout.n_info_i == 8


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_teacher_forced_steps_match_reference(dtype):
    g = load_golden(CASE)
    T, B, A = g["actions"].shape
    env = _env(B, getattr(torch, dtype))
    pos, vel, alive, holding = _state_before(g, 0)
    env.set_state(pos, vel, goal=_aux(alive, holding))
    env.observe()
    assert _ok(dtype, env.obs.cpu().numpy(), g["obs0"], "simple_predator_prey reset:obs")
    exact = dtype == "float64"
    spec = orc.make_spec(CASE)
    sure_total = 0
    for t in range(T):
        pos, vel, alive, holding = _state_before(g, t)
        env.set_state(pos, vel, goal=_aux(alive, holding))
        obs, rew, done, (info_i, _) = env.step(torch.as_tensor(g["actions"][t]))
        gp, gv, _ = env.get_state()
        assert _ok(dtype, gp[:, :A], g["pos"][t], "simple_predator_prey single step:pos"), (t, np.abs(gp[:, :A] - g["pos"][t]).max())
        # teacher-forced: a position within the bar bounds the velocity (position change / dt) to 10 times it
        assert _ok(dtype, gv, g["vel"][t], "simple_predator_prey single step:vel", 10.0, (1e-8, 1e-11)), t
        # float32 may flip a threshold test of the fixture's knife-edge placements (a distance within the float32
        # error of a contact radius): the float64 kernel must agree everywhere, the float32 kernel in every env whose
        # tests the reference's own positions decide with a 4e-4 margin (the position tolerance is 1e-4)
        ii = info_i.cpu().numpy()
        after = pos.copy()
        after[:, :A] = g["pos"][t]
        sure = np.ones(B, bool) if exact else decided_with_margin(spec.size, after, 4e-4)
        sure_total += int(sure.sum())
        assert _ok(dtype, obs.cpu().numpy()[sure], g["obs"][t][sure], "simple_predator_prey single step:obs"), t
        assert _ok(dtype, rew.cpu().numpy()[sure], g["rew"][t][sure], "simple_predator_prey single step:rew", 1.0, (1e-9, 2e-12)), t
        assert np.array_equal(ii[sure, :A], g["collisions"][t][sure]) and np.array_equal(ii[:, A], _flags(alive, holding))
        assert np.array_equal(env.aux[0].cpu().numpy()[sure], _flags(g["alive"][t], g["holding"][t])[sure]), t
        assert not done.any()
        # food: eaten ones are parked at -999, untouched ones keep their place, respawned ones land in the arena
        food = gp[:, A + 1:A + 4]
        was_alive = alive
        now_alive = ((env.aux[0].cpu().numpy()[:, None] >> np.arange(3)) & 1).astype(bool)
        assert np.array_equal(food[was_alive & now_alive], pos[:, A + 1:A + 4][was_alive & now_alive].astype(food.dtype))
        assert np.all(food[~now_alive] == -999.0)
        assert np.all(np.abs(food[~was_alive]) <= 0.95) and now_alive[~was_alive].all()
        # the respawn stream advanced by the number of respawns
        assert np.array_equal(env.aux[2].cpu().numpy(), (~was_alive).sum(axis=1))
    assert sure_total > 0.8 * T * B, (sure_total, T * B)


def test_free_running_until_first_respawn_matches_reference():
    """Envs of the fixture replayed free-running in float64 for as long as no food has respawned."""
    g = load_golden(CASE)
    T, B, A = g["actions"].shape
    env = _env(B, torch.float64)
    env.set_state(g["pos0"], g["vel0"], goal=_aux(g["alive0"], g["holding0"]))
    clean = g["alive0"].all(axis=1)
    steps_checked = 0
    for t in range(T):
        obs, rew, done, _ = env.step(torch.as_tensor(g["actions"][t]))
        gp, gv, _ = env.get_state()
        assert _close(gp[clean][:, :A], g["pos"][t][clean], 1e-9, 1e-12), t
        assert _close(obs.cpu().numpy()[clean], g["obs"][t][clean], 1e-9, 1e-12), t
        assert _close(rew.cpu().numpy()[clean], g["rew"][t][clean], 1e-9, 1e-12), t
        assert np.array_equal(env.aux[0].cpu().numpy()[clean], _flags(g["alive"][t], g["holding"][t])[clean])
        steps_checked += int(clean.sum())
        clean &= g["alive"][t].all(axis=1)        # a food eaten at step t respawns somewhere else at t + 1
    assert steps_checked > 300 and clean.sum() >= 4


def test_device_reset_matches_the_oracle_stream():
    """mpe_reset: Philox draws value for value, foods clear of barrier and forests, fresh flags."""
    spec = orc.make_spec(CASE)
    B = 5000
    env = _env(B, torch.float64)
    env._episode = 3
    env.reset(seed=77, observe=False)
    pos, vel, _ = env.get_state()
    want, key = orc.pp_device_reset(spec, 77, 3, B)
    assert np.array_equal(pos, want)
    aux = env.aux.cpu().numpy()
    assert np.all(aux[0] == 7) and np.array_equal(aux[1], key) and not aux[2].any() and not vel.any()
    A = spec.A
    redrawn = (want[:, A + 1:A + 4] != np.stack([orc.device_reset_draw(77, np.arange(B, dtype=np.uint64), 3, A + f, 0.9) for f in spec.food], axis=1)).any(axis=(1, 2))
    assert 0.2 < redrawn.mean() < 0.7          # barrier + 2 forests cover a good part of the arena
    for f in spec.food:
        for o in (A, A + 4, A + 5):
            assert not (orc._dist(pos[:, A + f] - pos[:, o]) < spec.size[A + f] + spec.size[o]).any()


@pytest.mark.parametrize("mode", ["step", "rollout", "episode"])
def test_free_running_episode_against_oracle(mode):
    """Device reset, 25 free-running float64 steps with pick-ups, deposits and respawns; the oracle follows
    with the device's own respawn stream."""
    spec = orc.make_spec(CASE)
    A = spec.A
    B, T = 3000, 25
    env = _env(B, torch.float64)
    rng = np.random.RandomState(3)
    acts = rng.randint(0, 5, (T, B, A)).astype(np.int32)
    if mode == "episode":
        out = env.rollout(torch.as_tensor(acts, device="cuda:0"), reset_seed=9)
        pos, key = orc.pp_device_reset(spec, 9, 0, B)
    else:
        env.reset(seed=9, observe=False)
        pos = env.get_state()[0]
        key = env.aux[1].cpu().numpy()
        # start half of the worlds mid-game: members next to food / holding next to their leader
        pos[::2, 4] = pos[::2, A + 1] + 0.01
        pos[::4, 5] = pos[::4, 3] + 0.02
        hold0 = np.zeros((B, 3), bool)
        hold0[::4, 1] = True
        env.set_state(pos, np.zeros((B, A, 2)), goal=_aux(np.ones((B, 3), bool), hold0, key=key))
        if mode == "rollout":
            out = env.rollout(torch.as_tensor(acts, device="cuda:0"))
    vel = np.zeros((B, A, 2))
    alive = np.ones((B, 3), bool)
    holding = np.zeros((B, 3), bool)
    if mode != "episode":
        holding[::4, 1] = True
    count = np.zeros(B, np.int64)
    bidx = np.arange(B, dtype=np.uint64)
    n_pick = n_respawn = 0
    for t in range(T):
        # where each dead food will respawn: the stream is consumed in food order
        resp = np.zeros((B, 3, 2))
        c = count.copy()
        for f in range(3):
            dead = ~alive[:, f]
            resp[dead, f] = orc.device_respawn_draw(key[dead], bidx[dead], c[dead])
            c += dead
        n_respawn += int((~alive).sum())
        count = c
        before = alive.copy()
        pos, vel, obs, rew, alive, holding = orc.pp_env_step(spec, pos, vel, acts[t], alive, holding, resp)
        n_pick += int((before & ~alive).sum())
        if mode == "step":
            g_obs, g_rew, _, (ii, _) = env.step(torch.as_tensor(acts[t]))
            g_obs, g_rew = g_obs.cpu().numpy(), g_rew.cpu().numpy()
            gp, gv, _ = env.get_state()
            assert _close(gp, pos, 1e-9, 1e-12) and _close(gv, vel, 1e-9, 1e-12), t
            assert np.array_equal(env.aux[0].cpu().numpy(), _flags(alive, holding)) and np.array_equal(env.aux[2].cpu().numpy(), count)
            assert np.array_equal(ii.cpu().numpy()[:, :A], orc.pp_info(spec, pos)["collisions"])      # agents do not move in post_step
        else:
            g_obs, g_rew = out["obs"][t].cpu().numpy(), out["rew"][t].cpu().numpy()
        assert _close(g_obs, np.concatenate(obs, axis=1), 1e-9, 1e-12), (mode, t)
        assert _close(g_rew, rew, 1e-9, 1e-12), (mode, t)
    gp, gv, _ = env.get_state()
    assert _close(gp, pos, 1e-9, 1e-12) and _close(gv, vel, 1e-9, 1e-12)
    assert np.array_equal(env.aux[0].cpu().numpy(), _flags(alive, holding)) and np.array_equal(env.aux[2].cpu().numpy(), count)
    assert n_pick > (50 if mode == "episode" else 500) and n_respawn == n_pick - int((~alive).sum())


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_rollout_and_episode_equal_stepping(dtype):
    dt = getattr(torch, dtype)
    B, T = 1000, 25
    e1, e2, e3 = _env(B, dt), _env(B, dt), _env(B, dt)
    a = torch.randint(0, 5, (T, B, 7), dtype=torch.int32, device="cuda:0", generator=torch.Generator("cuda:0").manual_seed(5))
    e1.reset(seed=11, observe=False)
    e2.reset(seed=11, observe=False)
    out2 = e2.rollout(a)
    out3 = e3.rollout(a, reset_seed=11)
    for t in range(T):
        obs, rew, done, (ii, _) = e1.step(a[t])
        assert torch.equal(obs, out2["obs"][t]) and torch.equal(rew, out2["rew"][t]) and torch.equal(ii, out2["info_i"][t]), t
    for k in ("obs", "rew", "done", "info_i"):
        assert torch.equal(out2[k], out3[k]), k
    assert torch.equal(e1.pos, e2.pos) and torch.equal(e1.pos, e3.pos) and torch.equal(e1.vel, e3.vel) and torch.equal(e1.aux, e3.aux)
    assert torch.equal(e1.aux, e2.aux)


def test_step_auto_restarts_flags_and_stream():
    B, H = 400, 6
    env = _env(B, torch.float64, max_episode_steps=H, seed=21)
    ref = _env(B, torch.float64)
    env.reset(seed=21, observe=False)
    pos = env.get_state()[0]
    pos[:, 4] = pos[:, 8] + 0.01            # every world's first member starts on the first food
    env.set_state(pos, np.zeros((B, 7, 2)), goal=np.stack([np.full(B, 7), env.aux[1].cpu().numpy(), np.zeros(B, np.int64)], axis=1))
    a = torch.zeros(B, 7, dtype=torch.int32, device="cuda:0")
    for t in range(H):
        obs, rew, done, _ = env.step(a)
        if t == 0:
            eaten = (env.aux[0] & 1) == 0                   # food 0 eaten, unless a contact pushed the member off it
            assert eaten.float().mean() > 0.9
        if t == 1:
            assert (env.aux[0] & 1)[eaten].all() and (env.aux[2][eaten] >= 1).all()       # respawned from the stream
    assert done.all()
    ref._episode = 1
    ref.reset(seed=21)
    assert torch.equal(env.aux, ref.aux) and torch.equal(env.pos, ref.pos) and torch.equal(obs, ref.obs)


@pytest.mark.parametrize("fused", [True, False])
def test_plugin_env_with_post_step_replays_the_reference(fused):
    """The single-world plugin through MultiAgentEnv with post_step as post_step_callback (its rules run in the
    kernels only).  Without a respawn in the replayed env the trajectory is the reference's."""
    import mpe_hip
    from mpe_hip.scenarios import load
    g = load_golden(CASE)
    sc = load(CASE + ".py").Scenario()
    world = sc.make_world()
    # the scenario's own post_step as callback: one fused mpe_step launch per step; wrapped (as a caller with its
    # own hook would): World.step + mpe_observe + the mpe_post_step kernel.  Either way no rule runs on the host.
    post = sc.post_step if fused else (lambda w, rng: sc.post_step(w, rng))
    env = mpe_hip.MultiAgentEnv(world, sc.reset_world, sc.reward, sc.observation, sc.benchmark_data, post_step_callback=post)
    assert (env._scenario is sc) == fused and [s.shape[0] for s in env.observation_space] == list(g["obs_dims"])
    env.seed(0)
    env.reset()
    b = 9       # member 1 holds food next to its leader and hands it over; no food is eaten
    assert g["alive"][:, b].all()
    A = env.n
    for i, e in enumerate(world.entities):
        e.state.p_pos = g["pos0"][b, i].copy()
        if i < A:
            e.state.p_vel = g["vel0"][b, i].copy()
    for k, m in enumerate(world.agents[4:]):
        m.holding = bool(g["holding0"][b, k])
    world.calculate_distances()
    off = np.concatenate([[0], np.cumsum(g["obs_dims"])])
    for t in range(25):
        obs_n, rew_n, done, info = env.step([int(x) for x in g["actions"][t, b]])
        for i in range(A):
            assert obs_n[i].dtype == np.float32 and _close(obs_n[i], g["obs"][t, b, off[i]:off[i + 1]], 1e-9, 1e-12), (t, i)
        assert _close(rew_n, g["rew"][t, b], 1e-9, 1e-12) and done is False
        assert info["n"] == list(g["collisions"][t, b])
        assert [m.holding for m in world.agents[4:]] == list(g["holding"][t, b])
    assert g["holding0"][b].any() and not g["holding"][-1, b].any()


@pytest.mark.parametrize("B", [1, 3, 64, 65, 128, 130, 132, 200])
def test_ragged_batches_rollout_equals_stepping(B):
    """Full tiles (the two-waves-per-SIMD instantiation), ragged tails and batches that are no multiple of 4 (every
    wave guarded) in one sweep: an episode launch with its reset observation against mpe_reset + single steps."""
    T = 9
    e1, e2 = _env(B, torch.float32, info=False), _env(B, torch.float32, info=False)
    a = torch.randint(0, 5, (T, B, 7), dtype=torch.int32, device="cuda:0", generator=torch.Generator("cuda:0").manual_seed(B))
    obs0 = e1.reset(seed=9)
    out = e2.rollout(a, reset_seed=9, obs0=True)
    assert torch.equal(out["obs0"].reshape(obs0.shape), obs0)
    for t in range(T):
        obs, rew, done, _ = e1.step(a[t])
        assert torch.equal(obs, out["obs"][t]) and torch.equal(rew, out["rew"][t]) and torch.equal(done, out["done"][t]), t
    for k in ("pos", "vel", "aux"):
        assert torch.equal(getattr(e1, k), getattr(e2, k)), k
