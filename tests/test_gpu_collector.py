"""collector (new_core physics with mass-ratio contacts, sorted-nearest observations, pooled reward,
pick-up / hand-over / respawn state machine) on the GPU.

As for simple_predator_prey the reference draws respawns from a numpy generator and the kernel from a
counter-based stream: the fixture is replayed teacher-forced and free-running until a world's first
respawn; whole free-running episodes are checked against the oracle evaluating the device's stream.
Bars as in tests/test_gpu_parity.py: float64 1e-9 relative (1e-12 floor), float32 1e-4; state bits and
benchmark counts bit-exact.
"""
import numpy as np
import pytest

import mpe_oracle as orc
from conftest import decided_with_margin, load_golden, within_fp32_bar

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
CASE = "collector"


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


def _word(holding, ttype, alive):
    h = ((holding + 1).astype(np.int64) << (2 * np.arange(6))).sum(axis=1)
    return h | (ttype.astype(np.int64) << (12 + np.arange(6))).sum(axis=1) | (alive.astype(np.int64) << (18 + np.arange(6))).sum(axis=1)


def _aux(holding, ttype, alive, key=None, count=None):
    B = holding.shape[0]
    return np.stack([_word(holding, ttype, alive), np.zeros(B, np.int64) if key is None else key, np.zeros(B, np.int64) if count is None else count], axis=1)


def _unword(w):
    w = np.asarray(w).astype(np.int64)
    return ((w[:, None] >> (2 * np.arange(6))) & 3) - 1, (w[:, None] >> (12 + np.arange(6))) & 1, ((w[:, None] >> (18 + np.arange(6))) & 1).astype(bool)


def _state_before(g, t):
    A = g["pos"].shape[2]
    pos = g["pos0"].copy()
    if t == 0:
        return pos, g["vel0"], g["holding0"], g["ttype0"], g["alive0"]
    pos[:, :A] = g["pos"][t - 1]
    pos[:, A:] = g["treasure_pos"][t - 1]
    return pos, g["vel"][t - 1], g["holding"][t - 1], g["ttype"][t - 1], g["alive"][t - 1]


def test_layout():
    g = load_golden(CASE)
    env = _env(8, torch.float64)
    assert env.obs_dims == list(g["obs_dims"]) and env.n_act == 8 and env.layout.n_aux == 3 and env.layout.n_info_i == 25


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_teacher_forced_steps_match_reference(dtype):
    g = load_golden(CASE)
    T, B, A = g["actions"].shape
    env = _env(B, getattr(torch, dtype))
    pos, vel, holding, ttype, alive = _state_before(g, 0)
    env.set_state(pos, vel, goal=_aux(holding, ttype, alive))
    env.observe()
    exact = dtype == "float64"
    if exact:
        assert _close(env.obs.cpu().numpy(), g["obs0"], 1e-9, 1e-12)
    spec = orc.make_spec(CASE)
    L = spec.L
    # nearest-first lists: every agent orders the other agents and the treasures by distance (collector.py:268-295)
    ordered = [(i, [j for j in range(A) if j != i]) for i in range(A)] + [(i, list(range(A, A + L))) for i in range(A)]
    sure_total = 0
    for t in range(T):
        pos, vel, holding, ttype, alive = _state_before(g, t)
        env.set_state(pos, vel, goal=_aux(holding, ttype, alive))
        obs, rew, done, (info_i, _) = env.step(torch.as_tensor(g["actions"][t]))
        gp, gv, _ = env.get_state()
        assert _ok(dtype, gp[:, :A], g["pos"][t], "collector single step:pos"), (t, np.abs(gp[:, :A] - g["pos"][t]).max())
        # teacher-forced: a position within the bar bounds the velocity (position change / dt) to 10 times it
        assert _ok(dtype, gv, g["vel"][t], "collector single step:vel", 10.0, (1e-8, 1e-11)), t
        ii = info_i.cpu().numpy()
        h2, t2, a2 = _unword(env.aux[0].cpu().numpy())
        # float32 may reorder two near-equidistant neighbours or flip a knife-edge contact of the fixture: the float64
        # kernel must agree everywhere, the float32 kernel in every env whose contacts and nearest-first orders the
        # reference's own positions decide with a 4e-4 margin (the position tolerance is 1e-4)
        after = pos.copy()
        after[:, :A] = g["pos"][t]
        sure = np.ones(B, bool) if exact else decided_with_margin(spec.size, after, 4e-4, ordered)
        sure_total += int(sure.sum())
        assert _ok(dtype, obs.cpu().numpy()[sure], g["obs"][t][sure], "collector single step:obs"), (t, np.abs(obs.cpu().numpy() - g["obs"][t])[sure].max())
        assert _ok(dtype, rew.cpu().numpy()[sure], g["rew"][t][sure], "collector single step:rew", 1.0, (1e-9, 2e-12)), (t, np.abs(rew.cpu().numpy() - g["rew"][t])[sure].max())
        assert np.array_equal(ii[sure, :3 * A].reshape(-1, A, 3), g["info"][t][sure]) and np.array_equal(ii[:, 3 * A], _word(holding, ttype, alive))
        assert np.array_equal(h2[sure], g["holding"][t][sure]) and np.array_equal(a2[sure], g["alive"][t][sure]), t
        keep = alive & a2 & sure[:, None]
        assert np.array_equal(t2[keep], g["ttype"][t][keep])
        assert not done.any()
        tre = gp[:, A:]
        assert np.array_equal(tre[alive & a2], pos[:, A:][alive & a2].astype(tre.dtype))
        assert np.all(tre[~a2] == 0.0)                                  # taken: parked at the origin
        assert np.all(np.abs(tre[~alive]) <= 0.95) and a2[~alive].all()  # respawned inside the arena
        assert np.array_equal(env.aux[2].cpu().numpy(), (~alive).sum(axis=1))
    assert sure_total > 0.7 * T * B, (sure_total, T * B)


def test_free_running_until_first_respawn_matches_reference():
    g = load_golden(CASE)
    T, B, A = g["actions"].shape
    env = _env(B, torch.float64)
    env.set_state(g["pos0"], g["vel0"], goal=_aux(g["holding0"], g["ttype0"], g["alive0"]))
    clean = g["alive0"].all(axis=1)
    checked = 0
    for t in range(T):
        obs, rew, done, _ = env.step(torch.as_tensor(g["actions"][t]))
        gp, gv, _ = env.get_state()
        assert _close(gp[clean][:, :A], g["pos"][t][clean], 1e-9, 1e-12), t
        assert _close(obs.cpu().numpy()[clean], g["obs"][t][clean], 1e-9, 1e-12), t
        assert _close(rew.cpu().numpy()[clean], g["rew"][t][clean], 1e-9, 1e-12), t
        h2, _, a2 = _unword(env.aux[0].cpu().numpy())
        assert np.array_equal(h2[clean], g["holding"][t][clean]) and np.array_equal(a2[clean], g["alive"][t][clean])
        checked += int(clean.sum())
        clean &= g["alive"][t].all(axis=1)
    assert checked > 300 and clean.sum() >= 4


def test_device_reset_matches_the_oracle_stream():
    spec = orc.make_spec(CASE)
    B = 4000
    env = _env(B, torch.float64)
    env._episode = 2
    env.reset(seed=31, observe=False)
    pos, vel, _ = env.get_state()
    want, ttype, key = orc.collector_device_reset(spec, 31, 2, B)
    assert np.array_equal(pos, want) and not vel.any()
    aux = env.aux.cpu().numpy()
    h, t, a = _unword(aux[0])
    assert (h == -1).all() and a.all() and np.array_equal(t, ttype) and 0.45 < t.mean() < 0.55
    assert np.array_equal(aux[1], key) and not aux[2].any()
    assert np.abs(pos[:, spec.A:]).max() <= 0.95 < np.abs(pos[:, :spec.A]).max()


@pytest.mark.parametrize("mode", ["step", "rollout", "episode"])
def test_free_running_episode_against_oracle(mode):
    spec = orc.make_spec(CASE)
    A, L = spec.A, spec.L
    B, T = 600, 25
    env = _env(B, torch.float64)
    rng = np.random.RandomState(3)
    acts = rng.randint(0, 5, (T, B, A)).astype(np.int32)
    holding = np.full((B, 6), -1, np.int64)
    if mode == "episode":
        out = env.rollout(torch.as_tensor(acts, device="cuda:0"), reset_seed=9)
        pos, ttype, key = orc.collector_device_reset(spec, 9, 0, B)
    else:
        env.reset(seed=9, observe=False)
        pos = env.get_state()[0]
        key = env.aux[1].cpu().numpy()
        _, ttype, _ = _unword(env.aux[0].cpu().numpy())
        # start many worlds mid-game: collectors on treasures, carriers next to their deposit
        pos[::2, 0] = pos[::2, A] + 0.01
        pos[::3, 1] = pos[::3, A + 1] - 0.02
        holding[::4, 2] = 1
        pos[::4, 2] = pos[::4, 7] + 0.03
        env.set_state(pos, np.zeros((B, A, 2)), goal=_aux(holding, ttype, np.ones((B, L), bool), key=key))
        if mode == "rollout":
            out = env.rollout(torch.as_tensor(acts, device="cuda:0"))
    vel = np.zeros((B, A, 2))
    alive = np.ones((B, L), bool)
    count = np.zeros(B, np.int64)
    bidx = np.arange(B, dtype=np.uint64)
    n_pick = n_hand = 0
    for t in range(T):
        resp, rtype = np.zeros((B, L, 2)), np.zeros((B, L), np.int64)
        c = count.copy()
        for l in range(L):                       # the stream is consumed in treasure order
            dead = ~alive[:, l]
            resp[dead, l] = orc.device_respawn_draw(key[dead], bidx[dead], c[dead])
            rtype[dead, l] = orc.device_respawn_type(key[dead], bidx[dead], c[dead])
            c += dead
        count = c
        before_alive, before_hold = alive.copy(), holding.copy()
        pos, vel, obs, rew, info, holding, ttype, alive = orc.collector_env_step(spec, pos, vel, acts[t], holding, ttype, alive, resp, rtype)
        n_pick += int((before_alive & ~alive).sum())
        n_hand += int(((before_hold >= 0) & (holding < 0)).sum())
        if mode == "step":
            g_obs, g_rew, _, (ii, _) = env.step(torch.as_tensor(acts[t]))
            g_obs, g_rew = g_obs.cpu().numpy(), g_rew.cpu().numpy()
            gp, gv, _ = env.get_state()
            assert _close(gp, pos, 1e-9, 1e-12) and _close(gv, vel, 1e-9, 1e-12), t
            assert np.array_equal(env.aux[0].cpu().numpy(), _word(holding, ttype, alive)) and np.array_equal(env.aux[2].cpu().numpy(), count), t
            assert np.array_equal(ii.cpu().numpy()[:, :3 * A].reshape(B, A, 3), info), t
        else:
            g_obs, g_rew = out["obs"][t].cpu().numpy(), out["rew"][t].cpu().numpy()
        assert _close(g_obs, obs, 1e-9, 1e-12), (mode, t, np.abs(g_obs - obs).max())
        assert _close(g_rew, rew, 1e-9, 1e-12), (mode, t, np.abs(g_rew - rew).max())
    gp, gv, _ = env.get_state()
    assert _close(gp, pos, 1e-9, 1e-12) and _close(gv, vel, 1e-9, 1e-12)
    assert np.array_equal(env.aux[0].cpu().numpy(), _word(holding, ttype, alive)) and np.array_equal(env.aux[2].cpu().numpy(), count)
    assert n_pick > (5 if mode == "episode" else 300), n_pick
    if mode != "episode":
        assert n_hand > 10, n_hand


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_rollout_and_episode_equal_stepping(dtype):
    dt = getattr(torch, dtype)
    B, T = 1000, 25
    e1, e2, e3 = _env(B, dt), _env(B, dt), _env(B, dt)
    a = torch.randint(0, 5, (T, B, 8), dtype=torch.int32, device="cuda:0", generator=torch.Generator("cuda:0").manual_seed(5))
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


@pytest.mark.parametrize("fused", [True, False])
def test_plugin_env_with_post_step_replays_the_reference(fused):
    """The single-world plugin through MultiAgentEnv with post_step as post_step_callback (its rules run in the
    kernels only), on an env of the fixture in which no treasure respawns."""
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
    b = [k for k in range(g["pos0"].shape[0]) if g["alive0"][k].all() and g["alive"][:, k].all() and (g["holding0"][k] >= 0).any()][0]
    A = env.n
    for i, e in enumerate(world.entities):
        e.state.p_pos = g["pos0"][b, i].copy()
        if i < A:
            e.state.p_vel = g["vel0"][b, i].copy()
    for k, a in enumerate(world.agents[:6]):
        a.holding = None if g["holding0"][b, k] < 0 else int(g["holding0"][b, k])
    for l, t in enumerate(world.landmarks):
        t.type, t.alive = int(g["ttype0"][b, l]), True
    world.calculate_distances()
    assert _close(np.concatenate([env._get_obs(a) for a in env.agents]), g["obs0"][b], 1e-9, 1e-12)
    off = np.concatenate([[0], np.cumsum(g["obs_dims"])])
    for t in range(25):
        obs_n, rew_n, done, info = env.step([int(x) for x in g["actions"][t, b]])
        for i in range(A):
            assert obs_n[i].dtype == np.float32 and _close(obs_n[i], g["obs"][t, b, off[i]:off[i + 1]], 1e-9, 1e-12), (t, i)
        assert _close(rew_n, g["rew"][t, b], 1e-9, 1e-12) and done is False
        assert [tuple(x) for x in info["n"]] == [tuple(x) for x in g["info"][t, b]]
        assert [-1 if a.holding is None else a.holding for a in world.agents[:6]] == list(g["holding"][t, b])


def _plugin_env(g, b, fused, seed):
    import mpe_hip
    from mpe_hip.scenarios import load
    sc = load(CASE + ".py").Scenario()
    world = sc.make_world()
    post = sc.post_step if fused else (lambda w, rng: sc.post_step(w, rng))
    env = mpe_hip.MultiAgentEnv(world, sc.reset_world, sc.reward, sc.observation, sc.benchmark_data, post_step_callback=post)
    env.seed(seed)
    env.reset()
    for i, e in enumerate(world.entities):
        e.state.p_pos = g["pos0"][b, i].copy()
        if i < env.n:
            e.state.p_vel = g["vel0"][b, i].copy()
    for k, a in enumerate(world.agents[:6]):
        a.holding = None if g["holding0"][b, k] < 0 else int(g["holding0"][b, k])
    for l, t in enumerate(world.landmarks):
        t.type, t.alive = int(g["ttype0"][b, l]), bool(g["alive0"][b, l])
    world.calculate_distances()
    return env, world


def test_plugin_env_respawns_agree_between_the_fused_step_and_the_post_step_kernel():
    """An env of the fixture in which treasures are taken and respawn: the fused step (post_step inside mpe_step) and
    the split step (mpe_world_step, mpe_observe, mpe_post_step) leave identical worlds, respawn values drawn from the
    env's generator in the reference's order; up to the first respawn both are the reference's trajectory."""
    g = load_golden(CASE)
    taken = [k for k in range(g["pos0"].shape[0]) if g["alive0"][k].all() and not g["alive"][:, k].all()]
    assert taken
    b = taken[0]
    first_dead = int(np.argmax(~g["alive"][:, b].all(axis=1)))        # the step whose post_step takes the first treasure
    envs = [_plugin_env(g, b, fused, seed=7) for fused in (True, False)]
    off = np.concatenate([[0], np.cumsum(g["obs_dims"])])
    respawned = False
    for t in range(25):
        outs = [env.step([int(x) for x in g["actions"][t, b]]) for env, _ in envs]
        (o1, r1, _, i1), (o2, r2, _, i2) = outs
        assert all(np.array_equal(x, y) for x, y in zip(o1, o2)) and r1 == r2 and i1["n"] == i2["n"], t
        w1, w2 = envs[0][1], envs[1][1]
        assert all(np.array_equal(e1.state.p_pos, e2.state.p_pos) for e1, e2 in zip(w1.entities, w2.entities)), t
        assert [a.holding for a in w1.agents[:6]] == [a.holding for a in w2.agents[:6]]
        assert [(x.alive, x.type) for x in w1.landmarks] == [(x.alive, x.type) for x in w2.landmarks]
        assert w1.respawn_stream == w2.respawn_stream
        if t <= first_dead + 1:       # the respawned treasure is first seen by the step after the next
            for i in range(envs[0][0].n):
                assert _close(o1[i], g["obs"][t, b, off[i]:off[i + 1]], 1e-9, 1e-12), (t, i)
        respawned = respawned or w1.respawn_stream[1] > 0
    assert respawned
