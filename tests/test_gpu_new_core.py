"""World.step of multiagent/new_core.py on the GPU (mpe_world_step with MPE_PHYSICS_NEW_CORE) against
the fixture captured from the reference's own new_core.World.step, and the new_core object facade."""
import numpy as np
import pytest

import mpe_oracle as orc
from conftest import load_golden, within_fp32_bar
from test_oracle_golden import new_core_spec

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= atol + rtol * np.abs(b))


def _walls(g):
    return [dict(orient="V" if v else "H", axis_pos=float(a), endpoints=(float(e[0]), float(e[1])), width=float(w), hard=bool(h))
            for v, a, e, w, h in zip(g["wall_vertical"], g["wall_axis_pos"], g["wall_endpoints"], g["wall_width"], g["wall_hard"])]


def _env(g, B, dtype):
    import mpe_hip
    A = g["accel"].shape[0]
    L = g["size"].shape[0] - A
    spec = mpe_hip.physics_spec(A, L, 3, g["size"], g["mass"], g["collide"], np.where(np.isnan(g["accel"]), 5.0, g["accel"]), g["max_speed"],
                                np.ones(A, bool), walls=_walls(g), ghost=g["ghost"], movable=g["movable"], new_core=True, accel=g["accel"])
    return mpe_hip.BatchedEnv(spec, B, dtype=dtype, device="cuda:0")


def _u(g, t):
    """agent.action.u as _set_action leaves it: unit move times (accel or 5.0)."""
    a = g["actions"][t]
    sens = np.where(np.isnan(g["accel"]), 5.0, g["accel"])
    u = np.zeros(a.shape + (2,))
    u[..., 0] = np.where(a == 1, -1.0, np.where(a == 2, 1.0, 0.0))
    u[..., 1] = np.where(a == 3, -1.0, np.where(a == 4, 1.0, 0.0))
    return u * sens[None, :, None]


def test_fp64_new_core_step_matches_reference():
    g = load_golden("new_core")
    T, B, A = g["actions"].shape
    env = _env(g, B, torch.float64)
    env.set_state(g["pos0"], g["vel0"])
    for t in range(T):
        env.world_step(_u(g, t).reshape(B, 2 * A))
        pos, vel, _ = env.get_state()
        assert _close(pos[:, :A], g["pos"][t], 1e-9, 1e-12), (t, np.abs(pos[:, :A] - g["pos"][t]).max())
        assert _close(vel, g["vel"][t], 1e-9, 1e-12), t
    assert np.array_equal(pos[:, 6], g["pos0"][:, 6])        # the immovable agent stayed put


def test_fp32_new_core_single_steps_within_1e4():
    """Teacher-forced from the reference's states (contacts between a heavy and a light agent are
    stiff: free-running float32 drifts past 1e-4 in a few of these envs)."""
    g = load_golden("new_core")
    T, B, A = g["actions"].shape
    env = _env(g, B, torch.float32)
    L = g["size"].shape[0] - A
    for t in range(T):
        p0 = g["pos0"].copy()
        v0 = g["vel0"]
        if t > 0:
            p0[:, :A] = g["pos"][t - 1]
            v0 = g["vel"][t - 1]
        env.set_state(p0, v0)
        env.world_step(_u(g, t).reshape(B, 2 * A))
        pos, vel, _ = env.get_state()
        assert within_fp32_bar(pos[:, :A], g["pos"][t], "new_core single step:pos"), (t, np.abs(pos[:, :A] - g["pos"][t]).max())
        # a step's velocity change is force * dt / mass, its position change that times dt again: a teacher-forced
        # position within the bar bounds the velocity only to 1 / dt = 10 times it
        assert within_fp32_bar(vel, g["vel"][t], "new_core single step:vel", 10.0), t


def test_core_handle_unchanged_by_the_new_fields():
    """The same world with core.py physics differs from new_core (masses matter) and matches the oracle."""
    g = load_golden("new_core")
    T, B, A = g["actions"].shape
    import mpe_hip
    L = g["size"].shape[0] - A
    spec = mpe_hip.physics_spec(A, L, 3, g["size"], g["mass"], g["collide"], np.where(np.isnan(g["accel"]), 5.0, g["accel"]), g["max_speed"],
                                np.ones(A, bool), walls=_walls(g), ghost=g["ghost"], movable=g["movable"])
    env = mpe_hip.BatchedEnv(spec, B, dtype=torch.float64, device="cuda:0")
    ospec = new_core_spec(g)
    ospec.new_core = False
    env.set_state(g["pos0"], g["vel0"])
    pos, vel = g["pos0"].copy(), g["vel0"].copy()
    for t in range(5):
        u = _u(g, t)
        env.world_step(u.reshape(B, 2 * A))
        pos, vel, _ = orc.world_step(ospec, pos, vel, u, np.zeros((B, A, 3)))
        gp, gv, _ = env.get_state()
        assert _close(gp, pos, 1e-9, 1e-12) and _close(gv, vel, 1e-9, 1e-12), t
    assert np.abs(gp[:, :A] - g["pos"][4]).max() > 1e-3


def test_new_core_needs_accel_and_the_physics_only_handle():
    import mpe_hip
    spec = mpe_hip.make_spec("simple_tag")
    spec.new_core = True
    with pytest.raises(mpe_hip.MpeError):
        mpe_hip.BatchedEnv(spec, 8)


def test_new_core_world_facade_replays_the_reference():
    """multiagent.new_core.World: GPU step, sliding history and the distance cache around it."""
    from multiagent.new_core import Agent, Landmark, Wall, World
    g = load_golden("new_core")
    T, B, A = g["actions"].shape
    L = g["size"].shape[0] - A
    b = 3
    w = World()
    w.dim_c = 3
    w.cache_dists = True
    for i in range(A):
        a = Agent()
        a.i, a.silent, a.collide, a.size, a.initial_mass = i, True, bool(g["collide"][i]), float(g["size"][i]), float(g["mass"][i])
        a.max_speed, a.accel = float(g["max_speed"][i]), (None if np.isnan(g["accel"][i]) else float(g["accel"][i]))
        a.ghost, a.movable = bool(g["ghost"][i]), bool(g["movable"][i])
        a.state.p_pos, a.state.p_vel, a.state.c = g["pos0"][b, i].copy(), g["vel0"][b, i].copy(), np.zeros(3)
        a.state.p_his_pos, a.state.p_his_vel = np.tile(a.state.p_pos, [10, 1]), np.tile(a.state.p_vel, [10, 1])
        a.action.c = np.zeros(3)
        w.agents.append(a)
    for l in range(L):
        lm = Landmark()
        lm.i, lm.collide, lm.movable, lm.size = A + l, bool(g["collide"][A + l]), False, float(g["size"][A + l])
        lm.state.p_pos, lm.state.p_vel = g["pos0"][b, A + l].copy(), np.zeros(2)
        w.landmarks.append(lm)
    w.walls = [Wall(orient=d["orient"], axis_pos=d["axis_pos"], endpoints=d["endpoints"], width=d["width"], hard=d["hard"]) for d in _walls(g)]
    w.calculate_distances()
    for t in range(12):
        u = _u(g, t)[b]
        for i, a in enumerate(w.agents):
            a.action.u = u[i]
        w.step()
        got = np.array([a.state.p_pos for a in w.agents])
        assert _close(got, g["pos"][t, b], 1e-9, 1e-12), t
        assert _close([a.state.p_vel for a in w.agents], g["vel"][t, b], 1e-9, 1e-12)
    # history rows are the last ten states, oldest first; the immovable agent's history never slides
    hist = w.agents[0].state.p_his_pos
    assert hist.shape == (10, 2) and _close(hist, g["pos"][2:12, b, 0], 1e-9, 1e-12)
    assert np.array_equal(w.agents[6].state.p_his_pos, np.tile(g["pos0"][b, 6], [10, 1]))
    # distance cache of the final state
    pos = np.array([e.state.p_pos for e in w.entities])
    assert np.array_equal(w.cached_dist_vect[2, 5], pos[2] - pos[5]) and np.array_equal(w.cached_dist_vect[5, 2], -(pos[2] - pos[5]))
    assert np.array_equal(w.cached_dist_mag[1, 8], np.sqrt(np.sum(np.square(pos[1] - pos[8]))))
    assert w.cached_collisions.shape == (A + L, A + L)


@pytest.mark.parametrize("seed", range(12))
def test_world_step_on_random_worlds_matches_oracle(seed):
    """mpe_world_step on worlds nobody designed: 1..64 agents and 0..64 landmarks with random sizes, masses, collide /
    movable / silent / ghost flags, speed limits, up to four random walls (hard and soft, entities near their rounded
    ends), random dt and damping, under the physics of core.py (even seeds) and new_core.py with random accel (odd
    seeds).  Every step starts from the oracle's state; the float64 kernel must stay within 1e-11, the float32 kernel within 1e-4
    (both relative to 1 + the largest speed in the batch)."""
    import mpe_hip
    rng = np.random.RandomState(4000 + seed)
    A, L = int(rng.randint(1, 65)), int(rng.randint(0, 65))
    if seed == 0:
        A, L = 64, 64
    if seed == 1:
        A, L = 1, 0
    E = A + L
    new_core = seed % 2 == 1
    size = rng.uniform(0.02, 0.2, E)
    mass = rng.uniform(0.5, 3.0, E)
    collide = rng.rand(E) < 0.6
    movable = np.r_[rng.rand(A) < 0.85, np.zeros(L, bool)]
    sens = rng.uniform(1.0, 5.0, A)
    max_speed = np.where(rng.rand(A) < 0.5, np.inf, rng.uniform(0.3, 1.5, A))
    silent = rng.rand(A) < 0.8
    dim_c = int(rng.randint(1, 5))
    ghost = rng.rand(A) < 0.3
    accel = np.where(rng.rand(A) < 0.3, np.nan, rng.uniform(1.0, 4.0, A)) if new_core else None
    walls = []
    for _ in range(int(rng.randint(0, 5))):
        lo, hi = np.sort(rng.uniform(-1.0, 1.0, 2))
        walls.append(dict(orient="HV"[int(rng.randint(2))], axis_pos=float(rng.uniform(-1, 1)), endpoints=(float(lo), float(hi)),
                          width=float(rng.uniform(0.05, 0.3)), hard=bool(rng.rand() < 0.5)))
    dt, damping = float(rng.uniform(0.05, 0.15)), float(rng.uniform(0.1, 0.4))
    spec = orc.Spec("physics", A, L, dim_c, size, movable, collide, mass, sens=sens, max_speed=max_speed, silent=silent,
                    adversary=np.zeros(A, bool), leader=np.zeros(A, bool), dt=dt, damping=damping, walls=walls, ghost=ghost,
                    new_core=new_core, accel=accel)
    kw = dict(new_core=True, accel=accel) if new_core else {}
    phys = mpe_hip.physics_spec(A, L, dim_c, size, mass, collide, sens, max_speed, silent, walls=walls, ghost=ghost, movable=movable,
                                dt=dt, damping=damping, **kw)
    B = 40
    env = mpe_hip.BatchedEnv(phys, B, dtype=torch.float64)
    env32 = mpe_hip.BatchedEnv(phys, B, dtype=torch.float32)
    worst32 = 0.0
    speakers = [i for i in range(A) if not silent[i]]
    pos = rng.uniform(-0.7, 0.7, (B, E, 2))         # dense: many contacts, entities beside and beyond the walls' ends
    vel = rng.uniform(-1, 1, (B, A, 2))
    worst = 0.0
    for t in range(4):
        u = rng.uniform(-1, 1, (B, A, 2)) * sens[None, :, None]
        c = rng.uniform(0, 1, (B, A, dim_c))
        env.set_state(pos, vel)
        env.world_step(u.reshape(B, -1), c[:, speakers, :].reshape(B, -1) if speakers else None)
        env32.set_state(pos, vel)
        env32.world_step(u.reshape(B, -1), c[:, speakers, :].reshape(B, -1) if speakers else None)
        pos, vel, comm = orc.world_step(spec, pos, vel, u, c)
        p32, v32, _ = env32.get_state()
        worst32 = max(worst32, np.abs(p32 - pos).max() / (1.0 + np.abs(vel).max()), np.abs(v32 - vel).max() / (1.0 + np.abs(vel).max()))
        g_pos, g_vel, g_comm = env.get_state()
        scale = 1.0 + np.abs(vel).max()
        worst = max(worst, np.abs(g_pos - pos).max() / scale, np.abs(g_vel - vel).max() / scale)
        if speakers:
            assert np.array_equal(g_comm, comm[:, speakers, :].reshape(B, -1))
        assert np.array_equal(g_pos[:, ~movable], pos[:, ~movable])         # what cannot move stays bit for bit
    assert worst < 1e-11, (A, L, new_core, len(walls), worst)
    assert worst32 < 1e-4, (A, L, new_core, len(walls), worst32)        # the float32 kernel, same single steps
