#!/usr/bin/env python3
"""Capture golden trajectories from the reference simulator.

Runs ONLY in the build container, where the reference tree is mounted at
/root/reference.  It imports the reference's own ``core.py``, ``environment.py``
and scenario files by path, replays seeded initial states and integer action
sequences through them for 25 steps, and writes small ``.npz`` fixtures to
``tests/golden/``.  Nothing from the reference is copied: the fixtures hold
inputs and the numbers the reference produced.

``gym`` is not installed here, so an inert stand-in supplying only container
classes (spaces, Env, registration, seeding) is registered before the import;
it performs no arithmetic of the simulated path.  ``seaborn`` (imported by
new_core.py for display colours only) is absent too and gets a one-function
stand-in that returns grey tuples.

Reference entry points used:
  * simple_spread, simple_tag: ``MultiAgentEnv.step`` (multiagent/environment.py:83-115)
    with ``benchmark_data`` as the info callback (make_env.py:40-41).
    simple_tag's one-argument ``reset_world`` (simple_tag.py:39) is wrapped
    because ``MultiAgentEnv.reset`` passes two (environment.py:119).
  * simple, simple_adversary, simple_push: ``MultiAgentEnv.step`` likewise; the goal landmark is
    the reference's own random draw, repeated until it gives the index the fixture asks for.
  * simple_world_comm: the leader's Tuple action cannot pass ``_set_action``
    (environment.py:159-171 only splits MultiDiscrete), so the leader's
    ``action.u`` / ``action.c`` are written with the semantics of
    environment.py:176-201 and ``World.step`` plus the scenario callbacks are
    called directly; the other five agents go through ``_set_action``.
  * simple_speaker_listener, simple_crypto: ``MultiAgentEnv.step`` (their agents take a single
    int each: an immovable speaker its comm word, the silent listener its move).
    simple_reference: both agents move and speak, so as for the simple_world_comm leader their
    ``action.u`` / ``action.c`` are written directly and ``World.step``, the callbacks and the
    reward sharing of environment.py:103-105 (``np.sum(reward_n)``) are called here.
  * simple_predator_prey (new_core physics): ``MultiAgentEnv.step`` with the scenario's ``post_step``
    as post_step_callback; it takes one argument (simple_predator_prey.py:83) where the env passes two
    (environment.py:108), so it is wrapped.  Food respawns come from the global numpy generator; the
    fixture records where each food went so that a replay can be fed the same positions.
  * simple_spread with 64 agents / 64 landmarks: the reference hard-codes 3/3
    (simple_spread.py:12-13); the world is assembled here from the reference's
    ``World/Agent/Landmark`` classes with the same property assignments
    (simple_spread.py:24-35) and stepped with the reference's functions.

Usage:  python tools/gen_golden.py [--check]
  --check additionally replays every fixture through oracle/mpe_oracle.py and
  reports the largest deviation (must be 0 for float64 state).
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
T_STEPS = 25


# ----------------------------------------------------------------------------
# inert gym stand-in + by-path loader
# ----------------------------------------------------------------------------
def _install_gym_stub():
    gym = types.ModuleType("gym")

    class Space:
        pass

    class Env:
        pass

    class Discrete(Space):
        def __init__(self, n):
            self.n = n

    class Box(Space):
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class Tuple(Space):
        def __init__(self, spaces):
            self.spaces = list(spaces)

    spaces = types.ModuleType("gym.spaces")
    spaces.Discrete, spaces.Box, spaces.Tuple, spaces.Space = Discrete, Box, Tuple, Space
    envs = types.ModuleType("gym.envs")
    registration = types.ModuleType("gym.envs.registration")

    class EnvSpec:
        def __init__(self, *a, **k):
            pass

    registration.EnvSpec = EnvSpec
    registration.register = lambda *a, **k: None
    envs.registration = registration
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = lambda seed=None: (np.random.RandomState(seed), seed)
    utils.seeding = seeding
    gym.Space, gym.Env, gym.spaces, gym.envs, gym.utils = Space, Env, spaces, envs, utils
    sns = types.ModuleType("seaborn")          # display colours only (new_core.py:181-182)
    sns.color_palette = lambda *a, n_colors=None, **k: [(0.5, 0.5, 0.5)] * (n_colors or (a[1] if len(a) > 1 else 6))
    sys.modules["seaborn"] = sns
    for name, mod in [("gym", gym), ("gym.spaces", spaces), ("gym.envs", envs),
                      ("gym.envs.registration", registration), ("gym.utils", utils),
                      ("gym.utils.seeding", seeding)]:
        sys.modules[name] = mod


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    _install_gym_stub()
    pkg = types.ModuleType("multiagent")
    pkg.__path__ = [os.path.join(REF, "multiagent")]   # skips __init__.py (imports real gym)
    sys.modules["multiagent"] = pkg
    ref = types.SimpleNamespace()
    ref.core = _load("multiagent.core", os.path.join(REF, "multiagent", "core.py"))
    ref.new_core = _load("multiagent.new_core", os.path.join(REF, "multiagent", "new_core.py"))
    _load("multiagent.scenario", os.path.join(REF, "multiagent", "scenario.py"))
    _load("multiagent.multi_discrete", os.path.join(REF, "multiagent", "multi_discrete.py"))
    ref.environment = _load("multiagent.environment", os.path.join(REF, "multiagent", "environment.py"))
    ref.scenarios = {}
    for n in ("simple_spread", "simple_tag", "simple_world_comm", "simple", "simple_adversary", "simple_push",
              "simple_speaker_listener", "simple_reference", "simple_crypto", "simple_predator_prey", "consensus", "collector"):
        ref.scenarios[n] = _load("ref_scn_" + n, os.path.join(REF, "multiagent", "scenarios", n + ".py"))
    return ref


# ----------------------------------------------------------------------------
# driving the reference
# ----------------------------------------------------------------------------
def make_env(ref, name):
    sc = ref.scenarios[name].Scenario()
    world = sc.make_world()
    if name == "simple_spread":
        reset_cb = sc.reset_world
    else:
        reset_cb = lambda w, rng: sc.reset_world(w)     # one-argument reset_world
    env = ref.environment.MultiAgentEnv(world, reset_cb, sc.reward, sc.observation, sc.benchmark_data)
    return sc, world, env


def make_spread_world(ref, A, L):
    """simple_spread world with A agents and L landmarks built from the
    reference classes (property assignments of simple_spread.py:24-35)."""
    core = ref.core
    sc = ref.scenarios["simple_spread"].Scenario()
    world = core.World()
    world.dim_c = 2
    world.agents = [core.Agent() for _ in range(A)]
    for i, ag in enumerate(world.agents):
        ag.name = "agent %d" % i
        ag.collide = True
        ag.silent = True
        ag.size = 0.15
    world.landmarks = [core.Landmark() for _ in range(L)]
    for i, lm in enumerate(world.landmarks):
        lm.name = "landmark %d" % i
        lm.collide = False
        lm.movable = False
    sc.reset_world(world, np.random.RandomState(0))
    env = ref.environment.MultiAgentEnv(world, sc.reset_world, sc.reward, sc.observation, sc.benchmark_data)
    return sc, world, env


def inject(world, pos, vel):
    A = len(world.agents)
    for i, ag in enumerate(world.agents):
        ag.state.p_pos = np.array(pos[i], dtype=np.float64)
        ag.state.p_vel = np.array(vel[i], dtype=np.float64)
        ag.state.c = np.zeros(world.dim_c)
    for l, lm in enumerate(world.landmarks):
        lm.state.p_pos = np.array(pos[A + l], dtype=np.float64)
        lm.state.p_vel = np.zeros(world.dim_p)


def run_env_api(sc, world, env, name, pos0, vel0, actions, obs_steps=None):
    """Replay through MultiAgentEnv.step.  actions [T, B, A] ints."""
    T, B, A = actions.shape
    E = pos0.shape[1]
    rec = _alloc(name, T, B, A, world, env, obs_steps)
    env.seed(0)
    for b in range(B):
        env.reset()
        inject(world, pos0[b], vel0[b])
        for t in range(T):
            obs_n, rew_n, done, info_n = env.step([int(a) for a in actions[t, b]])
            _record(rec, name, t, b, world, obs_n, rew_n, done, info_n["n"], obs_steps)
    return rec


def run_world_comm(sc, world, env, pos0, vel0, actions):
    """actions [T, B, 7]: [leader move, leader comm, move of agents 1..5]."""
    name = "simple_world_comm"
    T, B, _ = actions.shape
    A = len(world.agents)
    rec = _alloc(name, T, B, A, world, env, None)
    rec["in_forest"] = np.zeros((T, B, A, 2), np.uint8)
    rec["visible"] = np.zeros((T, B, A, A), np.uint8)
    for b in range(B):
        sc.reset_world(world)
        inject(world, pos0[b], vel0[b])
        for t in range(T):
            act = actions[t, b]
            leader = world.agents[0]
            # environment.py:176-181,192-195 (move) and :199-201 (comm one-hot)
            u = np.zeros(world.dim_p)
            m = int(act[0])
            if m == 1: u[0] = -1.0
            if m == 2: u[0] = +1.0
            if m == 3: u[1] = -1.0
            if m == 4: u[1] = +1.0
            u *= leader.accel
            leader.action.u = u
            c = np.zeros(world.dim_c)
            c[int(act[1])] = 1.0
            leader.action.c = c
            for i in range(1, A):
                env._set_action(int(act[1 + i]), world.agents[i], env.action_space[i])
            world.step()
            obs_n = [sc.observation(ag, world) for ag in world.agents]
            rew_n = [sc.reward(ag, world) for ag in world.agents]
            inf_n = [sc.benchmark_data(ag, world) for ag in world.agents]
            _record(rec, name, t, b, world, obs_n, rew_n, False, inf_n, None)
            rec["comm"][t, b] = world.agents[0].state.c
            # masks: forest membership through the reference's is_collision
            inf = np.array([[sc.is_collision(ag, f) for f in world.forests] for ag in world.agents])
            rec["in_forest"][t, b] = inf
            for i in range(A):
                for j in range(A):
                    if i == j:
                        continue
                    v = (inf[i, 0] and inf[j, 0]) or (inf[i, 1] and inf[j, 1]) or \
                        (not inf[i, 0] and not inf[j, 0] and not inf[i, 1] and not inf[j, 1]) or \
                        world.agents[i].leader
                    rec["visible"][t, b, i, j] = v
    return rec


def _alloc(name, T, B, A, world, env, obs_steps):
    dims = [s.shape[0] for s in env.observation_space]
    n_obs_t = T if obs_steps is None else len(obs_steps)
    rec = {
        "pos": np.zeros((T, B, A, 2)), "vel": np.zeros((T, B, A, 2)),
        "obs": np.zeros((n_obs_t, B, sum(dims))), "obs_dims": np.array(dims, np.int64),
        "rew": np.zeros((T, B, A)), "done": np.zeros((T, B), np.uint8),
        "collisions": np.zeros((T, B, A), np.int64),
    }
    if name == "simple_spread":
        rec.update({"info_rew": np.zeros((T, B, A)), "min_dists": np.zeros((T, B)),
                    "occupied": np.zeros((T, B), np.int64)})
    if name == "simple_world_comm":
        rec["comm"] = np.zeros((T, B, world.dim_c))
    return rec


def _record(rec, name, t, b, world, obs_n, rew_n, done, inf_n, obs_steps):
    for i, ag in enumerate(world.agents):
        rec["pos"][t, b, i] = ag.state.p_pos
        rec["vel"][t, b, i] = ag.state.p_vel
    if obs_steps is None:
        rec["obs"][t, b] = np.concatenate([np.asarray(o, np.float64) for o in obs_n])
    elif t in obs_steps:
        rec["obs"][obs_steps.index(t), b] = np.concatenate([np.asarray(o, np.float64) for o in obs_n])
    rec["rew"][t, b] = [float(r) for r in rew_n]
    rec["done"][t, b] = bool(done)
    if name == "simple_spread":
        for i, inf in enumerate(inf_n):
            rec["info_rew"][t, b, i] = inf[0]
            rec["collisions"][t, b, i] = inf[1]
        rec["min_dists"][t, b] = inf_n[0][2]
        rec["occupied"][t, b] = inf_n[0][3]
    else:
        rec["collisions"][t, b] = inf_n


# ----------------------------------------------------------------------------
# seeded initial states (inputs)
# ----------------------------------------------------------------------------
def init_spread3(rng, B=32):
    A, L = 3, 3
    pos = rng.uniform(-1, 1, (B, A + L, 2))
    vel = np.zeros((B, A, 2))
    # envs 16..23: two agents already overlapping / touching (dist_min = 0.3)
    for k, b in enumerate(range(16, 24)):
        d = [0.05, 0.15, 0.25, 0.29, 0.2999, 0.3001, 0.31, 0.35][k]
        ang = rng.uniform(0, 2 * np.pi)
        pos[b, 1] = pos[b, 0] + d * np.array([np.cos(ang), np.sin(ang)])
    # envs 24..27: an agent sitting on / near a landmark (occupied threshold 0.1)
    for k, b in enumerate(range(24, 28)):
        d = [0.0, 0.05, 0.0999, 0.1001][k]
        pos[b, A + 1] = pos[b, 2] + np.array([d, 0.0])
    # env 28: contact anchor -- agents at (0,0), (0.2,0), (5,5)
    pos[28, 0], pos[28, 1], pos[28, 2] = (0.0, 0.0), (0.2, 0.0), (5.0, 5.0)
    # envs 29..31: all three agents clustered, with initial velocity
    for b in range(29, 32):
        c = rng.uniform(-0.5, 0.5, 2)
        pos[b, :A] = c + rng.uniform(-0.2, 0.2, (A, 2))
        vel[b] = rng.uniform(-1, 1, (A, 2))
    actions = rng.randint(0, 5, (T_STEPS, B, A))
    actions[:, 28, :] = 0
    actions[:, 20, :] = 2                         # constant push
    actions[:, 21, :] = np.array([2, 1, 4])       # agents 0 and 1 driven together
    return pos, vel, actions


def init_tag(rng, B=32):
    A, L = 4, 2
    pos = np.concatenate([rng.uniform(-1, 1, (B, A, 2)), rng.uniform(-0.9, 0.9, (B, L, 2))], axis=1)
    vel = np.zeros((B, A, 2))
    actions = rng.randint(0, 5, (T_STEPS, B, A))
    # env 8: speed-clamp anchor, everybody pushes +x from rest, far apart
    pos[8, :, :] = [(-3, -3), (-3, 3), (3, -3), (3, 3), (8, 8), (-8, 8)]
    actions[:, 8, :] = 2
    # envs 9..15: good agent (index 3) near / beyond the arena edge -> bound() branches
    for k, b in enumerate(range(9, 16)):
        x = [0.85, 0.8999, 0.95, 0.9999, 1.05, 1.6, 2.4][k]
        pos[b, 3] = (x if k % 2 == 0 else -x, 0.3 * (-1) ** k * x)
        actions[:, b, 3] = 2 if k % 2 == 0 else 1          # keeps heading outwards
    # envs 16..21: adversary touching the good agent (dist_min = 0.125)
    for k, b in enumerate(range(16, 22)):
        d = [0.0, 0.06, 0.12, 0.1249, 0.1251, 0.2][k]
        if d == 0.0:
            d = 1e-3
        ang = rng.uniform(0, 2 * np.pi)
        pos[b, k % 3] = pos[b, 3] + d * np.array([np.cos(ang), np.sin(ang)])
    # envs 22..27: agent pressed against an obstacle (dist_min = 0.275 / 0.25)
    for k, b in enumerate(range(22, 28)):
        ag = k % A
        d = [0.1, 0.2, 0.27, 0.28, 0.24, 0.26][k]
        ang = rng.uniform(0, 2 * np.pi)
        pos[b, ag] = pos[b, A + (k % L)] + d * np.array([np.cos(ang), np.sin(ang)])
    # envs 28..31: moving start
    vel[28:32] = rng.uniform(-1.5, 1.5, (4, A, 2))
    return pos, vel, actions


def init_world_comm(rng, B=32):
    A, L = 6, 5
    pos = np.concatenate([rng.uniform(-1, 1, (B, A, 2)), rng.uniform(-0.9, 0.9, (B, L, 2))], axis=1)
    vel = np.zeros((B, A, 2))
    actions = np.zeros((T_STEPS, B, 7), np.int64)
    actions[:, :, 0] = rng.randint(0, 5, (T_STEPS, B))
    actions[:, :, 1] = (np.arange(T_STEPS)[:, None] + np.arange(B)[None, :]) % 4   # comm cycles 0..3
    actions[:, :, 2:] = rng.randint(0, 5, (T_STEPS, B, 5))
    f1, f2 = A + 3, A + 4
    food1 = A + 1

    def near(center, r):
        ang = rng.uniform(0, 2 * np.pi)
        return center + r * np.array([np.cos(ang), np.sin(ang)])

    # envs 8..23: forests kept apart, agents placed to hit every visibility branch
    for b in range(8, 24):
        pos[b, f1] = (-0.5, -0.4)
        pos[b, f2] = (0.5, 0.4)
        k = b - 8
        for i in range(A):
            where = (k + i * (1 + k // 4)) % 4
            if where == 0:
                pos[b, i] = near(pos[b, f1], rng.uniform(0, 0.25))
            elif where == 1:
                pos[b, i] = near(pos[b, f2], rng.uniform(0, 0.25))
            elif where == 2:
                pos[b, i] = near(pos[b, f1], 0.3 + 0.075 + rng.uniform(-0.02, 0.02))   # at the rim
            # where == 3: stays at its random position
    # envs 24..25: overlapping forests, agents inside both
    for b in (24, 25):
        pos[b, f1] = (0.0, 0.0)
        pos[b, f2] = (0.2, 0.0)
        for i in range(A):
            pos[b, i] = near(np.array([0.1, 0.0]), rng.uniform(0, 0.45))
    # envs 26..28: good agent on food (dist_min = 0.075) and near the arena edge
    for k, b in enumerate((26, 27, 28)):
        pos[b, 4] = pos[b, food1] + np.array([[0.0, 0.01], [0.0749, 0.0], [0.0751, 0.0]][k])
        pos[b, 5] = ([0.95, -0.2], [1.2, 0.97], [-0.93, -1.4])[k]
        actions[:, b, 6] = 0
    # envs 29..31: adversary touching a good agent (dist_min = 0.12) / the obstacle (0.275, 0.245)
    for k, b in enumerate((29, 30, 31)):
        pos[b, k] = near(pos[b, 4], [0.05, 0.1199, 0.1201][k])
        pos[b, 3] = near(pos[b, A + 0], [0.2, 0.27, 0.28][k])
        pos[b, 5] = near(pos[b, A + 0], [0.2, 0.24, 0.25][k])
    vel[30:32] = rng.uniform(-1, 1, (2, A, 2))
    return pos, vel, actions


def init_spread64(rng, B=2):
    A = L = 64
    pos = rng.uniform(-1, 1, (B, A + L, 2))
    vel = np.zeros((B, A, 2))
    actions = rng.randint(0, 5, (T_STEPS, B, A))
    return pos, vel, actions


def _move_u(m, accel):
    """environment.py:176-181,192-195 for one int move."""
    u = np.zeros(2)
    if m == 1: u[0] = -1.0
    if m == 2: u[0] = +1.0
    if m == 3: u[1] = -1.0
    if m == 4: u[1] = +1.0
    return u * (accel if accel is not None else 5.0)


# ----------------------------------------------------------------------------
def run_walls(ref):
    """World.step with walls (core.py:182-188, 270-307): no in-scope scenario has walls, so a
    simple_tag-like world is assembled from the reference classes with a box of walls (one soft,
    one agent a ghost) and stepped through the reference's own World.step."""
    core = ref.core
    rng = np.random.RandomState(20240605)
    B, A, L = 24, 4, 2
    walls = [dict(orient="H", axis_pos=-1.0, endpoints=(-0.8, 0.8), width=0.1, hard=True),
             dict(orient="H", axis_pos=1.0, endpoints=(-1.0, 1.0), width=0.2, hard=False),
             dict(orient="V", axis_pos=-1.0, endpoints=(-1.0, 0.6), width=0.1, hard=True),
             dict(orient="V", axis_pos=1.0, endpoints=(-0.5, 1.0), width=0.05, hard=True)]
    world = core.World()
    world.dim_c = 2
    world.agents = [core.Agent() for _ in range(A)]
    for i, a in enumerate(world.agents):
        a.collide, a.silent = True, True
        a.size, a.accel, a.max_speed = (0.075, 3.0, 1.0) if i < 3 else (0.05, 4.0, 1.3)
        a.ghost = (i == 1)
        a.action.c = np.zeros(2)
    world.landmarks = [core.Landmark() for _ in range(L)]
    for lm in world.landmarks:
        lm.collide, lm.movable, lm.size = True, False, 0.2
    world.walls = [core.Wall(orient=w["orient"], axis_pos=w["axis_pos"], endpoints=w["endpoints"], width=w["width"], hard=w["hard"])
                   for w in walls]
    pos0 = np.concatenate([rng.uniform(-1, 1, (B, A, 2)), rng.uniform(-0.6, 0.6, (B, L, 2))], axis=1)
    vel0 = rng.uniform(-0.5, 0.5, (B, A, 2))
    # agents pressed against / just past the ends of each wall
    for b in range(8):
        w = walls[b % 4]
        along = [w["endpoints"][0] - 0.05, w["endpoints"][1] + 0.04, 0.5 * sum(w["endpoints"]), w["endpoints"][1] - 0.01][b // 2 % 4]
        off = w["axis_pos"] + (0.06 if b % 2 else -0.06)
        for i in range(A):
            pos0[b, i] = (along + 0.017 * i, off) if w["orient"] == "H" else (off, along + 0.017 * i)
    actions = rng.randint(0, 5, (T_STEPS, B, A))
    rec = {"pos": np.zeros((T_STEPS, B, A, 2)), "vel": np.zeros((T_STEPS, B, A, 2))}
    for b in range(B):
        inject(world, pos0[b], vel0[b])
        for t in range(T_STEPS):
            for i, a in enumerate(world.agents):
                u = np.zeros(2)
                m = int(actions[t, b, i])
                if m == 1: u[0] = -1.0
                if m == 2: u[0] = +1.0
                if m == 3: u[1] = -1.0
                if m == 4: u[1] = +1.0
                a.action.u = u * a.accel
            world.step()
            rec["pos"][t, b] = [a.state.p_pos for a in world.agents]
            rec["vel"][t, b] = [a.state.p_vel for a in world.agents]
    rec.update(pos0=pos0, vel0=vel0, actions=actions.astype(np.int32),
               wall_vertical=np.array([w["orient"] == "V" for w in walls]), wall_axis_pos=np.array([w["axis_pos"] for w in walls]),
               wall_endpoints=np.array([w["endpoints"] for w in walls]), wall_width=np.array([w["width"] for w in walls]),
               wall_hard=np.array([w["hard"] for w in walls]), ghost=np.array([a.ghost for a in world.agents]))
    return rec


def run_new_core(ref):
    """World.step of multiagent/new_core.py (:188-205): action force scaled by mass (* accel) (:214),
    contacts between movable entities split by their mass ratio (:290-294), walls, speed clamp.  A
    predator-prey-like world (heavy adversaries, light good agents, one immovable agent, an obstacle,
    walls) is assembled from the reference classes; it is stepped with and without the distance cache
    (cache_dists), which must not change a bit."""
    nc = ref.new_core
    rng = np.random.RandomState(20240631)
    B, A, L = 24, 7, 3
    walls = [dict(orient="H", axis_pos=-1.2, endpoints=(-1.2, 1.2), width=0.4, hard=True),
             dict(orient="V", axis_pos=-1.2, endpoints=(-1.2, 1.2), width=0.4, hard=True),
             dict(orient="H", axis_pos=1.2, endpoints=(-1.2, 1.2), width=0.4, hard=True),
             dict(orient="V", axis_pos=1.2, endpoints=(-1.2, 1.2), width=0.4, hard=False)]

    def build(cache):
        world = nc.World()
        world.dim_c = 3
        world.cache_dists = cache
        world.agents = [nc.Agent() for _ in range(A)]
        for i, a in enumerate(world.agents):
            a.i = i
            a.collide, a.silent = True, True
            adv = i < 3
            a.size, a.initial_mass, a.max_speed = (0.065, 2.25, 1.0) if adv else (0.045, 1.0, 1.3)
            a.accel = 1.5 if i in (1, 5) else None
            a.ghost = (i == 4)
            a.action.c = np.zeros(3)
        world.agents[6].movable = False
        world.landmarks = [nc.Landmark() for _ in range(L)]
        for l, lm in enumerate(world.landmarks):
            lm.i = A + l
            lm.collide, lm.movable, lm.size = (l == 0), False, (0.2 if l == 0 else 0.03)
        world.walls = [nc.Wall(orient=w["orient"], axis_pos=w["axis_pos"], endpoints=w["endpoints"], width=w["width"], hard=w["hard"])
                       for w in walls]
        return world

    pos0 = np.concatenate([rng.uniform(-1, 1, (B, A, 2)), rng.uniform(-0.9, 0.9, (B, L, 2))], axis=1)
    vel0 = rng.uniform(-0.5, 0.5, (B, A, 2))
    vel0[:, 6] = 0.0
    # heavy-light, heavy-heavy, light-light pairs and an agent-obstacle pair in contact; agents at the walls
    for b in range(12):
        i, j = [(0, 3), (0, 1), (3, 5), (1, 4), (2, 6), (5, 6)][b % 6]
        d = [0.02, 0.06, 0.1, 0.105, 0.115, 0.09][b // 2]
        ang = rng.uniform(0, 2 * np.pi)
        pos0[b, j] = pos0[b, i] + d * np.array([np.cos(ang), np.sin(ang)])
    for b in range(12, 16):
        pos0[b, b % 6] = pos0[b, A] + 0.22 * np.array([np.cos(b), np.sin(b)])
    for b in range(16, 24):
        k = b - 16
        pos0[b, k % 6] = [(-0.99, 0.3), (0.3, -1.01), (0.98, 0.2), (-0.2, 1.0), (1.05, 1.05), (-1.0, -1.0), (0.0, 0.97), (0.99, 0.99)][k]
    actions = rng.randint(0, 5, (T_STEPS, B, A))
    recs = []
    for cache in (False, True):
        world = build(cache)
        rec = {"pos": np.zeros((T_STEPS, B, A, 2)), "vel": np.zeros((T_STEPS, B, A, 2))}
        for b in range(B):
            inject(world, pos0[b], vel0[b])
            if cache:
                world.cached_dist_vect = None
                world.calculate_distances()
            for t in range(T_STEPS):
                for i, a in enumerate(world.agents):
                    a.action.u = _move_u(int(actions[t, b, i]), a.accel)      # what _set_action leaves in action.u
                world.step()
                rec["pos"][t, b] = [a.state.p_pos for a in world.agents]
                rec["vel"][t, b] = [a.state.p_vel for a in world.agents]
        recs.append(rec)
    assert np.array_equal(recs[0]["pos"], recs[1]["pos"]) and np.array_equal(recs[0]["vel"], recs[1]["vel"]), "cache_dists changed the step"
    rec = recs[0]
    world = build(False)
    rec.update(pos0=pos0, vel0=vel0, actions=actions.astype(np.int32),
               size=np.array([e.size for e in world.entities]), mass=np.array([e.mass for e in world.entities]),
               movable=np.array([e.movable for e in world.entities]), collide=np.array([e.collide for e in world.entities]),
               accel=np.array([np.nan if a.accel is None else a.accel for a in world.agents]),
               max_speed=np.array([a.max_speed for a in world.agents]), ghost=np.array([a.ghost for a in world.agents]),
               wall_vertical=np.array([w["orient"] == "V" for w in walls]), wall_axis_pos=np.array([w["axis_pos"] for w in walls]),
               wall_endpoints=np.array([w["endpoints"] for w in walls]), wall_width=np.array([w["width"] for w in walls]),
               wall_hard=np.array([w["hard"] for w in walls]))
    return rec


def run_classic(ref, name, seed, B=32):
    """simple / simple_adversary / simple_push through MultiAgentEnv.step.  The goal landmark is the
    reference's own np.random.choice draw (reset is repeated until it yields the wanted index, so the
    goal-dependent colours are the ones reset_world assigns); positions are then injected."""
    sc = ref.scenarios[name].Scenario()
    world = sc.make_world()
    info_cb = sc.benchmark_data if hasattr(sc, "benchmark_data") else None
    env = ref.environment.MultiAgentEnv(world, lambda w, rng: sc.reset_world(w), sc.reward, sc.observation, info_cb)
    A, L = len(world.agents), len(world.landmarks)
    rng = np.random.RandomState(seed)
    pos0 = rng.uniform(-1, 1, (B, A + L, 2))
    vel0 = np.zeros((B, A, 2))
    vel0[B // 2:] = rng.uniform(-1, 1, (B - B // 2, A, 2))
    goal = rng.randint(0, L, B)
    if name == "simple_push":       # agents start in contact (dist_min = 0.1) in a quarter of the envs
        for k, b in enumerate(range(0, B, 4)):
            d = [0.02, 0.05, 0.09, 0.0999, 0.1001, 0.11, 0.15, 0.08][k % 8]
            ang = rng.uniform(0, 2 * np.pi)
            pos0[b, 1] = pos0[b, 0] + d * np.array([np.cos(ang), np.sin(ang)])
    if name == "simple_adversary":  # a good agent sitting on / next to the goal in a few envs
        for b in range(0, B, 8):
            pos0[b, 1] = pos0[b, A + goal[b]] + rng.uniform(-0.01, 0.01, 2)
    actions = rng.randint(0, 5, (T_STEPS, B, A))
    dims = [sp.shape[0] for sp in env.observation_space]
    rec = {"pos": np.zeros((T_STEPS, B, A, 2)), "vel": np.zeros((T_STEPS, B, A, 2)),
           "obs": np.zeros((T_STEPS, B, sum(dims))), "obs_dims": np.array(dims, np.int64),
           "rew": np.zeros((T_STEPS, B, A)), "done": np.zeros((T_STEPS, B), np.uint8), "goal": goal.astype(np.int32)}
    if info_cb is not None:
        rec["info_sq"] = np.zeros((T_STEPS, B, 1 + (A - 1) * (L + 1)))
    env.seed(0)
    np.random.seed(seed)
    for b in range(B):
        for _ in range(1000):
            env.reset()
            if not hasattr(world.agents[0], "goal_a") or world.agents[0].goal_a is world.landmarks[goal[b]]:
                break
        else:
            raise RuntimeError("goal draw never matched")
        inject(world, pos0[b], vel0[b])
        for t in range(T_STEPS):
            obs_n, rew_n, done, info_n = env.step([int(a) for a in actions[t, b]])
            for i, ag in enumerate(world.agents):
                rec["pos"][t, b, i] = ag.state.p_pos
                rec["vel"][t, b, i] = ag.state.p_vel
            rec["obs"][t, b] = np.concatenate([np.asarray(o, np.float64) for o in obs_n])
            rec["rew"][t, b] = [float(r) for r in rew_n]
            rec["done"][t, b] = bool(done)
            if info_cb is not None:
                rec["info_sq"][t, b] = np.concatenate([np.atleast_1d(np.asarray(x, np.float64)) for x in info_n["n"]])
    return pos0, vel0, actions, rec


def run_comm(ref, name, seed, B=32):
    """simple_speaker_listener / simple_reference / simple_crypto.  The landmark indices reset_world
    draws with np.random.choice (goal_b of the speaker; goal_b of both agents; goal and key) are the
    reference's own draws, repeated until they give the indices the fixture asks for."""
    sc = ref.scenarios[name].Scenario()
    world = sc.make_world()
    info_cb = sc.benchmark_data if name == "simple_crypto" else None      # the other two define none that runs
    env = ref.environment.MultiAgentEnv(world, lambda w, rng: sc.reset_world(w), sc.reward, sc.observation, info_cb)
    assert env.shared_reward == (name != "simple_crypto")
    A, L, dim_c = len(world.agents), len(world.landmarks), world.dim_c
    speakers = [i for i, a in enumerate(world.agents) if not a.silent]
    rng = np.random.RandomState(seed)
    pos0 = rng.uniform(-1, 1, (B, A + L, 2))
    vel0 = np.zeros((B, A, 2))
    vel0[B // 2:] = rng.uniform(-1, 1, (B - B // 2, A, 2))
    for i, a in enumerate(world.agents):
        if not a.movable:
            vel0[:, i] = 0.0
    n_aux = 1 if name == "simple_speaker_listener" else 2
    aux = rng.randint(0, L, (B, n_aux))
    cols = []
    for a in world.agents:
        if a.movable:
            cols.append(rng.randint(0, 5, (T_STEPS, B)))
        if not a.silent:
            cols.append(rng.randint(0, dim_c, (T_STEPS, B)))
    actions = np.stack(cols, axis=2)

    def drawn():
        lm = world.landmarks
        if name == "simple_speaker_listener":
            return [lm.index(world.agents[0].goal_b)]
        if name == "simple_reference":
            return [lm.index(world.agents[0].goal_b), lm.index(world.agents[1].goal_b)]
        return [lm.index(world.agents[0].goal_a), int(np.argmax(world.agents[2].key))]

    dims = [sp.shape[0] for sp in env.observation_space]
    rec = {"pos": np.zeros((T_STEPS, B, A, 2)), "vel": np.zeros((T_STEPS, B, A, 2)),
           "obs": np.zeros((T_STEPS, B, sum(dims))), "obs_dims": np.array(dims, np.int64),
           "rew": np.zeros((T_STEPS, B, A)), "rew_own": np.zeros((T_STEPS, B, A)), "done": np.zeros((T_STEPS, B), np.uint8),
           "comm": np.zeros((T_STEPS, B, len(speakers) * dim_c)), "goal": aux.astype(np.int32) if n_aux > 1 else aux[:, 0].astype(np.int32),
           "obs0": np.zeros((B, sum(dims))), "rew0": np.zeros((B, A))}
    if info_cb is not None:
        rec["info_c_goal"] = np.zeros((T_STEPS, B, A, 2 * dim_c))
    env.seed(0)
    np.random.seed(seed)
    for b in range(B):
        for _ in range(2000):
            env.reset()
            if drawn() == list(aux[b]):
                break
        else:
            raise RuntimeError("landmark draws never matched")
        inject(world, pos0[b], vel0[b])
        rec["obs0"][b] = np.concatenate([np.asarray(sc.observation(a, world), np.float64) for a in world.agents])
        rec["rew0"][b] = [float(sc.reward(a, world)) for a in world.agents]
        for t in range(T_STEPS):
            if name == "simple_reference":
                for i, a in enumerate(world.agents):
                    a.action.u = _move_u(int(actions[t, b, 2 * i]), a.accel)
                    c = np.zeros(dim_c)
                    c[int(actions[t, b, 2 * i + 1])] = 1.0          # environment.py:199-201
                    a.action.c = c
                world.step()
                obs_n = [sc.observation(a, world) for a in world.agents]
                own = [sc.reward(a, world) for a in world.agents]
                rew_n = [np.sum(own)] * A                            # environment.py:103-105
                done, inf_n = False, None
            else:
                obs_n, rew_n, done, info_n = env.step([int(x) for x in actions[t, b]])
                own = [info_n["r%d" % i] for i in range(A)]
                inf_n = info_n["n"]
            for i, ag in enumerate(world.agents):
                rec["pos"][t, b, i] = ag.state.p_pos
                rec["vel"][t, b, i] = ag.state.p_vel
            rec["obs"][t, b] = np.concatenate([np.asarray(o, np.float64) for o in obs_n])
            rec["rew"][t, b] = [float(r) for r in rew_n]
            rec["rew_own"][t, b] = [float(r) for r in own]
            rec["done"][t, b] = bool(done)
            rec["comm"][t, b] = np.concatenate([world.agents[i].state.c for i in speakers])
            if info_cb is not None:
                rec["info_c_goal"][t, b] = [np.concatenate([np.asarray(x[0], np.float64), np.asarray(x[1], np.float64)]) for x in inf_n]
    return pos0, vel0, actions, rec


def run_predator_prey(ref, seed=20240641, B=32):
    sc = ref.scenarios["simple_predator_prey"].Scenario()
    world = sc.make_world()
    env = ref.environment.MultiAgentEnv(world, sc.reset_world, sc.reward, sc.observation, sc.benchmark_data,
                                        post_step_callback=lambda w, rng: sc.post_step(w))
    A, L = len(world.agents), len(world.landmarks)
    assert (A, L) == (7, 6) and not env.shared_reward
    rng = np.random.RandomState(seed)
    pos0 = np.concatenate([rng.uniform(-1, 1, (B, A, 2)), rng.uniform(-0.9, 0.9, (B, L, 2))], axis=1)
    vel0 = np.zeros((B, A, 2))
    alive0 = np.ones((B, 3), bool)
    holding0 = np.zeros((B, 3), bool)
    actions = rng.randint(0, 5, (T_STEPS, B, A))
    food = lambda f: A + 1 + f
    forest = lambda f: A + 4 + f

    def near(center, r):
        ang = rng.uniform(0, 2 * np.pi)
        return center + r * np.array([np.cos(ang), np.sin(ang)])

    # envs 0..7: a member on / next to a food (pick-up distance 0.075); in 6 and 7 two members compete
    for b in range(8):
        d = [0.01, 0.05, 0.0749, 0.0751, 0.03, 0.07, 0.02, 0.04][b]
        pos0[b, 4 + b % 3] = near(pos0[b, food(b % 3)], d)
        actions[:3, b, 4 + b % 3] = 0
        if b >= 6:
            pos0[b, 4 + (b + 1) % 3] = near(pos0[b, food(b % 3)], d + 0.01)
    # envs 8..11: a member holding food next to the good leader (deposit distance 0.09)
    for k, b in enumerate(range(8, 12)):
        holding0[b, k % 3] = True
        pos0[b, 4 + k % 3] = near(pos0[b, 3], [0.02, 0.0899, 0.0901, 0.2][k])
    # envs 12..15: food already eaten (parked at -999 until it respawns)
    for k, b in enumerate(range(12, 16)):
        alive0[b, k % 3] = False
        pos0[b, food(k % 3)] = (-999.0, -999.0)
        holding0[b, (k + 1) % 3] = True
    # envs 16..19: forests kept apart, agents dealt into / out of them
    for b in range(16, 20):
        pos0[b, forest(0)], pos0[b, forest(1)] = (-0.5, -0.4), (0.5, 0.4)
        for i in range(A):
            where = (b + i * (1 + b % 3)) % 4
            if where < 2:
                pos0[b, i] = near(pos0[b, forest(where)], rng.uniform(0, 0.2))
            elif where == 2:
                pos0[b, i] = near(pos0[b, forest(0)], 0.25 + world.agents[i].size + rng.uniform(-0.02, 0.02))
    # envs 20..23: an adversary on a member that holds food (contact distance 0.11), heavy on light
    for k, b in enumerate(range(20, 24)):
        holding0[b, k % 3] = True
        pos0[b, k % 3] = near(pos0[b, 4 + k % 3], [0.03, 0.1099, 0.1101, 0.08][k])
    # envs 24..27: agents at the walls, pushing outwards
    for k, b in enumerate(range(24, 28)):
        pos0[b, k] = [(-0.97, 0.3), (0.3, -0.99), (0.985, 0.2), (-0.2, 0.96)][k]
        actions[:, b, k] = [1, 3, 2, 4][k]
        pos0[b, 6] = [(0.99, 0.99), (-1.02, -0.98), (1.0, -1.0), (-0.96, 1.01)][k]
    # envs 28..31: a moving cluster around the barrier
    for b in range(28, 32):
        for i in range(A):
            pos0[b, i] = near(pos0[b, A], rng.uniform(0.2, 0.35))
        vel0[b] = rng.uniform(-1, 1, (A, 2))
    dims = [sp.shape[0] for sp in env.observation_space]
    rec = {"pos": np.zeros((T_STEPS, B, A, 2)), "vel": np.zeros((T_STEPS, B, A, 2)),
           "obs": np.zeros((T_STEPS, B, sum(dims))), "obs_dims": np.array(dims, np.int64),
           "rew": np.zeros((T_STEPS, B, A)), "done": np.zeros((T_STEPS, B), np.uint8), "collisions": np.zeros((T_STEPS, B, A), np.int64),
           "food_pos": np.zeros((T_STEPS, B, 3, 2)), "alive": np.zeros((T_STEPS, B, 3), bool), "holding": np.zeros((T_STEPS, B, 3), bool),
           "obs0": np.zeros((B, sum(dims)))}
    env.seed(0)
    np.random.seed(seed)
    for b in range(B):
        env.reset()
        inject(world, pos0[b], vel0[b])
        for f, lm in enumerate(world.foods):
            lm.alive = bool(alive0[b, f])
        for k, m in enumerate(world.agents[4:]):
            m.holding = bool(holding0[b, k])
        world.calculate_distances()
        rec["obs0"][b] = np.concatenate([np.asarray(sc.observation(a, world), np.float64) for a in world.agents])
        for t in range(T_STEPS):
            obs_n, rew_n, done, info_n = env.step([int(a) for a in actions[t, b]])
            for i, ag in enumerate(world.agents):
                rec["pos"][t, b, i] = ag.state.p_pos
                rec["vel"][t, b, i] = ag.state.p_vel
            rec["obs"][t, b] = np.concatenate([np.asarray(o, np.float64) for o in obs_n])
            rec["rew"][t, b] = [float(r) for r in rew_n]
            rec["done"][t, b] = bool(done)
            rec["collisions"][t, b] = info_n["n"]
            rec["food_pos"][t, b] = [f.state.p_pos for f in world.foods]
            rec["alive"][t, b] = [f.alive for f in world.foods]
            rec["holding"][t, b] = [m.holding for m in world.agents[4:]]
    rec.update(alive0=alive0, holding0=holding0)
    assert np.array_equal(rec["obs"].astype(np.float32), rec["obs"])        # the scenario casts to float32 (:349)
    rec["obs"], rec["obs0"] = rec["obs"].astype(np.float32), rec["obs0"].astype(np.float32)
    print("predator_prey: pick-ups", int((~rec["alive"]).sum()), "holding steps", int(rec["holding"].sum()),
          "collisions", int(rec["collisions"].sum()))
    return pos0, vel0, actions, rec


def run_consensus(ref, seed=20240651, B=32):
    """consensus through MultiAgentEnv.step (new_core physics, walls, 10-step history observations,
    shared reward over 6 agents)."""
    sc = ref.scenarios["consensus"].Scenario()
    world = sc.make_world()
    env = ref.environment.MultiAgentEnv(world, sc.reset_world, sc.reward, sc.observation, sc.benchmark_data)
    A = len(world.agents)
    assert A == 6 and len(world.landmarks) == 0 and env.shared_reward
    rng = np.random.RandomState(seed)
    pos0 = rng.uniform(-1, 1, (B, A, 2))
    vel0 = np.zeros((B, A, 2))
    vel0[B // 2:] = rng.uniform(-1, 1, (B - B // 2, A, 2))
    # agents at / beyond the arena edge (bound() branches, wall contact) and on the target circle
    for k, b in enumerate(range(8)):
        x = [0.85, 0.8999, 0.95, 0.9999, 1.0, 1.05, 1.3, 0.97][k]
        pos0[b, k % A] = (x if k % 2 == 0 else -x, 0.99 * (-1) ** k)
    pos0[8, 0] = (0.8, 0.0)
    pos0[9, 0] = (0.0, 0.0)
    # histories: a fresh episode (all rows equal the start, consensus.py:56-58) or a recorded past
    his_pos0 = np.repeat(pos0[:, :, None, :], 10, axis=2)
    his_vel0 = np.repeat(vel0[:, :, None, :], 10, axis=2)
    past = slice(B // 4, B)
    his_pos0[past, :, :9] += rng.uniform(-0.3, 0.3, (B - B // 4, A, 9, 2))
    his_vel0[past, :, :9] = rng.uniform(-1, 1, (B - B // 4, A, 9, 2))
    actions = rng.randint(0, 5, (T_STEPS, B, A))
    for k in range(4):
        actions[:, k, k % A] = [2, 1, 2, 1][k]           # keep pushing into the wall
    dims = [sp.shape[0] for sp in env.observation_space]
    rec = {"pos": np.zeros((T_STEPS, B, A, 2)), "vel": np.zeros((T_STEPS, B, A, 2)),
           "obs": np.zeros((T_STEPS, B, sum(dims)), np.float32), "obs_dims": np.array(dims, np.int64),
           "rew": np.zeros((T_STEPS, B, A)), "rew_own": np.zeros((T_STEPS, B, A)), "done": np.zeros((T_STEPS, B), np.uint8),
           "obs0": np.zeros((B, sum(dims)), np.float32), "his_pos0": his_pos0, "his_vel0": his_vel0}
    env.seed(0)
    for b in range(B):
        env.reset()
        inject(world, pos0[b], vel0[b])
        for i, a in enumerate(world.agents):
            a.state.p_his_pos = his_pos0[b, i].copy()
            a.state.p_his_vel = his_vel0[b, i].copy()
        rec["obs0"][b] = np.concatenate([sc.observation(a, world) for a in world.agents])
        for t in range(T_STEPS):
            obs_n, rew_n, done, info_n = env.step([int(a) for a in actions[t, b]])
            for i, ag in enumerate(world.agents):
                rec["pos"][t, b, i] = ag.state.p_pos
                rec["vel"][t, b, i] = ag.state.p_vel
            rec["obs"][t, b] = np.concatenate(obs_n)
            rec["rew"][t, b] = [float(r) for r in rew_n]
            rec["rew_own"][t, b] = [float(info_n["r%d" % i]) for i in range(A)]
            rec["done"][t, b] = bool(done)
            assert all(x == (0, 0, 0, 0) for x in info_n["n"])          # no landmarks, agents do not collide
    return pos0, vel0, actions, rec


def run_collector(ref, seed=20240661, B=32):
    """collector through MultiAgentEnv.step with post_step as post_step_callback (new_core physics with
    mass-ratio contacts, sorted-nearest observations, pick-up / deposit / respawn state machine).  Respawn
    positions and types come from env.np_random; the fixture records where every treasure went."""
    sc = ref.scenarios["collector"].Scenario()
    world = sc.make_world()
    env = ref.environment.MultiAgentEnv(world, sc.reset_world, sc.reward, sc.observation, sc.benchmark_data,
                                        post_step_callback=sc.post_step)
    A, L, NC = len(world.agents), len(world.landmarks), 6
    assert (A, L) == (8, 6) and not env.shared_reward and not world.walls
    rng = np.random.RandomState(seed)
    pos0 = np.concatenate([rng.uniform(-1, 1, (B, A, 2)), rng.uniform(-0.95, 0.95, (B, L, 2))], axis=1)
    vel0 = np.zeros((B, A, 2))
    vel0[B // 2:] = rng.uniform(-0.7, 0.7, (B - B // 2, A, 2))
    holding0 = np.full((B, NC), -1, np.int64)
    ttype0 = rng.randint(0, 2, (B, L))
    alive0 = np.ones((B, L), bool)
    actions = rng.randint(0, 5, (T_STEPS, B, A))

    def near(center, r):
        ang = rng.uniform(0, 2 * np.pi)
        return center + r * np.array([np.cos(ang), np.sin(ang)])

    # envs 0..7: a collector on / next to a treasure (pick-up distance 0.075); in 6 and 7 two collectors compete
    for b in range(8):
        d = [0.01, 0.04, 0.0749, 0.0751, 0.03, 0.06, 0.02, 0.05][b]
        pos0[b, b % NC] = near(pos0[b, A + b % L], d)
        actions[:2, b, b % NC] = 0
        if b >= 6:
            pos0[b, (b + 1) % NC] = pos0[b, b % NC] + 0.11 * np.array([np.cos(b), np.sin(b)])
    # envs 8..11: a collector holding a treasure next to a deposit (hand-over distance 0.125), right or wrong type
    for k, b in enumerate(range(8, 12)):
        holding0[b, k] = k % 2
        pos0[b, k] = near(pos0[b, NC + (k % 2 if k < 3 else 0)], [0.02, 0.1249, 0.1251, 0.05][k])
    # envs 12..15: treasures already taken (parked at the origin until they respawn); two at once tie in distance
    for k, b in enumerate(range(12, 16)):
        for l in ([k], [k, k + 1])[k % 2]:
            alive0[b, l % L] = False
            pos0[b, A + l % L] = (-0.0, -0.0)
        holding0[b, k] = ttype0[b, k % L]
    # envs 16..19: collectors in contact (0.1), a collector pressed on a heavy deposit (0.125)
    for k, b in enumerate(range(16, 20)):
        pos0[b, 1] = near(pos0[b, 0], [0.03, 0.0999, 0.1001, 0.07][k])
        pos0[b, 2] = near(pos0[b, NC + k % 2], [0.05, 0.12, 0.13, 0.1][k])
    # envs 20..23: everybody carries something; in 22 and 23 nobody carries type 1
    for b in range(20, 24):
        holding0[b] = rng.randint(0, 2, NC) if b < 22 else 0
    dims = [sp.shape[0] for sp in env.observation_space]
    rec = {"pos": np.zeros((T_STEPS, B, A, 2)), "vel": np.zeros((T_STEPS, B, A, 2)),
           "obs": np.zeros((T_STEPS, B, sum(dims)), np.float32), "obs_dims": np.array(dims, np.int64),
           "rew": np.zeros((T_STEPS, B, A)), "done": np.zeros((T_STEPS, B), np.uint8), "info": np.zeros((T_STEPS, B, A, 3), np.int64),
           "treasure_pos": np.zeros((T_STEPS, B, L, 2)), "ttype": np.zeros((T_STEPS, B, L), np.int64), "alive": np.zeros((T_STEPS, B, L), bool),
           "holding": np.zeros((T_STEPS, B, NC), np.int64), "obs0": np.zeros((B, sum(dims)), np.float32)}
    env.seed(seed)
    for b in range(B):
        env.reset()
        inject(world, pos0[b], vel0[b])
        for l, lm in enumerate(world.landmarks):
            lm.alive, lm.type = bool(alive0[b, l]), int(ttype0[b, l])
        for k, a in enumerate(world.agents[:NC]):
            a.holding = None if holding0[b, k] < 0 else int(holding0[b, k])
        world.calculate_distances()
        sc.reset_cached_rewards()
        rec["obs0"][b] = np.concatenate([sc.observation(a, world) for a in world.agents])
        for t in range(T_STEPS):
            obs_n, rew_n, done, info_n = env.step([int(a) for a in actions[t, b]])
            for i, ag in enumerate(world.agents):
                rec["pos"][t, b, i] = ag.state.p_pos
                rec["vel"][t, b, i] = ag.state.p_vel
            rec["obs"][t, b] = np.concatenate(obs_n)
            rec["rew"][t, b] = [float(r) for r in rew_n]
            rec["done"][t, b] = bool(done)
            rec["info"][t, b] = info_n["n"]
            rec["treasure_pos"][t, b] = [lm.state.p_pos for lm in world.landmarks]
            rec["ttype"][t, b] = [lm.type for lm in world.landmarks]
            rec["alive"][t, b] = [lm.alive for lm in world.landmarks]
            rec["holding"][t, b] = [-1 if a.holding is None else a.holding for a in world.agents[:NC]]
    rec.update(holding0=holding0, ttype0=ttype0, alive0=alive0)
    print("collector: pick-ups", int((~rec["alive"]).sum()), "holding steps", int((rec["holding"] >= 0).sum()),
          "hand-overs", int(((rec["holding"][:-1] >= 0) & (rec["holding"][1:] < 0)).sum()), "collisions", int(rec["info"][..., 2].sum()))
    return pos0, vel0, actions, rec


def run_noise(ref, seed=20240681, B=8, T=10):
    """simple_world_comm with Agent.u_noise / Agent.c_noise set: the noise terms of World.step (core.py:165,211),
    which no built-in scenario switches on.  The reference draws them with np.random.randn from the global
    generator; the generator is seeded per env, and a twin RandomState with the same seed, asked for the same
    draws in the same order (movable agents with u_noise in agent order, then speakers with c_noise), gives the
    draws the fixture stores next to the trajectory."""
    rng = np.random.RandomState(seed)
    pos0, vel0, actions = init_world_comm(rng)                  # 32 envs; the first B are used
    pos0, vel0, actions = pos0[:B], vel0[:B], actions[:T, :B]
    sc, world, env = make_env(ref, "simple_world_comm")
    A, dim_c = len(world.agents), world.dim_c
    u_noise = np.array([0.3, 0.1, 0.0, 0.2, 0.5, 0.0])
    c_noise = np.array([0.25, 0.0, 0.0, 0.0, 0.0, 0.0])
    for ag, un, cn in zip(world.agents, u_noise, c_noise):
        ag.u_noise = float(un) if un else None
        ag.c_noise = float(cn) if cn else None
    dims = [s.shape[0] for s in env.observation_space]
    rec = {"pos0": pos0, "vel0": vel0, "actions": actions, "u_noise": u_noise, "c_noise": c_noise,
           "noise_u": np.zeros((T, B, A, 2)), "noise_c": np.zeros((T, B, A, dim_c)),
           "pos": np.zeros((T, B, A, 2)), "vel": np.zeros((T, B, A, 2)), "comm": np.zeros((T, B, dim_c)),
           "obs": np.zeros((T, B, sum(dims))), "rew": np.zeros((T, B, A)), "obs_dims": np.array(dims, np.int64)}
    for b in range(B):
        sc.reset_world(world)
        inject(world, pos0[b], vel0[b])
        np.random.seed(seed + 1 + b)
        twin = np.random.RandomState(seed + 1 + b)
        for t in range(T):
            act = actions[t, b]
            leader = world.agents[0]
            leader.action.u = _move_u(int(act[0]), leader.accel)        # environment.py:176-181,192-195
            c = np.zeros(dim_c)
            c[int(act[1])] = 1.0                                        # environment.py:199-201
            leader.action.c = c
            for i in range(1, A):
                env._set_action(int(act[1 + i]), world.agents[i], env.action_space[i])
            world.step()
            for i, ag in enumerate(world.agents):                       # the draws of apply_action_force
                if ag.movable and ag.u_noise:
                    rec["noise_u"][t, b, i] = twin.randn(2)
            for i, ag in enumerate(world.agents):                       # the draws of update_agent_state
                if not ag.silent and ag.c_noise:
                    rec["noise_c"][t, b, i] = twin.randn(dim_c)
            for i, ag in enumerate(world.agents):
                rec["pos"][t, b, i], rec["vel"][t, b, i] = ag.state.p_pos, ag.state.p_vel
            rec["comm"][t, b] = leader.state.c
            rec["obs"][t, b] = np.concatenate([np.asarray(sc.observation(ag, world), np.float64) for ag in world.agents])
            rec["rew"][t, b] = [float(sc.reward(ag, world)) for ag in world.agents]
        # the twin has been asked for exactly the draws the reference took: both generators now agree
        assert np.random.randn() == twin.randn()
    return rec


def run_done(ref, seed=20240602, bound=1.0):
    """simple_tag through MultiAgentEnv.step with a done_callback (environment.py:15-18): the agent is beyond the
    arena bound in either coordinate -- what the comment at environment.py:141-142 has in mind.  step returns
    any(done_n) (:98,110-113).  Same initial states and actions as tag.npz (its envs 9..15 head outwards)."""
    rng = np.random.RandomState(seed)
    pos0, vel0, actions = init_tag(rng)
    sc = ref.scenarios["simple_tag"].Scenario()
    world = sc.make_world()
    done_cb = lambda agent, w: bool(np.any(np.abs(agent.state.p_pos) > bound))
    env = ref.environment.MultiAgentEnv(world, lambda w, rng_: sc.reset_world(w), sc.reward, sc.observation, sc.benchmark_data,
                                        done_callback=done_cb)
    T, B, A = actions.shape
    rec = {"pos0": pos0, "vel0": vel0, "actions": actions.astype(np.int32), "bound": np.float64(bound),
           "pos": np.zeros((T, B, A, 2)), "done": np.zeros((T, B), np.uint8), "done_n": np.zeros((T, B, A), np.uint8)}
    env.seed(0)
    for b in range(B):
        env.reset()
        inject(world, pos0[b], vel0[b])
        for t in range(T):
            _, _, done, _ = env.step([int(a) for a in actions[t, b]])
            assert done is True or done is False            # a single bool (environment.py:110-113)
            rec["done"][t, b] = done
            rec["done_n"][t, b] = [env._get_done(ag) for ag in world.agents]
            rec["pos"][t, b] = [ag.state.p_pos for ag in world.agents]
    print("done: done steps", int(rec["done"].sum()), "of", T * B, "; envs ever done", int(rec["done"].any(axis=0).sum()))
    return rec


def run_gae(ref, seed=20240671):
    """Advantages and returns from the reference's own Runner.run (multiagent/backup/runner_old.py:39-80,
    the nenv > 1 branch): the runner is driven with a scripted env and model that replay fixed rewards,
    values and done flags, so the fixture holds exactly what its scan computes."""
    runner_mod = _load("ref_runner_old", os.path.join(REF, "multiagent", "backup", "runner_old.py"))
    rng = np.random.RandomState(seed)
    T, B, A = 25, 48, 3
    rewards = rng.uniform(-2, 1, (T, B, A)).astype(np.float32)
    values = rng.uniform(-3, 3, (T, B, A)).astype(np.float32)
    last_values = rng.uniform(-3, 3, (B, A)).astype(np.float32)
    dones = rng.uniform(size=(T, B)) < 0.08
    dones[T - 1, ::5] = True
    gamma, lam = 0.99, 0.95

    class Env:
        num_envs = B

        def __init__(self):
            self.t = 0

        def reset(self):
            return np.zeros((B, A, 1), np.float32)

        def step(self, actions):
            t = self.t
            self.t += 1
            return np.zeros((B, A, 1), np.float32), rewards[t], dones[t].copy(), {}

    class Model:
        def __init__(self):
            self.t = 0

        def step(self, obs, nenv):
            t = self.t
            self.t += 1
            return np.zeros((B, A), np.int64), values[t]

        def value(self, obs, nenv):
            return last_values

        def share_actions(self, mb_actions):
            return mb_actions

    world = types.SimpleNamespace(n=A)
    runner = runner_mod.Runner(env=Env(), world=world, model=Model(), nsteps=T, gamma=gamma, lam=lam)
    mb_obs, mb_actions, mb_rewards, mb_values, mb_advs, mb_returns, mb_dones = runner.run()
    unpack = lambda per_agent: np.stack([np.asarray(x).reshape(T, B) for x in per_agent], axis=2)
    assert np.array_equal(unpack(mb_rewards), rewards) and np.array_equal(unpack(mb_values), values)
    return dict(rewards=rewards, values=values, dones=dones, last_values=last_values, last_dones=dones[T - 1].copy(),
                gamma=np.array(gamma), lam=np.array(lam), advs=unpack(mb_advs).astype(np.float32), returns=unpack(mb_returns).astype(np.float32))


def known_answers(ref):
    """Anchors that do not use state injection."""
    out = {}
    # KAT1: simple_spread, env.seed(1); env.reset(); actions a[t][i] = (t+i) % 5
    sc, world, env = make_env(ref, "simple_spread")
    env.seed(1)
    env.reset()
    out["kat1_pos0"] = np.array([e.state.p_pos for e in world.entities])
    ret0 = 0.0
    for t in range(T_STEPS):
        obs_n, rew_n, done, info = env.step([(t + i) % 5 for i in range(3)])
        ret0 += rew_n[0]
    out["kat1_pos"] = np.array([a.state.p_pos for a in world.agents])
    out["kat1_vel"] = np.array([a.state.p_vel for a in world.agents])
    out["kat1_rew"] = np.array([float(r) for r in rew_n])
    out["kat1_return0"] = np.array(ret0)
    out["kat1_info0"] = np.array([float(x) for x in info["n"][0]])
    out["kat1_obs"] = np.concatenate([np.asarray(o, np.float64) for o in obs_n])
    # comm_matrix metadata of simple_spread.make_world (simple_spread.py:15-22)
    out["spread_comm_matrix"] = np.asarray(world.comm_matrix, np.float64)
    # softplus penetration values through get_collision_force
    core = ref.core
    w = core.World()
    a, b = core.Agent(), core.Agent()
    a.size = b.size = 0.15
    gaps = np.array([-0.3, -0.01, -1e-4, 0.0, 1e-4, 0.01, 0.05, 0.7, 0.75, 1.0])
    fx = []
    for g in gaps:
        a.state.p_pos = np.array([0.0, 0.0])
        b.state.p_pos = np.array([0.3 + g, 0.0])
        fa, fb = w.get_collision_force(a, b)
        fx.append(fa[0])
    out["pen_gaps"] = gaps
    out["pen_force_x"] = np.array(fx)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--only", default=None, help="write this fixture alone (supported: done)")
    args = ap.parse_args()
    ref = load_reference()
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "done.npz"), **run_done(ref))
    print("wrote done.npz")
    if args.only == "done":
        return
    cases = {}

    rng = np.random.RandomState(20240601)
    pos0, vel0, actions = init_spread3(rng)
    sc, world, env = make_env(ref, "simple_spread")
    cases["spread3"] = ("simple_spread", dict(A=3, L=3), pos0, vel0, actions,
                        run_env_api(sc, world, env, "simple_spread", pos0, vel0, actions))

    rng = np.random.RandomState(20240602)
    pos0, vel0, actions = init_tag(rng)
    sc, world, env = make_env(ref, "simple_tag")
    cases["tag"] = ("simple_tag", {}, pos0, vel0, actions,
                    run_env_api(sc, world, env, "simple_tag", pos0, vel0, actions))

    rng = np.random.RandomState(20240603)
    pos0, vel0, actions = init_world_comm(rng)
    sc, world, env = make_env(ref, "simple_world_comm")
    cases["world_comm"] = ("simple_world_comm", {}, pos0, vel0, actions,
                           run_world_comm(sc, world, env, pos0, vel0, actions))

    rng = np.random.RandomState(20240604)
    pos0, vel0, actions = init_spread64(rng)
    sc, world, env = make_spread_world(ref, 64, 64)
    obs_steps = [0, T_STEPS - 1]
    rec = run_env_api(sc, world, env, "simple_spread", pos0, vel0, actions, obs_steps=obs_steps)
    rec["obs_steps"] = np.array(obs_steps)
    cases["spread64"] = ("simple_spread", dict(A=64, L=64), pos0, vel0, actions, rec)

    for k, (name, seed) in enumerate((("simple", 20240611), ("simple_adversary", 20240612), ("simple_push", 20240613))):
        pos0, vel0, actions, rec = run_classic(ref, name, seed)
        cases[name] = (name, {}, pos0, vel0, actions, rec)

    for name, seed in (("simple_speaker_listener", 20240621), ("simple_reference", 20240622), ("simple_crypto", 20240623)):
        pos0, vel0, actions, rec = run_comm(ref, name, seed)
        cases[name] = (name, {}, pos0, vel0, actions, rec)

    pos0, vel0, actions, rec = run_predator_prey(ref)
    np.savez_compressed(os.path.join(OUT, "simple_predator_prey.npz"), scenario=np.array("simple_predator_prey"), pos0=pos0, vel0=vel0,
                        actions=actions.astype(np.int32), **rec)
    print("wrote simple_predator_prey.npz")

    pos0, vel0, actions, rec = run_collector(ref)
    np.savez_compressed(os.path.join(OUT, "collector.npz"), scenario=np.array("collector"), pos0=pos0, vel0=vel0,
                        actions=actions.astype(np.int32), **rec)
    print("wrote collector.npz")

    pos0, vel0, actions, rec = run_consensus(ref)
    np.savez_compressed(os.path.join(OUT, "consensus.npz"), scenario=np.array("consensus"), pos0=pos0, vel0=vel0,
                        actions=actions.astype(np.int32), **rec)
    print("wrote consensus.npz")

    for key, (scn, kw, pos0, vel0, actions, rec) in cases.items():
        path = os.path.join(OUT, key + ".npz")
        np.savez_compressed(path, scenario=np.array(scn), pos0=pos0, vel0=vel0,
                            actions=actions.astype(np.int32), **rec)
        print("wrote", path, os.path.getsize(path), "bytes")
    np.savez_compressed(os.path.join(OUT, "gae.npz"), **run_gae(ref))
    np.savez_compressed(os.path.join(OUT, "noise.npz"), **run_noise(ref))
    print("wrote gae.npz")
    walls = run_walls(ref)
    np.savez_compressed(os.path.join(OUT, "walls.npz"), **walls)
    print("wrote walls.npz")
    np.savez_compressed(os.path.join(OUT, "new_core.npz"), **run_new_core(ref))
    print("wrote new_core.npz")
    kat = known_answers(ref)
    np.savez_compressed(os.path.join(OUT, "kat.npz"), **kat)
    print("wrote kat.npz")

    if args.check:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import mpe_oracle as orc
        for key, (scn, kw, pos0, vel0, actions, rec) in cases.items():
            spec = orc.make_spec(scn, **kw)
            pos, vel = pos0.copy(), vel0.copy()
            worst = {}
            for t in range(actions.shape[0]):
                pos, vel, comm, obs, rew, done = orc.env_step(spec, pos, vel, actions[t], rec.get("goal"))
                dev = {
                    "pos": np.abs(pos[:, :spec.A] - rec["pos"][t]).max(),
                    "vel": np.abs(vel - rec["vel"][t]).max(),
                    "rew": np.abs(rew - rec["rew"][t]).max(),
                }
                if "obs_steps" not in rec:
                    dev["obs"] = np.abs(np.concatenate(obs, axis=1) - rec["obs"][t]).max()
                if "info_sq" in rec:
                    dev["info_sq"] = np.abs(orc.info(spec, pos, rec["goal"])["sq"] - rec["info_sq"][t]).max()
                if "rew_own" in rec:
                    dev["rew_own"] = np.abs(orc.reward(spec, pos, rec["goal"], comm) - rec["rew_own"][t]).max()
                    dev["comm"] = np.abs(comm[:, ~spec.silent].reshape(comm.shape[0], -1) - rec["comm"][t]).max()
                if "info_c_goal" in rec:
                    dev["info_c_goal"] = np.abs(orc.info(spec, pos, rec["goal"], comm)["c_goal"] - rec["info_c_goal"][t]).max()
                for k, v in dev.items():
                    worst[k] = max(worst.get(k, 0.0), float(v))
            print(key, "oracle-vs-reference max |delta|:", worst)


if __name__ == "__main__":
    main()
