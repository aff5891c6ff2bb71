"""The drop-in single-world API (MultiAgentEnv / make_env / Scenario plugins) on the GPU path,
against the reference's own outputs (golden fixtures and known answers)."""
import numpy as np
import pytest

import mpe_oracle as orc
from conftest import load_golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _close(a, b, rtol=1e-9, atol=1e-12):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= atol + rtol * np.abs(b))


def test_make_env_simple_spread_known_answer():
    """env.seed(1); env.reset(); 25 steps of actions (t+i) % 5 -- the same calls made on the
    reference when tests/golden/kat.npz was captured (numpy RandomState(1) reset draws)."""
    from make_env import make_env
    k = load_golden("kat")
    env = make_env("simple_spread", benchmark=True)
    assert env.n == 3 and [s.n for s in env.action_space] == [5, 5, 5]
    assert [s.shape for s in env.observation_space] == [(14,)] * 3
    assert np.array_equal(env.world.comm_matrix, k["spread_comm_matrix"])
    env.seed(1)
    obs_n = env.reset()
    assert np.array_equal(np.array([e.state.p_pos for e in env.world.entities]), k["kat1_pos0"])
    assert obs_n[0].dtype == np.float32 and obs_n[0].shape == (14,)
    ret0 = 0.0
    for t in range(25):
        obs_n, rew_n, done, info = env.step([(t + i) % 5 for i in range(3)])
        ret0 += rew_n[0]
        assert done is False and set(info) == {"n", "r0", "r1", "r2"}
    assert _close([a.state.p_pos for a in env.world.agents], k["kat1_pos"])
    assert _close([a.state.p_vel for a in env.world.agents], k["kat1_vel"])
    assert _close(rew_n, k["kat1_rew"])
    assert _close(ret0, k["kat1_return0"])
    assert _close(np.concatenate(obs_n), k["kat1_obs"], 1e-6, 1e-7)
    r, coll, md, occ = info["n"][0]
    assert _close([r, md], k["kat1_info0"][[0, 2]]) and [coll, occ] == list(k["kat1_info0"][[1, 3]])


@pytest.mark.parametrize("name,case", [("simple_tag", "tag"), ("simple_world_comm", "world_comm")])
def test_builtin_scenarios_replay_reference_episode(name, case):
    from make_env import make_env
    g = load_golden(case)
    env = make_env(name, benchmark=True)
    env.seed(0)
    env.reset()
    A = env.n
    dims = list(g["obs_dims"])
    assert [s.shape[0] for s in env.observation_space] == dims
    for b in (0, 9, 17, 30):
        for i, e in enumerate(env.world.entities):
            e.state.p_pos = g["pos0"][b, i].copy()
        for i, a in enumerate(env.world.agents):
            a.state.p_vel = g["vel0"][b, i].copy()
            a.state.c = np.zeros(env.world.dim_c)
        for t in range(25):
            act = g["actions"][t, b]
            action_n = [(int(act[0]), int(act[1]))] + [int(x) for x in act[2:]] if name == "simple_world_comm" else [int(x) for x in act]
            obs_n, rew_n, done, info = env.step(action_n)
            assert _close(np.concatenate(obs_n), g["obs"][t, b]), (b, t)
            assert _close(rew_n, g["rew"][t, b]), (b, t)
            assert list(info["n"]) == list(g["collisions"][t, b]) and done is False
        assert _close([a.state.p_pos for a in env.world.agents], g["pos"][24, b])
        assert _close([a.state.p_vel for a in env.world.agents], g["vel"][24, b])
    if name == "simple_world_comm":
        assert type(env.action_space[0]).__name__ == "Tuple" and env.action_space[1].n == 5
        assert np.array_equal(env.world.agents[0].state.c, g["comm"][24, 30])


def test_agent_noise_levels_use_numpys_global_generator_like_the_reference():
    """Agent.u_noise / c_noise on a built-in scenario (core.py:165,211): the step leaves the fused launch and
    World.step draws from np.random in the reference's order (every movable agent's force noise, then the
    speakers' comm noise).  tests/golden/noise.npz was captured from the reference with the global generator
    seeded 20240681 + 1 + b for env b, so seeding it alike must replay that trajectory."""
    from make_env import make_env
    g = load_golden("noise")
    T, B, _ = g["actions"].shape
    env = make_env("simple_world_comm")
    assert env._scenario is not None                      # without noise levels this env steps fused
    env.seed(0)
    env.reset()
    for a, un, cn in zip(env.world.agents, g["u_noise"], g["c_noise"]):
        a.u_noise, a.c_noise = (float(un) if un else None), (float(cn) if cn else None)
    for b in (0, 5):
        for i, e in enumerate(env.world.entities):
            e.state.p_pos = g["pos0"][b, i].copy()
        for i, a in enumerate(env.world.agents):
            a.state.p_vel = g["vel0"][b, i].copy()
            a.state.c = np.zeros(env.world.dim_c)
        np.random.seed(20240681 + 1 + b)
        for t in range(T):
            act = g["actions"][t, b]
            obs_n, rew_n, done, _ = env.step([(int(act[0]), int(act[1]))] + [int(x) for x in act[2:]])
            assert _close([a.state.p_pos for a in env.world.agents], g["pos"][t, b]), (b, t)
            assert _close([a.state.p_vel for a in env.world.agents], g["vel"][t, b]), (b, t)
            assert _close(env.world.agents[0].state.c, g["comm"][t, b]), (b, t)
            assert _close(np.concatenate(obs_n), g["obs"][t, b]) and _close(rew_n, g["rew"][t, b]), (b, t)
    # and the noise does change the step: the same start without it ends elsewhere
    for a in env.world.agents:
        a.u_noise = a.c_noise = None
    for i, e in enumerate(env.world.entities):
        e.state.p_pos = g["pos0"][0, i].copy()
    for i, a in enumerate(env.world.agents):
        a.state.p_vel = g["vel0"][0, i].copy()
    act = g["actions"][0, 0]
    env.step([(int(act[0]), int(act[1]))] + [int(x) for x in act[2:]])
    assert not _close([a.state.p_pos for a in env.world.agents], g["pos"][0, 0], 1e-6, 1e-6)


class _UserTagScenario:
    """A scenario written against the plugin contract only (Python callbacks reading
    agent.state / agent attributes), as a user of the reference would write it: simple_tag's
    world with a custom reward and observation."""

    def make_world(self):
        from multiagent.core import World, Agent, Landmark
        w = World()
        w.dim_c = 2
        w.agents = [Agent() for _ in range(4)]
        for i, a in enumerate(w.agents):
            a.name, a.collide, a.silent, a.adversary = "agent %d" % i, True, True, i < 3
            a.size, a.accel, a.max_speed = (0.075, 3.0, 1.0) if a.adversary else (0.05, 4.0, 1.3)
        w.landmarks = [Landmark() for _ in range(2)]
        for lm in w.landmarks:
            lm.collide, lm.movable, lm.size = True, False, 0.2
        self.reset_world(w)
        return w

    def reset_world(self, world):
        rng = np.random.RandomState(5)
        for a in world.agents:
            a.state.p_pos, a.state.p_vel, a.state.c = rng.uniform(-1, 1, 2), np.zeros(2), np.zeros(2)
        for lm in world.landmarks:
            lm.state.p_pos, lm.state.p_vel = rng.uniform(-0.9, 0.9, 2), np.zeros(2)

    def reward(self, agent, world):
        return -float(np.sum(np.square(agent.state.p_pos)))

    def observation(self, agent, world):
        return np.concatenate([agent.state.p_vel, agent.state.p_pos] + [lm.state.p_pos - agent.state.p_pos for lm in world.landmarks])


def test_user_scenario_plugin_runs_physics_on_gpu():
    from multiagent.environment import MultiAgentEnv
    sc = _UserTagScenario()
    world = sc.make_world()
    env = MultiAgentEnv(world, sc.reset_world, sc.reward, sc.observation)
    assert env._scenario is None                          # generic path: kernels for physics, Python callbacks
    env.seed(0)
    obs_n = env.reset()
    assert [o.shape for o in obs_n] == [(8,)] * 4
    spec = orc.spec_tag()
    pos = np.array([e.state.p_pos for e in world.entities])[None]
    vel = np.zeros((1, 4, 2))
    rng = np.random.RandomState(3)
    for t in range(25):
        act = rng.randint(0, 5, 4)
        obs_n, rew_n, done, info = env.step(list(act))
        u, c = orc.decode_actions(spec, act[None])
        pos, vel, _ = orc.world_step(spec, pos, vel, u, c)
        assert _close([a.state.p_pos for a in world.agents], pos[0, :4]), t
        assert _close([a.state.p_vel for a in world.agents], vel[0]), t
        assert _close(rew_n[2], -np.sum(pos[0, 2] ** 2))
        assert info == {"n": [{}] * 4, "r0": rew_n[0], "r1": rew_n[1], "r2": rew_n[2], "r3": rew_n[3]}
    assert world.t == 25


def test_env_error_behaviour():
    from make_env import make_env
    env = make_env("simple_spread")
    with pytest.raises(AttributeError):
        env.reset()                                        # the reference also fails before seed()
    env.seed(3)
    env.reset()
    with pytest.raises(ValueError):
        env.step([0, 1])
    wc = make_env("simple_world_comm")
    wc.seed(0)
    wc.reset()
    with pytest.raises(ValueError):
        wc.step([1, 1, 1, 1, 1, 1])                        # the leader needs (move, comm)
    with pytest.raises(IndexError):
        wc.step([(1, 7), 1, 1, 1, 1, 1])
    from mpe_hip import scenarios
    with pytest.raises(FileNotFoundError):
        scenarios.load("no_such_scenario.py")


def test_world_with_walls_steps_on_gpu():
    """A user world with walls: World.step() applies the wall contact of core.py:270-307."""
    from multiagent.core import World, Agent, Wall
    from test_oracle_golden import walls_spec
    g = load_golden("walls")
    spec = walls_spec(g)
    w = World()
    w.dim_c = 2
    for i in range(4):
        a = Agent()
        a.collide, a.silent, a.ghost = True, True, bool(spec.ghost[i])
        a.size, a.accel, a.max_speed = (0.075, 3.0, 1.0) if i < 3 else (0.05, 4.0, 1.3)
        w.agents.append(a)
    from multiagent.core import Landmark
    for _ in range(2):
        lm = Landmark()
        lm.collide, lm.movable, lm.size = True, False, 0.2
        w.landmarks.append(lm)
    w.walls = [Wall(orient=d["orient"], axis_pos=d["axis_pos"], endpoints=d["endpoints"], width=d["width"], hard=d["hard"])
               for d in spec.walls]
    b = 3
    for i, e in enumerate(w.entities):
        e.state.p_pos = g["pos0"][b, i].copy()
        e.state.p_vel = g["vel0"][b, i].copy() if i < 4 else np.zeros(2)
    for a in w.agents:
        a.state.c = np.zeros(2)
        a.action.c = np.zeros(2)
    for t in range(25):
        u, _ = orc.decode_actions(spec, g["actions"][t, b][None])
        for i, a in enumerate(w.agents):
            a.action.u = u[0, i].copy()
        w.step()
        assert _close([a.state.p_pos for a in w.agents], g["pos"][t, b]), t
        assert _close([a.state.p_vel for a in w.agents], g["vel"][t, b]), t


@pytest.mark.parametrize("scenario", ["simple_tag", "simple_world_comm", "simple_predator_prey", "consensus"])
def test_batch_slice_as_world_objects(scenario):
    """BatchedEnv.to_world(b): one env of a batch as the reference's object graph.  The scenario's own callbacks on
    that world give the batch's observation row, and stepping the copy is stepping that env."""
    import mpe_hip
    from mpe_hip.scenarios import load
    B, b = 37, 23
    env = mpe_hip.BatchedEnv(scenario, B, dtype=torch.float64)
    obs = env.reset(seed=4).clone()
    gen = torch.Generator(device="cuda").manual_seed(2)
    for _ in range(12):          # move away from the reset state (consensus: fills its history)
        a = torch.stack([torch.randint(0, n, (B,), dtype=torch.int32, device="cuda", generator=gen) for n in env.action_dims], dim=1).contiguous()
        obs = env.step(a)[0].clone()
    world = env.to_world(b)
    sc = load(scenario + ".py").Scenario()
    mine = sc.make_world()
    assert [e.size for e in world.entities] == [e.size for e in mine.entities]
    assert [(a.max_speed, a.accel, a.silent, a.collide) for a in world.agents] == [(a.max_speed, a.accel, a.silent, a.collide) for a in mine.agents]
    assert len(world.walls) == len(mine.walls)
    pos, vel, _ = env.get_state()
    assert np.array_equal(np.array([e.state.p_pos for e in world.entities]), pos[b])
    assert np.array_equal(np.array([a.state.p_vel for a in world.agents]), vel[b])
    if scenario == "simple_predator_prey":       # the scenario's helper lists and int state live beside the entities
        world.barrier, world.foods, world.forests = world.landmarks[:1], world.landmarks[1:4], world.landmarks[4:]
        flags = int(env.aux[0, b])
        for k, f in enumerate(world.foods):
            f.alive = bool(flags >> k & 1)
        for k, m in enumerate(world.agents[4:]):
            m.holding = bool(flags >> (3 + k) & 1)
    if scenario == "consensus":
        world.comm_matrix = mine.comm_matrix
    got = np.concatenate([sc.observation(a, world) for a in world.agents])
    now = env.observe()         # the callbacks on the current state (a post_step may have changed it since the step's own)
    assert np.array_equal(got, now[b].cpu().numpy().astype(got.dtype))
    if scenario != "simple_predator_prey":
        assert torch.equal(now, obs)
    # refreshing an existing world in place
    a = torch.zeros(B, env.n_act, dtype=torch.int32, device="cuda")
    env.step(a)
    again = env.to_world(b, world)
    assert again is world and np.array_equal(np.array([e.state.p_pos for e in world.entities]), env.get_state()[0][b])
    with pytest.raises(IndexError):
        env.to_world(B)


@pytest.mark.parametrize("name", ["collector", "simple_predator_prey"])
def test_make_env_without_post_step_is_one_fused_launch_and_leaves_post_step_out(name):
    """make_env passes no post_step_callback (make_env.py:36-43), so the reference's step never calls post_step
    (environment.py:107-108): nothing is picked up, handed over or respawned.  That env steps as one fused launch whose
    kernel leaves post_step out; it must equal the unfused path (World.step kernel + callbacks) that a do-nothing
    post_step_callback forces, step for step."""
    from make_env import make_env
    from multiagent.environment import MultiAgentEnv
    from mpe_hip import scenarios
    fused = make_env(name)
    assert fused._scenario is not None and not fused._fused_post_step
    sc = scenarios.load(name + ".py").Scenario()
    world = sc.make_world()
    plain = MultiAgentEnv(world, sc.reset_world, sc.reward, sc.observation, post_step_callback=lambda w, rng: None)
    assert plain._scenario is None
    for env in (fused, plain):
        env.seed(11)
    o1, o2 = fused.reset(), plain.reset()
    assert all(np.array_equal(a, b) for a, b in zip(o1, o2))
    landmarks0 = np.array([lm.state.p_pos for lm in fused.world.landmarks])
    rng = np.random.RandomState(5)
    for t in range(25):
        act = [int(rng.randint(0, 5)) for _ in range(fused.n)]
        o1, r1, d1, _ = fused.step(act)
        o2, r2, d2, _ = plain.step(act)
        assert all(_close(a, b) for a, b in zip(o1, o2)), t
        assert _close(r1, r2) and d1 is False and d2 is False, t
    assert np.array_equal(np.array([lm.state.p_pos for lm in fused.world.landmarks]), landmarks0)      # nothing was taken or respawned
