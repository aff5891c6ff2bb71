"""The numpy oracle against the fixtures captured from the reference.

The fixtures hold what the reference's own MultiAgentEnv.step / World.step and
scenario callbacks produced (tools/gen_golden.py); the oracle must reproduce
float64 state, observations and rewards bit for bit and every integer field
exactly.
"""
import numpy as np
import pytest

import mpe_oracle as orc
from conftest import CLASSIC_CASES, COMM_CASES, GOLDEN_CASES, load_golden


@pytest.mark.parametrize("case", list(GOLDEN_CASES))
def test_oracle_reproduces_reference_trajectory(case):
    scn, kw = GOLDEN_CASES[case]
    g = load_golden(case)
    spec = orc.make_spec(scn, **kw)
    assert spec.obs_dims == list(g["obs_dims"])
    pos, vel = g["pos0"].copy(), g["vel0"].copy()
    obs_steps = list(g["obs_steps"]) if "obs_steps" in g else None
    for t in range(g["actions"].shape[0]):
        pos, vel, comm, obs, rew, done = orc.env_step(spec, pos, vel, g["actions"][t])
        assert np.array_equal(pos[:, :spec.A], g["pos"][t]), (case, t)
        assert np.array_equal(vel, g["vel"][t]), (case, t)
        assert np.array_equal(rew, g["rew"][t]), (case, t)
        assert not done.any() and not g["done"][t].any()
        if obs_steps is None:
            assert np.array_equal(np.concatenate(obs, axis=1), g["obs"][t]), (case, t)
        elif t in obs_steps:
            assert np.array_equal(np.concatenate(obs, axis=1), g["obs"][obs_steps.index(t)])
        inf = orc.info(spec, pos)
        assert np.array_equal(inf["collisions"], g["collisions"][t])
        if scn == "simple_spread":
            assert np.array_equal(inf["rew"], g["info_rew"][t])
            assert np.array_equal(inf["min_dists"], g["min_dists"][t])
            assert np.array_equal(inf["occupied"], g["occupied"][t])
        if scn == "simple_world_comm":
            assert np.array_equal(comm[:, 0, :], g["comm"][t])
            inforest, vis = orc.visibility_masks(spec, pos)
            assert np.array_equal(inforest, g["in_forest"][t].astype(bool))
            assert np.array_equal(vis, g["visible"][t].astype(bool))


def test_known_answers():
    k = load_golden("kat")
    spec = orc.spec_spread()
    # KAT1: seeded reset of the reference + actions (t+i) % 5
    pos = k["kat1_pos0"][None].copy()
    vel = np.zeros((1, 3, 2))
    ret0 = 0.0
    for t in range(25):
        act = np.array([[(t + i) % 5 for i in range(3)]])
        pos, vel, comm, obs, rew, done = orc.env_step(spec, pos, vel, act)
        ret0 += rew[0, 0]
    assert np.array_equal(pos[0, :3], k["kat1_pos"])
    assert np.array_equal(vel[0], k["kat1_vel"])
    assert np.array_equal(rew[0], k["kat1_rew"])
    assert ret0 == float(k["kat1_return0"])
    assert np.array_equal(np.concatenate(obs, axis=1)[0], k["kat1_obs"])
    inf = orc.info(spec, pos)
    assert [inf["rew"][0, 0], inf["collisions"][0, 0], inf["min_dists"][0], inf["occupied"][0]] == list(k["kat1_info0"])
    # softplus penetration table through the contact force
    for gap, fx in zip(k["pen_gaps"], k["pen_force_x"]):
        f = orc.collision_force(spec, np.array([[0.0, 0.0]]), np.array([[0.3 + gap, 0.0]]), 0.15, 0.15)
        assert np.array_equal(f[0, 0], fx, equal_nan=True)
    # contact anchor: agents at (0,0),(0.2,0),(5,5) at rest, one no-op step
    pos = np.zeros((1, 6, 2)); pos[0, 1] = (0.2, 0); pos[0, 2] = (5, 5); pos[0, 3:] = 9.0
    p, v, *_ = orc.env_step(spec, pos, np.zeros((1, 3, 2)), np.zeros((1, 3), int))
    assert np.allclose(p[0, :3], [[-0.1, 0], [0.3, 0], [5, 5]], atol=1e-15)
    assert np.allclose(v[0], [[-1, 0], [1, 0], [0, 0]], atol=1e-15)


def test_tag_speed_clamp_anchor():
    g = load_golden("tag")
    spec = orc.spec_tag()
    speeds = np.hypot(g["vel"][9, 8, :, 0], g["vel"][9, 8, :, 1])
    assert np.allclose(speeds, spec.max_speed, rtol=1e-15)


def test_reset_distribution_shapes():
    rng = np.random.RandomState(0)
    for scn in ("simple_spread", "simple_tag", "simple_world_comm"):
        spec = orc.make_spec(scn)
        pos, vel, comm = orc.reset(spec, rng, B=3)
        assert pos.shape == (3, spec.E, 2) and vel.shape == (3, spec.A, 2)
        lim = 1.0 if scn == "simple_spread" else 0.9
        assert np.abs(pos[:, spec.A:]).max() <= lim and np.abs(pos).max() <= 1.0


def walls_spec(g):
    """Oracle spec of the walled world of tests/golden/walls.npz (simple_tag-like entities)."""
    spec = orc.spec_tag()
    spec.walls = [dict(orient="V" if v else "H", axis_pos=float(a), endpoints=(float(e[0]), float(e[1])), width=float(w), hard=bool(h))
                  for v, a, e, w, h in zip(g["wall_vertical"], g["wall_axis_pos"], g["wall_endpoints"], g["wall_width"], g["wall_hard"])]
    spec.ghost = g["ghost"].astype(bool)
    return spec


def test_oracle_reproduces_reference_wall_contact():
    """World.step with walls (core.py:182-188, 270-307), captured from the reference's own World.step."""
    g = load_golden("walls")
    spec = walls_spec(g)
    pos, vel = g["pos0"].copy(), g["vel0"].copy()
    for t in range(g["actions"].shape[0]):
        u, c = orc.decode_actions(spec, g["actions"][t])
        pos, vel, _ = orc.world_step(spec, pos, vel, u, c)
        assert np.array_equal(pos[:, :4], g["pos"][t]) and np.array_equal(vel, g["vel"][t]), t


@pytest.mark.parametrize("case", CLASSIC_CASES)
def test_oracle_reproduces_classic_scenarios(case):
    """simple / simple_adversary / simple_push through the reference's MultiAgentEnv.step, goal
    landmark from the fixture: state, observations, rewards and benchmark_data bit for bit."""
    g = load_golden(case)
    spec = orc.make_spec(case)
    assert spec.obs_dims == list(g["obs_dims"])
    goal = g["goal"] if case != "simple" else None
    pos, vel = g["pos0"].copy(), g["vel0"].copy()
    for t in range(g["actions"].shape[0]):
        pos, vel, comm, obs, rew, done = orc.env_step(spec, pos, vel, g["actions"][t], goal)
        assert np.array_equal(pos[:, :spec.A], g["pos"][t]), (case, t)
        assert np.array_equal(vel, g["vel"][t]), (case, t)
        assert np.array_equal(rew, g["rew"][t]), (case, t)
        assert np.array_equal(np.concatenate(obs, axis=1), g["obs"][t]), (case, t)
        assert not done.any() and not g["done"][t].any()
        if "info_sq" in g:
            assert np.array_equal(orc.info(spec, pos, goal)["sq"], g["info_sq"][t])


def test_reset_with_goal_distribution():
    spec = orc.make_spec("simple_push")
    pos, vel, comm, goal = orc.reset_with_goal(spec, np.random.RandomState(3), 400)
    assert set(goal) == {0, 1} and 0.35 < goal.mean() < 0.65
    assert np.abs(pos).max() <= 1.0 and not vel.any()


@pytest.mark.parametrize("case", COMM_CASES)
def test_oracle_reproduces_comm_scenarios(case):
    """simple_speaker_listener / simple_crypto through the reference's MultiAgentEnv.step and
    simple_reference through World.step + callbacks + np.sum sharing: state, comm state, observations,
    shared and own rewards, benchmark_data and the callbacks on the reset state, bit for bit."""
    g = load_golden(case)
    spec = orc.make_spec(case)
    assert spec.obs_dims == list(g["obs_dims"]) and spec.n_act == g["actions"].shape[2]
    goal = g["goal"]
    pos, vel = g["pos0"].copy(), g["vel0"].copy()
    B = pos.shape[0]
    comm0 = np.zeros((B, spec.A, spec.dim_c))
    assert np.array_equal(np.concatenate(orc.observe(spec, pos, vel, comm0, goal), axis=1), g["obs0"])
    assert np.array_equal(orc.reward(spec, pos, goal, comm0), g["rew0"])
    for t in range(g["actions"].shape[0]):
        pos, vel, comm, obs, rew, done = orc.env_step(spec, pos, vel, g["actions"][t], goal)
        assert np.array_equal(pos[:, :spec.A], g["pos"][t]), (case, t)
        assert np.array_equal(vel, g["vel"][t]), (case, t)
        assert np.array_equal(comm[:, ~spec.silent].reshape(B, -1), g["comm"][t]), (case, t)
        assert np.array_equal(rew, g["rew"][t]), (case, t)
        assert np.array_equal(orc.reward(spec, pos, goal, comm), g["rew_own"][t]), (case, t)
        assert np.array_equal(np.concatenate(obs, axis=1), g["obs"][t]), (case, t)
        assert not done.any() and not g["done"][t].any()
        if "info_c_goal" in g:
            assert np.array_equal(orc.info(spec, pos, goal, comm)["c_goal"], g["info_c_goal"][t])
    if spec.shared:
        assert np.array_equal(g["rew"][..., 0], g["rew_own"].sum(axis=-1)) and np.array_equal(g["rew"][..., 0], g["rew"][..., 1])
    # immovable agents never move
    for i in range(spec.A):
        if not spec.movable[i]:
            assert np.array_equal(g["pos"][-1][:, i], g["pos0"][:, i])


def test_reset_with_two_landmark_draws():
    spec = orc.make_spec("simple_crypto")
    pos, vel, comm, goal = orc.reset_with_goal(spec, np.random.RandomState(4), 300)
    assert goal.shape == (300, 2) and set(goal.ravel()) == {0, 1}


def new_core_spec(g):
    """Oracle spec of the world of tests/golden/new_core.npz (new_core.py physics, walls, one immovable agent)."""
    A = g["accel"].shape[0]
    L = g["size"].shape[0] - A
    spec = orc.Spec("physics", A, L, 3, g["size"], g["movable"].astype(bool), g["collide"].astype(bool), g["mass"],
                    sens=np.where(np.isnan(g["accel"]), 5.0, g["accel"]), max_speed=g["max_speed"], silent=np.ones(A, bool),
                    adversary=np.zeros(A, bool), leader=np.zeros(A, bool), new_core=True, accel=g["accel"])
    spec.walls = [dict(orient="V" if v else "H", axis_pos=float(a), endpoints=(float(e[0]), float(e[1])), width=float(w), hard=bool(h))
                  for v, a, e, w, h in zip(g["wall_vertical"], g["wall_axis_pos"], g["wall_endpoints"], g["wall_width"], g["wall_hard"])]
    spec.ghost = g["ghost"].astype(bool)
    return spec


def test_oracle_reproduces_reference_new_core_step():
    """World.step of multiagent/new_core.py (:188-205), captured from the reference's own World.step."""
    g = load_golden("new_core")
    spec = new_core_spec(g)
    A = spec.A
    pos, vel = g["pos0"].copy(), g["vel0"].copy()
    for t in range(g["actions"].shape[0]):
        u, c = orc.decode_actions(spec, np.where(spec.movable[:A], g["actions"][t], 0)[:, spec.movable[:A]])
        pos, vel, _ = orc.world_step(spec, pos, vel, u, c)
        assert np.array_equal(pos[:, :A], g["pos"][t]) and np.array_equal(vel, g["vel"][t]), t
    assert np.array_equal(g["pos"][-1][:, 6], g["pos0"][:, 6])      # the immovable agent stayed put


def test_oracle_reproduces_predator_prey():
    """simple_predator_prey through the reference's MultiAgentEnv.step (new_core physics, walls,
    post_step pick-up / deposit / respawn), free-running for 25 steps with the respawn positions the
    reference drew fed back in: state, observations, rewards, collisions and flags bit for bit."""
    g = load_golden("simple_predator_prey")
    spec = orc.make_spec("simple_predator_prey")
    A = spec.A
    assert spec.obs_dims == list(g["obs_dims"])
    pos, vel = g["pos0"].copy(), g["vel0"].copy()
    alive, holding = g["alive0"].copy(), g["holding0"].copy()
    assert np.array_equal(np.concatenate(orc.pp_observe(spec, pos, vel, holding), axis=1), g["obs0"])
    for t in range(g["actions"].shape[0]):
        pos, vel, obs, rew, alive, holding = orc.pp_env_step(spec, pos, vel, g["actions"][t], alive, holding, g["food_pos"][t])
        assert np.array_equal(pos[:, :A], g["pos"][t]), t
        assert np.array_equal(vel, g["vel"][t]), t
        assert np.array_equal(np.concatenate(obs, axis=1), g["obs"][t]), t
        assert np.array_equal(rew, g["rew"][t]), t
        assert np.array_equal(alive, g["alive"][t]) and np.array_equal(holding, g["holding"][t]), t
        assert np.array_equal(pos[:, A + 1:A + 4], g["food_pos"][t]), t
    # the fixture exercises the state machine
    assert (~g["alive"]).sum() > 5 and g["holding"].sum() > 50 and g["collisions"].sum() > 10
    # re-evaluate the collision counts on the recorded states
    pos = g["pos0"].copy()
    pos[:, :A] = g["pos"][-1]
    assert np.array_equal(orc.pp_info(spec, pos)["collisions"], g["collisions"][-1])


def test_philox_known_answers():
    """Random123's published vectors for philox4x32-10 (counter words, key words -> output)."""
    full = np.uint64(0xFFFFFFFFFFFFFFFF)
    assert [hex(x) for x in orc.philox4x32(0, 0, 0)] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(x) for x in orc.philox4x32(full, full, full)] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    pi = orc.philox4x32(np.uint64(0x299F31D0A4093822), np.uint64(0x85A308D3243F6A88), np.uint64(0x0370734413198A2E))
    assert [hex(x) for x in pi] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_oracle_reproduces_consensus():
    """consensus through the reference's MultiAgentEnv.step: state, history observations, shared
    (np.sum over six agents) and own rewards bit for bit."""
    g = load_golden("consensus")
    spec = orc.make_spec("consensus")
    assert spec.obs_dims == list(g["obs_dims"]) and spec.n_act == 6
    pos, vel, hp, hv = g["pos0"].copy(), g["vel0"].copy(), g["his_pos0"].copy(), g["his_vel0"].copy()
    assert np.array_equal(np.concatenate(orc.consensus_observe(spec, hp, hv), axis=1), g["obs0"])
    for t in range(g["actions"].shape[0]):
        pos, vel, hp, hv, obs, rew, own = orc.consensus_env_step(spec, pos, vel, hp, hv, g["actions"][t])
        assert np.array_equal(pos, g["pos"][t]) and np.array_equal(vel, g["vel"][t]), t
        assert np.array_equal(np.concatenate(obs, axis=1), g["obs"][t]), t
        assert np.array_equal(own, g["rew_own"][t]) and np.array_equal(rew, g["rew"][t]), t


def test_oracle_reproduces_collector():
    """collector through the reference's MultiAgentEnv.step (post_step as post_step_callback), free-running
    for 25 steps with the respawn positions and types the reference drew fed back in: state, sorted-nearest
    observations, the reward (all of it goes to collector 0), benchmark_data and the pick-up / hand-over
    state bit for bit."""
    g = load_golden("collector")
    spec = orc.make_spec("collector")
    A = spec.A
    assert spec.obs_dims == list(g["obs_dims"])
    pos, vel = g["pos0"].copy(), g["vel0"].copy()
    holding, ttype, alive = g["holding0"].copy(), g["ttype0"].copy(), g["alive0"].copy()
    B = pos.shape[0]
    obs0 = np.stack([np.concatenate(orc.collector_observe1(spec, pos[b], vel[b], holding[b], ttype[b])) for b in range(B)])
    assert np.array_equal(obs0, g["obs0"])
    for t in range(g["actions"].shape[0]):
        pos, vel, obs, rew, info, holding, ttype, alive = orc.collector_env_step(spec, pos, vel, g["actions"][t], holding, ttype, alive,
                                                                                 g["treasure_pos"][t], g["ttype"][t])
        assert np.array_equal(pos[:, :A], g["pos"][t]) and np.array_equal(vel, g["vel"][t]), t
        assert np.array_equal(obs, g["obs"][t]), t
        assert np.array_equal(rew, g["rew"][t]), (t, np.abs(rew - g["rew"][t]).max())
        assert np.array_equal(info, g["info"][t]), t
        assert np.array_equal(holding, g["holding"][t]) and np.array_equal(ttype, g["ttype"][t]) and np.array_equal(alive, g["alive"][t]), t
        assert np.array_equal(pos[:, A:], g["treasure_pos"][t]), t
    assert (~g["alive"]).sum() > 5 and (g["holding"] >= 0).sum() > 100 and g["info"][..., 2].sum() > 10


def test_oracle_reproduces_reference_gae():
    """Advantages / returns of the reference's own Runner.run (runner_old.py:58-80), bit for bit."""
    g = load_golden("gae")
    advs, rets = orc.gae_reference(g["rewards"], g["values"], g["dones"], g["last_values"], g["last_dones"], float(g["gamma"]), float(g["lam"]))
    assert advs.dtype == np.float32 and np.array_equal(advs, g["advs"]) and np.array_equal(rets, g["returns"])
    # the per-agent scan differs from it (the runner chains its accumulator through the agents)
    a2, _ = orc.gae_standard(g["rewards"], g["values"], g["dones"], g["last_values"], float(g["gamma"]), float(g["lam"]))
    assert np.abs(a2 - advs).max() > 0.1
    # and reduces to the textbook recursion for one agent without episode ends
    r, v = g["rewards"][:, :, :1], g["values"][:, :, :1]
    a1, _ = orc.gae_standard(r, v, np.zeros_like(g["dones"]), g["last_values"][:, :1], 0.9, 0.8)
    nxt = np.concatenate([v[1:], g["last_values"][None, :, :1]], axis=0).astype(np.float64)
    delta = r + 0.9 * nxt - v
    want = np.zeros_like(delta)
    acc = 0.0
    for t in reversed(range(r.shape[0])):
        acc = delta[t] + 0.9 * 0.8 * acc
        want[t] = acc
    assert np.allclose(a1, want, rtol=1e-5, atol=1e-5)


def noise_spec(g):
    spec = orc.make_spec("simple_world_comm")
    spec.u_noise, spec.c_noise = g["u_noise"].copy(), g["c_noise"].copy()
    return spec


def test_oracle_reproduces_reference_action_and_comm_noise():
    """Agent.u_noise / c_noise (core.py:165,211): the reference's World.step with its own np.random.randn draws,
    which the fixture stores; the oracle takes the draws as arguments and must land on the same bits."""
    g = load_golden("noise")
    spec = noise_spec(g)
    pos, vel = g["pos0"].copy(), g["vel0"].copy()
    assert np.abs(g["noise_u"][:, :, [2, 5]]).max() == 0 and np.abs(g["noise_u"][:, :, 0]).min() > 0     # only where u_noise is set
    for t in range(g["actions"].shape[0]):
        pos, vel, comm, obs, rew, _ = orc.env_step(spec, pos, vel, g["actions"][t], noise_u=g["noise_u"][t], noise_c=g["noise_c"][t])
        assert np.array_equal(pos[:, :spec.A], g["pos"][t]) and np.array_equal(vel, g["vel"][t]), t
        assert np.array_equal(comm[:, 0, :], g["comm"][t]), t
        assert np.array_equal(np.concatenate(obs, axis=1), g["obs"][t]) and np.array_equal(rew, g["rew"][t]), t
    # the noise matters: without it the trajectory is another one
    p2, v2, *_ = orc.env_step(orc.make_spec("simple_world_comm"), g["pos0"], g["vel0"], g["actions"][0])
    assert np.abs(p2[:, :spec.A] - g["pos"][0]).max() > 1e-3


def test_device_noise_stream_is_standard_normal():
    spec = noise_spec(load_golden("noise"))
    nu, nc = orc.device_noise(spec, seed=5, step=3, B=20000)
    for z in (nu.reshape(-1), nc[:, 0].reshape(-1)):
        assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02 and abs((z ** 3).mean()) < 0.05
    assert not nc[:, 1:].any()                                  # silent agents draw nothing
    nu2, _ = orc.device_noise(spec, seed=5, step=4, B=8)
    assert not np.array_equal(nu[:8], nu2)


@pytest.mark.parametrize("case", ["tag", "world_comm"])
def test_decided_with_margin_marks_envs_whose_integers_survive_perturbation(case):
    """The filter the float32 GPU tests use to demand bit-exact integers: in an env it passes, moving every
    entity by less than a quarter of the margin in any direction leaves the oracle's collision counts and
    forest / visibility masks as they are; the fixtures also hold envs it must reject (seeded onto thresholds)."""
    from conftest import decided_with_margin
    scn, kw = GOLDEN_CASES[case]
    g = load_golden(case)
    spec = orc.make_spec(scn, **kw)
    A, margin = spec.A, 4e-4
    rng = np.random.RandomState(5)
    pos = g["pos0"].copy()
    rejected = 0
    for t in range(0, g["actions"].shape[0], 4):
        pos[:, :A] = g["pos"][t]
        sure = decided_with_margin(spec.size, pos, margin)
        rejected += int((~sure).sum())
        want = orc.info(spec, pos)["collisions"]
        for _ in range(8):
            moved = pos + rng.uniform(-margin / 4, margin / 4, pos.shape) / np.sqrt(2.0)
            assert np.array_equal(orc.info(spec, moved)["collisions"][sure], want[sure]), (case, t)
            if case == "world_comm":
                for a, b in zip(orc.visibility_masks(spec, moved), orc.visibility_masks(spec, pos)):
                    assert np.array_equal(a[sure], b[sure]), (case, t)
    assert rejected > 0


def test_done_callback_restatement_matches_the_reference():
    """done.npz: the reference's MultiAgentEnv.step on simple_tag with done_callback = "the agent is beyond the bound";
    the oracle's done_outside on the oracle's own free-running trajectory gives the same flags at every step."""
    g = load_golden("done")
    spec = orc.make_spec("simple_tag")
    pos, vel = g["pos0"].copy(), g["vel0"].copy()
    bound = float(g["bound"])
    for t in range(g["actions"].shape[0]):
        pos, vel, *_ = orc.env_step(spec, pos, vel, g["actions"][t])
        assert np.array_equal(pos[:, :spec.A], g["pos"][t])
        assert np.array_equal(orc.done_outside(spec, pos, bound), g["done"][t].astype(bool)), t
        assert np.array_equal(np.any(np.abs(pos[:, :spec.A]) > bound, axis=2), g["done_n"][t].astype(bool)), t
    assert g["done"].any() and not g["done"].all()
