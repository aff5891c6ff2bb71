"""Agent.u_noise / Agent.c_noise in the batched kernels (multiagent/core.py:165,211; no built-in scenario sets
them).  The reference's own np.random.randn draws, stored in the fixture, are fed to the kernels; the device's
counter-based generator is checked against its numpy restatement."""
import dataclasses

import numpy as np
import pytest

import mpe_oracle as orc
from conftest import load_golden, within_fp32_bar
from test_oracle_golden import noise_spec

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= atol + rtol * np.abs(b))


def _ok(dtype, got, ref, label, f32_factor=1.0, f64=(1e-9, 1e-12)):
    """float64: within f64 = (rtol, atol) of the reference; float32: the stated bar |got - ref| <= f32_factor * 1e-4 *
    max(1, |ref|), its observed margin recorded under `label`."""
    if dtype == "float64":
        return _close(got, ref, *f64)
    return within_fp32_bar(got, ref, label, f32_factor)


def _noisy(name, u_noise, c_noise=None, **kw):
    import mpe_hip
    spec = mpe_hip.make_spec(name, **kw)
    return dataclasses.replace(spec, u_noise=np.asarray(u_noise, float), c_noise=None if c_noise is None else np.asarray(c_noise, float))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_reference_draws_give_the_reference_trajectory(dtype):
    import mpe_hip
    g = load_golden("noise")
    T, B, _ = g["actions"].shape
    env = mpe_hip.BatchedEnv(_noisy("simple_world_comm", g["u_noise"], g["c_noise"]), B, dtype=getattr(torch, dtype))
    env.set_state(g["pos0"], g["vel0"])
    for t in range(T):
        env.set_noise(noise_u=g["noise_u"][t], noise_c=g["noise_c"][t])
        obs, rew, _, _ = env.step(torch.as_tensor(g["actions"][t], dtype=torch.int32))
        pos, vel, comm = env.get_state()
        assert _ok(dtype, pos[:, :6], g["pos"][t], "noise:pos") and _ok(dtype, vel, g["vel"][t], "noise:vel"), t
        assert _ok(dtype, comm, g["comm"][t], "noise:comm") and _ok(dtype, rew.cpu(), g["rew"][t], "noise:rew", 1.0, (1e-8, 1e-11)), t
        assert _ok(dtype, obs.cpu(), g["obs"][t], "noise:obs"), t
    # the same draws for all steps at once through a rollout launch
    env.set_state(g["pos0"], g["vel0"])
    env.set_noise(noise_u=g["noise_u"], noise_c=g["noise_c"])
    out = env.rollout(torch.as_tensor(g["actions"], dtype=torch.int32))
    assert _ok(dtype, out["obs"].cpu(), g["obs"], "noise:obs") and _ok(dtype, env.get_state()[0][:, :6], g["pos"][-1], "noise:pos")


def test_zero_noise_levels_change_nothing():
    import mpe_hip
    B = 200
    a = torch.randint(0, 5, (6, B, 4), dtype=torch.int32, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    plain = mpe_hip.BatchedEnv("simple_tag", B, dtype=torch.float64)
    zero = mpe_hip.BatchedEnv(_noisy("simple_tag", np.zeros(4), np.zeros(4)), B, dtype=torch.float64)
    o1, o2 = plain.rollout(a, reset_seed=3), zero.rollout(a, reset_seed=3)
    assert torch.equal(o1["obs"], o2["obs"]) and torch.equal(plain.pos, zero.pos)
    with pytest.raises(mpe_hip.MpeError):
        zero.set_noise(seed=1)                      # nothing to draw for


@pytest.mark.parametrize("name,kw,levels", [("simple_tag", {}, [0.4, 0.0, 0.2, 0.7]),
                                             ("simple_spread", dict(n_agents=5, n_landmarks=4), [0.3, 0.0, 0.6, 0.1, 0.2]),
                                             ("consensus", {}, [0.5] * 6)])
def test_device_generator_matches_its_numpy_restatement(name, kw, levels):
    """Steps, then a rollout continuing the step numbering, against the oracle fed with device_noise()."""
    import mpe_hip
    B, seed = 300, 77
    env = mpe_hip.BatchedEnv(_noisy(name, levels, **kw), B, dtype=torch.float64)
    env.reset(seed=2, observe=False)
    env.set_noise(seed=seed)
    ospec = orc.make_spec(name, **({"A": 5, "L": 4} if kw else {}))
    ospec.u_noise = np.asarray(levels, float)
    pos, vel, _ = env.get_state()
    if name == "consensus":
        his_pos, his_vel = env.get_history()
    gen = torch.Generator("cuda").manual_seed(5)
    actions = torch.randint(0, 5, (7, B, env.n_act), dtype=torch.int32, device="cuda", generator=gen)

    def oracle_step(t):
        nonlocal pos, vel, his_pos, his_vel
        nu, nc = orc.device_noise(ospec, seed, t, B)
        a = actions[t].cpu().numpy()
        if name == "consensus":
            u, c = orc.decode_actions(ospec, a)
            pos, vel, _ = orc.world_step(ospec, pos, vel, u, c, nu, nc)
        else:
            pos, vel, *_ = orc.env_step(ospec, pos, vel, a, noise_u=nu, noise_c=nc)

    his_pos = his_vel = None
    for t in range(3):
        env.step(actions[t])
        oracle_step(t)
        gp, gv, _ = env.get_state()
        assert _close(gp, pos, 1e-9, 1e-12) and _close(gv, vel, 1e-9, 1e-12), t
    env.rollout(actions[3:])
    for t in range(3, 7):
        oracle_step(t)
    gp, gv, _ = env.get_state()
    assert _close(gp, pos, 1e-9, 1e-12) and _close(gv, vel, 1e-9, 1e-12)
    # another seed is another trajectory; the same seed replays it
    e2 = mpe_hip.BatchedEnv(_noisy(name, levels, **kw), B, dtype=torch.float64)
    e2.reset(seed=2, observe=False)
    e2.set_noise(seed=seed)
    e2.rollout(actions)
    assert torch.equal(e2.pos, env.pos)
    e2.reset(seed=2, observe=False)
    e2._episode = 0
    e2.set_noise(seed=seed + 1)
    e2.rollout(actions)
    assert not torch.equal(e2.pos, env.pos)
