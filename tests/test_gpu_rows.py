"""mpe_rollout_rows / mpe_episode_rows: one packed output row per env-step (observation, rewards, done) against the
three-array launches, bit for bit -- full tiles, ragged tails, batches that are no multiple of 4, both float types,
with and without a done bound."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

PACKABLE = ["simple_spread", "simple_tag", "simple", "simple_adversary", "simple_push", "simple_speaker_listener", "simple_reference",
            "simple_crypto"]


def _env(scenario, B, dtype, **kw):
    import mpe_hip
    return mpe_hip.BatchedEnv(scenario, B, dtype=dtype, device="cuda:0", **kw)


def _actions(env, T, seed):
    gen = torch.Generator("cuda:0").manual_seed(seed)
    a = torch.randint(0, 5, (T, env.n_envs, env.n_act), dtype=torch.int32, device="cuda:0", generator=gen)
    for col, n in enumerate(env.action_dims):
        if n != 5:
            a[:, :, col] = torch.randint(0, n, (T, env.n_envs), dtype=torch.int32, device="cuda:0", generator=gen)
    return a


@pytest.mark.parametrize("scenario", PACKABLE)
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("B", [4096, 1001])
def test_packed_rows_equal_the_three_arrays(scenario, dtype, B):
    dt = getattr(torch, dtype)
    T = 25
    e1, e2 = _env(scenario, B, dt), _env(scenario, B, dt)
    lo = e1.layout
    assert lo.row_width == lo.obs_total + e1.n_agents + 1
    a = _actions(e1, T, 3)
    want = e1.rollout(a, reset_seed=7, obs0=True)
    got = e2.rollout(a, reset_seed=7, obs0=True, packed=True)
    assert tuple(got["rows"].shape) == (T, B, lo.row_width)
    assert got["obs"].data_ptr() == got["rows"].data_ptr()            # views, nothing copied
    assert torch.equal(got["obs0"], want["obs0"])
    assert torch.equal(got["obs"], want["obs"]) and torch.equal(got["rew"], want["rew"])
    assert torch.equal(got["done"], want["done"].to(dt))
    for k in ("pos", "vel", "comm", "aux"):
        if getattr(e1, k) is not None:
            assert torch.equal(getattr(e1, k), getattr(e2, k)), k
    # continuing from the state (no reset), a launch that is not a whole episode
    w2, g2 = e1.rollout(a[:7]), e2.rollout(a[:7], packed=True)
    assert torch.equal(g2["obs"], w2["obs"]) and torch.equal(g2["rew"], w2["rew"]) and torch.equal(e1.pos, e2.pos)


@pytest.mark.parametrize("B", [64, 130, 37, 3])
def test_ragged_batches(B):
    e1, e2 = _env("simple_tag", B, torch.float32), _env("simple_tag", B, torch.float32)
    a = _actions(e1, 12, B)
    want, got = e1.rollout(a, reset_seed=1, obs0=True), e2.rollout(a, reset_seed=1, obs0=True, packed=True)
    for k in ("obs0", "obs", "rew"):
        assert torch.equal(got[k], want[k]), k
    assert not got["done"].any()


def test_done_bound_lands_in_the_row():
    e1, e2 = (_env("simple_tag", 2048, torch.float32, done_bound=0.9) for _ in range(2))
    a = _actions(e1, 25, 5)
    want, got = e1.rollout(a, reset_seed=2), e2.rollout(a, reset_seed=2, packed=True)
    assert want["done"].any() and not want["done"].all()
    assert torch.equal(got["done"], want["done"].float()) and torch.equal(got["obs"], want["obs"])


def test_refusals():
    import mpe_hip
    for scenario in ("simple_world_comm", "simple_predator_prey", "collector", "consensus"):
        env = _env(scenario, 64, torch.float32)
        assert env.layout.row_width == 0
        with pytest.raises(ValueError):
            env.rollout(_actions(env, 3, 1), packed=True)
    env = _env("simple_spread", 64, torch.float32, n_agents=9, n_landmarks=4)        # a wave-per-world handle
    assert env.layout.row_width == 0
    with pytest.raises(ValueError):
        _env("simple_tag", 64, torch.float32, info=True).rollout(_actions(env, 3, 1)[:, :, :4].contiguous(), packed=True)
    # straight through the C ABI: an unsupported scenario is MPE_ERR_UNSUPPORTED, a misaligned array MPE_ERR_INVALID
    import ctypes as C
    from mpe_hip import _lib
    lib = _lib.load()
    wc = _env("simple_world_comm", 64, torch.float32)
    buf = torch.zeros(1 << 16, dtype=torch.float32, device="cuda:0")
    acts = _actions(wc, 1, 1)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.mpe_rollout_rows(wc._h, p(acts), p(wc.pos), p(wc.vel), p(wc.comm), None, p(buf), 1, None)
    assert rc == -2 and b"packed rows" in lib.mpe_last_error(wc._h)
    tag = _env("simple_tag", 64, torch.float32)
    acts = _actions(tag, 1, 1)
    rc = lib.mpe_rollout_rows(tag._h, p(acts), p(tag.pos), p(tag.vel), None, None, C.c_void_p(buf.data_ptr() + 4), 1, None)
    assert rc == -1 and b"aligned" in lib.mpe_last_error(tag._h)
