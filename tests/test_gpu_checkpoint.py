"""Checkpoint / resume of a batch: BatchedEnv.state_dict() taken in mid-episode and loaded into a fresh env (through
torch.save / torch.load) continues bit for bit, across time-limit resets on the device, for every kernel family."""
import io

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

CASES = [("simple_spread", {}), ("simple_world_comm", {}), ("simple_reference", {}), ("simple_predator_prey", {}), ("collector", {}),
         ("consensus", {}), ("simple_spread", dict(n_agents=9, n_landmarks=4))]


@pytest.mark.parametrize("scenario,kw", CASES)
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_resume_from_a_checkpoint_is_bit_exact(scenario, kw, dtype):
    import mpe_hip
    B, H, before, after = 150, 6, 8, 11            # the checkpoint falls inside the second episode, the rest crosses two limits
    make = lambda: mpe_hip.BatchedEnv(scenario, B, dtype=getattr(torch, dtype), device="cuda:0", max_episode_steps=H, seed=3, **kw)
    env = make()
    gen = torch.Generator("cuda:0").manual_seed(17)
    acts = torch.randint(0, 5, (before + after, B, env.n_act), dtype=torch.int32, device="cuda:0", generator=gen)
    for col, n in enumerate(env.action_dims):
        if n != 5:
            acts[:, :, col] = torch.randint(0, n, (before + after, B), dtype=torch.int32, device="cuda:0", generator=gen)
    env.reset(seed=3)
    for t in range(before):
        env.step(acts[t])
    buf = io.BytesIO()
    torch.save(env.state_dict(), buf)
    want = []
    for t in range(before, before + after):
        obs, rew, done, _ = env.step(acts[t])
        want.append((obs.clone(), rew.clone(), done.clone()))
    final = env.state_dict()

    buf.seek(0)
    other = make()
    other.load_state_dict(torch.load(buf, weights_only=True))
    for k, t in enumerate(range(before, before + after)):
        obs, rew, done, _ = other.step(acts[t])
        assert torch.equal(obs, want[k][0]) and torch.equal(rew, want[k][1]) and torch.equal(done, want[k][2]), (scenario, dtype, t)
    got = other.state_dict()
    for name, x in final["tensors"].items():
        assert torch.equal(x, got["tensors"][name]), (scenario, dtype, name)
    # a manual reset after the resume draws the same worlds in both
    a, b = env.reset(seed=9).clone(), other.reset(seed=9)
    assert torch.equal(a, b)


def test_a_checkpoint_of_another_env_is_refused():
    import mpe_hip
    sd = mpe_hip.BatchedEnv("simple_spread", 64, device="cuda:0").state_dict()
    with pytest.raises(ValueError):
        mpe_hip.BatchedEnv("simple_spread", 128, device="cuda:0").load_state_dict(sd)
    with pytest.raises(ValueError):
        mpe_hip.BatchedEnv("simple_tag", 64, device="cuda:0").load_state_dict(sd)
    with pytest.raises(ValueError):      # bookkeeping arrays missing from the checkpoint
        mpe_hip.BatchedEnv("simple_spread", 64, device="cuda:0", max_episode_steps=5).load_state_dict(sd)


def test_a_checkpoint_of_another_world_description_is_refused():
    """Same scenario, size and dtype, but another time limit, another place in the job's env index or other noise
    levels: the tensors would load, the continuation would not be the one that was saved."""
    import dataclasses

    import numpy as np

    import mpe_hip
    sd = mpe_hip.BatchedEnv("simple_tag", 64, device="cuda:0", max_episode_steps=5).state_dict()
    mpe_hip.BatchedEnv("simple_tag", 64, device="cuda:0", max_episode_steps=5).load_state_dict(sd)
    with pytest.raises(ValueError, match="max_episode_steps"):
        mpe_hip.BatchedEnv("simple_tag", 64, device="cuda:0", max_episode_steps=7).load_state_dict(sd)
    with pytest.raises(ValueError, match="env_base"):
        mpe_hip.BatchedEnv("simple_tag", 64, device="cuda:0", max_episode_steps=5, env_base=64).load_state_dict(sd)
    noisy = dataclasses.replace(mpe_hip.make_spec("simple_tag"), u_noise=np.array([0.1, 0.0, 0.0, 0.2]))
    with pytest.raises(ValueError, match="fingerprint"):
        mpe_hip.BatchedEnv(noisy, 64, device="cuda:0", max_episode_steps=5).load_state_dict(sd)


def test_resume_continues_the_noise_stream():
    """A world with Agent.u_noise: the checkpoint carries the generator's seed and step number."""
    import dataclasses

    import numpy as np

    import mpe_hip
    B = 100
    spec = dataclasses.replace(mpe_hip.make_spec("simple_tag"), u_noise=np.array([0.3, 0.0, 0.1, 0.2]))
    acts = torch.randint(0, 5, (9, B, 4), dtype=torch.int32, device="cuda:0", generator=torch.Generator("cuda:0").manual_seed(2))
    env = mpe_hip.BatchedEnv(spec, B, dtype=torch.float64, device="cuda:0")
    env.reset(seed=4)
    env.set_noise(seed=11)
    for t in range(4):
        env.step(acts[t])
    sd = env.state_dict()
    assert sd["noise"] == [11, 4]
    want = [env.step(acts[t])[0].clone() for t in range(4, 9)]
    other = mpe_hip.BatchedEnv(spec, B, dtype=torch.float64, device="cuda:0")
    other.load_state_dict(sd)
    for k, t in enumerate(range(4, 9)):
        assert torch.equal(other.step(acts[t])[0], want[k]), t


def test_a_masked_reset_keeps_the_seed_of_the_automatic_resets():
    import mpe_hip
    env = mpe_hip.BatchedEnv("simple_spread", 64, device="cuda:0", max_episode_steps=5, seed=3)
    env.reset(seed=3)
    mask = torch.zeros(64, dtype=torch.uint8)
    mask[:10] = 1
    env.reset(seed=3, mask=mask)
    with pytest.raises(ValueError, match="seed"):
        env.reset(seed=4, mask=mask)
    env.reset(seed=4)                       # every env: the batch moves to the new seed
    assert env.seed == 4
