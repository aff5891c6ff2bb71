"""Launches of several steps of float32 simple_spread worlds larger than 3/3 run on a kernel of their own (one wave
steps the world, the other streams the observations as 16-byte vectors); a launch of one step runs on the
workgroup-per-world kernel that both waves share.  The two must agree bit for bit: same operations in the same order
(simple_spread.py:58-130, core.py:150-204)."""
import dataclasses

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

# A * (A + L + 1) even: the world's observation block is a whole number of 16-byte vectors (the streaming kernel);
# the odd cases stay on the shared kernel and are here as the control
SIZES = [(64, 64), (10, 7), (4, 9), (33, 32), (2, 1), (9, 5)]


def _env(A, L, B, **kw):
    import mpe_hip
    spec = kw.pop("spec", None)
    return mpe_hip.BatchedEnv(spec if spec is not None else "simple_spread", B, dtype=torch.float32, device="cuda:0",
                              **({} if spec is not None else dict(n_agents=A, n_landmarks=L)), **kw)


def _actions(T, B, A, seed):
    gen = torch.Generator("cuda:0").manual_seed(seed)
    return torch.randint(0, 5, (T, B, A), dtype=torch.int32, device="cuda:0", generator=gen)


def _same_state(e1, e2):
    assert torch.equal(e1.pos, e2.pos) and torch.equal(e1.vel, e2.vel)


@pytest.mark.parametrize("A,L", SIZES)
@pytest.mark.parametrize("T", [2, 25, 71])
def test_rollout_equals_repeated_step(A, L, T):
    """71 steps: beyond two refills of the 32-step action buffer."""
    B = 37 if A < 64 else 5
    stepper, roller = _env(A, L, B, info=True), _env(A, L, B, info=True)
    for e in (stepper, roller):
        e.reset(seed=11, observe=False)
    a = _actions(T, B, A, 1)
    out = roller.rollout(a)
    for t in range(T):
        obs, rew, done, (ii, ff) = stepper.step(a[t])
        assert torch.equal(obs, out["obs"][t]), (t, (obs != out["obs"][t]).nonzero()[:4])
        assert torch.equal(rew, out["rew"][t]) and torch.equal(done, out["done"][t]), t
        assert torch.equal(ii, out["info_i"][t]) and torch.equal(ff, out["info_f"][t]), t
    _same_state(stepper, roller)


@pytest.mark.parametrize("A,L", SIZES)
def test_episode_launch_equals_reset_then_steps(A, L):
    B, T = 130, 9          # 130 worlds: the done flags' 128-byte blocks, one whole and one ragged
    stepper, epi = _env(A, L, B), _env(A, L, B)
    a = _actions(T, B, A, 2)
    out = epi.rollout(a, reset_seed=5, obs0=True)
    first = stepper.reset(seed=5)
    assert torch.equal(first, out["obs0"][0])
    for t in range(T):
        obs, rew, done, _ = stepper.step(a[t])
        assert torch.equal(obs, out["obs"][t]) and torch.equal(rew, out["rew"][t]) and torch.equal(done, out["done"][t]), t
    _same_state(stepper, epi)
    assert not bool(out["done"].any())


@pytest.mark.parametrize("A,L", [(64, 64), (10, 7), (9, 5)])
def test_launch_of_several_episodes_equals_single_episode_launches(A, L):
    B, K, P = 33, 3, 7
    many, single = _env(A, L, B), _env(A, L, B)
    a = _actions(K * P, B, A, 3)
    out = many.rollout(a, reset_seed=21, episodes=K, obs0=True)
    for k in range(K):
        o = single.rollout(a[k * P:(k + 1) * P].contiguous(), reset_seed=21, obs0=True)
        assert torch.equal(o["obs0"][0], out["obs0"][k]), k
        for key in ("obs", "rew", "done"):
            assert torch.equal(o[key], out[key][k * P:(k + 1) * P]), (k, key)
    _same_state(many, single)
    # the landmarks of the last episode are the ones left in the state
    assert not torch.equal(out["obs0"][0], out["obs0"][K - 1])


@pytest.mark.parametrize("A,L", [(64, 64), (4, 9)])
def test_done_bound_and_speed_limit(A, L):
    """A world's own done flag (any agent beyond the bound, environment.py:110-113,143-146) and Entity.max_speed
    (core.py:199-203) in the streaming kernel."""
    import mpe_hip
    spec = mpe_hip.make_spec("simple_spread", n_agents=A, n_landmarks=L)
    spec = dataclasses.replace(spec, max_speed=np.where(np.arange(A) % 2 == 0, 0.4, np.inf))
    B, T = 40, 30
    bound = 1.05 if A < 64 else 4.0        # 64 overlapping agents push one another out of the arena within a step or two
    stepper, roller = _env(A, L, B, spec=spec, done_bound=bound), _env(A, L, B, spec=spec, done_bound=bound)
    for e in (stepper, roller):
        e.reset(seed=8, observe=False)
    a = _actions(T, B, A, 4)
    a[:, :, 0] = 2          # agent 0 runs to the right until it is out
    out = roller.rollout(a)
    for t in range(T):
        obs, rew, done, _ = stepper.step(a[t])
        assert torch.equal(obs, out["obs"][t]) and torch.equal(rew, out["rew"][t]) and torch.equal(done, out["done"][t]), t
    _same_state(stepper, roller)
    assert bool(out["done"].any()) and not bool(out["done"].all())
    speed = stepper.vel.view(B, A, 2).norm(dim=-1)
    assert float(speed[:, 0::2].max()) <= 0.4 * (1 + 1e-6)


def test_out_of_range_actions_move_nothing():
    """environment.py:176-195: an action index outside 1..4 leaves action.u at zero, also as a staged byte."""
    A, L, B, T = 10, 7, 16, 4
    good, odd = _env(A, L, B), _env(A, L, B)
    for e in (good, odd):
        e.reset(seed=1, observe=False)
    a = _actions(T, B, A, 5)
    b = a.clone()
    b[a == 0] = torch.tensor([256, -1, 5, 1 << 20], dtype=torch.int32, device="cuda:0")[torch.arange(int((a == 0).sum()), device="cuda:0") % 4]
    o1, o2 = good.rollout(a), odd.rollout(b)
    assert torch.equal(o1["obs"], o2["obs"]) and torch.equal(o1["rew"], o2["rew"])


@pytest.mark.parametrize("B", [1300, 3100])
def test_worlds_split_unevenly_over_the_xcds(B):
    """From six worlds per CU on, a launch of 64/64 worlds deals more worlds to the even XCDs than to the odd ones
    (workgroup index -> world is then a permutation with idle workgroups at the end of every XCD's share): every
    world must still be stepped exactly once, also when the batch does not fill the last shares."""
    A = L = 64
    T = 3
    stepper, roller = _env(A, L, B), _env(A, L, B)
    a = _actions(T, B, A, 6)
    out = roller.rollout(a, reset_seed=4, obs0=True)
    assert torch.equal(stepper.reset(seed=4), out["obs0"][0])
    for t in range(T):
        obs, rew, done, _ = stepper.step(a[t])
        assert torch.equal(obs, out["obs"][t]) and torch.equal(rew, out["rew"][t]) and torch.equal(done, out["done"][t]), t
    _same_state(stepper, roller)


def test_stream_launch_under_graph_capture():
    """The launch is one kernel on the caller's stream: it can be captured and replayed."""
    A, L, B, T = 10, 7, 64, 6
    env, ref = _env(A, L, B), _env(A, L, B)
    a = _actions(T, B, A, 7)
    want = ref.rollout(a, reset_seed=9, obs0=True)
    out = env.rollout(a, reset_seed=9, obs0=True)          # allocates the outputs
    for v in out.values():
        if v is not None:
            v.zero_()
    env._episode = 0
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            env.rollout(a, out=out, reset_seed=9, obs0=True)
    g.replay()
    torch.cuda.synchronize()
    for key in ("obs0", "obs", "rew", "done"):
        assert torch.equal(out[key], want[key]), key


def test_uneven_deal_on_a_second_stream_after_other_work():
    """The XCD that a launch's workgroup 0 lands on changes with the history of the streams (0 in a fresh process); the
    deal takes it from a one-workgroup launch on the same stream right before the kernel, and is a bijection for any
    value of it: a launch on a second stream, after work on the first, still steps every world exactly once."""
    A = L = 64
    B, T = 1600, 2
    stepper, roller = _env(A, L, B), _env(A, L, B)
    a = _actions(T, B, A, 8)
    first = stepper.reset(seed=6).clone()
    want = [tuple(x.clone() for x in stepper.step(a[t])[:3]) for t in range(T)]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for _ in range(3):      # alternate streams so that the launches travel through different hardware queues
        with torch.cuda.stream(side):
            out = roller.rollout(a, reset_seed=6, obs0=True)
        side.synchronize()
        roller._episode = 0
        assert torch.equal(out["obs0"][0], first)
        for t in range(T):
            assert torch.equal(out["obs"][t], want[t][0]) and torch.equal(out["rew"][t], want[t][1]) and torch.equal(out["done"][t], want[t][2]), t
        (torch.randn(1 << 20, device="cuda:0") * 2).sum().item()


_DIGESTS = r"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "multiagent-particle-envs_amd"))
import torch, mpe_hip
def digest(scenario, B, T, **kw):
    env = mpe_hip.BatchedEnv(scenario, B, dtype=torch.float32, device="cuda:0", **kw)
    a = torch.randint(0, 5, (T, B, env.n_act), dtype=torch.int32, device="cuda:0", generator=torch.Generator("cuda:0").manual_seed(B))
    out = env.rollout(a, reset_seed=3, obs0=True)
    h = hashlib.sha256()
    for k in ("obs0", "obs", "rew", "done"):
        h.update(out[k].cpu().numpy().tobytes())
    h.update(env.pos.cpu().numpy().tobytes())
    return h.hexdigest()
print(digest("simple_spread", 1600, 2, n_agents=64, n_landmarks=64))
print(digest("simple_spread", 8192 * 64 + 64, 2))
print(digest("consensus", 65536 + 32, 2))
"""


def test_results_do_not_depend_on_the_deal():
    """MPE_XCD_SPLIT=0 switches the uneven deal over the XCDs off for a process: the three kinds of launch it applies to
    must produce the same bytes either way (it only decides which workgroup steps which world)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for split in ("1", "0"):
        env = dict(os.environ, MPE_XCD_SPLIT=split)
        r = subprocess.run([sys.executable, "-c", _DIGESTS, root], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([line for line in r.stdout.split() if len(line) == 64])
    assert len(outs[0]) == 3 and outs[0] == outs[1]


@pytest.mark.parametrize("A,L", [(64, 64), (10, 7)])
def test_episode_launch_without_the_reset_observation(A, L):
    """The reset state's observation is optional (obs0 = NULL at the C ABI): the steps' outputs are the same."""
    B, T = 20, 5
    with_obs0, without = _env(A, L, B, info=True), _env(A, L, B, info=True)
    a = _actions(T, B, A, 9)
    o1 = with_obs0.rollout(a, reset_seed=12, obs0=True)
    o2 = without.rollout(a, reset_seed=12)
    for key in ("obs", "rew", "done", "info_i", "info_f"):
        assert torch.equal(o1[key], o2[key]), key
    _same_state(with_obs0, without)
