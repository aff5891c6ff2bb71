"""Trainer-side glue around the batched simulator: what the reference's training loop does between
the env and the policy update (multiagent/backup/mappo.py:81-129, multiagent/backup/runner_old.py).

* ``BatchedVecEnv`` gives ``BatchedEnv`` the vectorised-env surface the reference's runner drives
  (``num_envs``, ``reset()``, ``step(actions) -> obs, rewards, dones, infos``; runner_old.py:9-16,44-56):
  the B worlds in HBM take the place of the ``SubprocVecEnv`` of B Python processes.
* ``gae`` is the runner's advantage scan (runner_old.py:58-80) as one kernel over all envs.
"""
import ctypes as C

import torch

from . import _lib
from .batch import BatchedEnv
from . import spaces


def gae(rewards, values, dones, last_values, last_dones=None, gamma=0.99, lam=0.95, reference=False, out=None):
    """Advantages and returns of a rollout.  ``rewards`` / ``values`` float32 ``[T, B, A]``, ``dones``
    uint8 / bool ``[T, B]`` (the flag step t returned), ``last_values`` float32 ``[B, A]``, all on one
    GPU.  ``reference=True`` reproduces runner_old.py:69-80 as written (one accumulator chained through
    the agent loop, the mask of step t read from ``dones[t+1]``, ``last_dones`` for the final step);
    the default scans every agent independently with the mask of step t read from ``dones[t]``.
    Returns ``(adv, ret)`` float32 ``[T, B, A]``."""
    if rewards.dtype != torch.float32 or values.dtype != torch.float32 or last_values.dtype != torch.float32:
        raise ValueError("rewards, values and last_values must be float32 (the runner casts them, runner_old.py:61-62)")
    if rewards.device.type != "cuda":
        raise ValueError("gae needs GPU tensors; there is no CPU path")
    T, B, A = rewards.shape
    rewards, values, last_values = rewards.contiguous(), values.contiguous(), last_values.contiguous()
    dones = dones.to(torch.uint8).contiguous()
    last_dones = dones[T - 1] if last_dones is None else last_dones.to(torch.uint8).contiguous()
    if values.shape != rewards.shape or tuple(dones.shape) != (T, B) or tuple(last_values.shape) != (B, A) or tuple(last_dones.shape) != (B,):
        raise ValueError("shape mismatch: rewards/values [T, B, A], dones [T, B], last_values [B, A], last_dones [B]")
    if out is None:
        out = (torch.empty_like(rewards), torch.empty_like(rewards))
    adv, ret = out
    lib = _lib.load()
    p = lambda t: C.c_void_p(t.data_ptr())
    dev = rewards.device.index if rewards.device.index is not None else torch.cuda.current_device()
    _lib.check(None, lib.mpe_gae(dev, _lib.GAE_REFERENCE if reference else _lib.GAE_STANDARD, B, A, T, p(rewards), p(values), p(dones),
                                 p(last_values), p(last_dones), float(gamma), float(lam), p(adv), p(ret),
                                 C.c_void_p(torch.cuda.current_stream(rewards.device).cuda_stream)))
    return adv, ret


class BatchedVecEnv:
    """``BatchedEnv`` behind the vectorised-env interface of the reference's runner.

    ``reset()`` returns the observations ``[num_envs, obs_total]`` (``obs_n`` gives the per-agent
    views); ``step(actions)`` takes int actions ``[num_envs, n_act]`` and returns
    ``(obs, rewards [num_envs, A], dones [num_envs] bool, infos)``.  Episodes end after
    ``max_episode_steps`` steps (the time limit the reference leaves to gym's registration,
    multiagent/__init__.py:9); an env that is done is reset on the device inside the same launch and
    ``obs`` already holds its first observation, which is what the runner's loop needs
    (runner_old.py:50-56).  ``infos`` carries ``last_return`` (the finished episodes' returns, what
    baselines' Monitor reports) and the benchmark tensors when ``info=True``.  Tensors are reused
    between calls."""

    def __init__(self, scenario, num_envs, max_episode_steps=25, seed=0, dtype=torch.float32, device="cuda:0", info=False, env_base=0,
                 done_bound=None, **spec_kw):
        # env_base: global index of this process's first env (rank r of a G-process job with num_envs envs each passes
        # r * num_envs and the job's one seed, as the reference's vec-env factory derives each env's seed from its
        # global position, multiagent/backup/mappo.py:97,113)
        self.env = BatchedEnv(scenario, num_envs, dtype=dtype, device=device, info=info, max_episode_steps=max_episode_steps, seed=seed,
                              env_base=env_base, done_bound=done_bound, **spec_kw)
        self.num_envs = int(num_envs)
        self.n_agents = self.env.n_agents
        self.seed = int(seed)
        self.action_space = [spaces.Discrete(n) for n in self.env.action_dims]
        self.observation_space = [spaces.Box(low=-float("inf"), high=float("inf"), shape=(d,), dtype="float32") for d in self.env.obs_dims]
        self._actions = None

    @property
    def obs_n(self):
        return self.env.obs_n

    def reset(self):
        return self.env.reset(seed=self.seed)

    def step_async(self, actions):
        self._actions = actions

    def step_wait(self):
        obs, rew, done, (info_i, info_f) = self.env.step(self._actions)
        infos = {"last_return": self.env.last_return, "episode_count": self.env.episode_count, "info_i": info_i, "info_f": info_f}
        return obs, rew, done.bool(), infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.env = None
