"""Batched particle worlds resident in HBM, stepped by the HIP kernels of libmpe_hip.so.

``BatchedEnv`` is the vectorised counterpart of the reference's ``MultiAgentEnv``
(multiagent/environment.py:15-127): ``reset()`` and ``step(actions)`` act on ``n_envs``
independent worlds at once.  All state and outputs are PyTorch tensors on one GPU; the library
receives their raw device pointers and the current HIP stream, so nothing is copied.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .spec import WorldSpec, make_spec


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class BatchedEnv:
    """n_envs worlds of one scenario.

    Tensors (T = float32 or float64):
      pos   [2E, B]  env-minor entity coordinates (agents first)     vel  [2A, B]
      comm  [S*dim_c, B] comm state of the S speaking agents (or None)
      aux   [n_aux, B] int32: the landmark indices reset_world draws (goal landmark; goal_b of both agents in
            simple_reference; goal and key in simple_crypto), or the state simple_predator_prey's post_step
            rewrites (row 0: bit f = food f alive, bit 3+k = member k holds food; rows 1-2: respawn stream), or None
      (worlds simulated by a whole wave -- ``env_major`` is True -- store these transposed:
      pos [B, 2E], vel [B, 2A], comm [B, S*dim_c]; use get_state / set_state to stay layout-agnostic)
      obs   [B, obs_total]; ``obs_n[i]`` is the view of agent i's columns
      rew   [B, A]    done [B] uint8    info_i [B, n_info_i] int32    info_f [B, n_info_f]
    Actions are int32 ``[B, n_act]``: per agent its move 0..4 if it is movable, then its comm word if it speaks.
    In a collaborative world (simple_speaker_listener, simple_reference) ``rew`` holds the shared reward as
    ``MultiAgentEnv.step`` returns it; the agents' own rewards are in ``info_f``.
    """

    def __init__(self, scenario, n_envs, dtype=torch.float32, device="cuda:0", info=False, max_episode_steps=None, seed=0,
                 scripted=None, env_base=0, done_bound=None, post_step=True, **spec_kw):
        self.spec = scenario if isinstance(scenario, WorldSpec) else make_spec(scenario, **spec_kw)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("BatchedEnv needs a GPU device (got %s); there is no CPU path" % (device,))
        if dtype not in (torch.float32, torch.float64):
            raise ValueError("dtype must be torch.float32 or torch.float64")
        self.dtype = dtype
        self.n_envs = int(n_envs)
        self._lib = _lib.load()
        cspec, self._keep = self.spec.to_c()
        h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(None, self._lib.mpe_create(C.byref(cspec), self.n_envs, _lib.MPE_F64 if dtype == torch.float64 else _lib.MPE_F32,
                                              dev_index, C.byref(h)))
        self._h = h
        # Position of this batch's first env in the whole job's env index (a shard of a larger batch: rank r of G
        # holds envs [r * B, (r + 1) * B) and passes env_base = r * B).  Every on-device draw is keyed by env_base + b,
        # so the shards of a job draw exactly the worlds of the unsharded batch under the same seed (the reference
        # seeds env k of a vectorised run by its global position, multiagent/backup/mappo.py:97,113).
        self.env_base = int(env_base)
        if self.env_base:
            _lib.check(h, self._lib.mpe_set_env_base(h, self.env_base))
        # done_callback of the batch (multiagent/environment.py:143-146 combined by any(done_n), :110-113): an env is
        # done when an agent is outside [-done_bound, done_bound]^2.  None = the reference's default, no done callback.
        self.done_bound = None if done_bound is None else float(done_bound)
        if self.done_bound is not None:
            _lib.check(h, self._lib.mpe_set_done_bound(h, self.done_bound))
        # post_step=False: the scenario's post_step is left out of the step, as in MultiAgentEnv(post_step_callback=None)
        # (what the reference's make_env builds, make_env.py:36-43); only scenarios that have one take the switch
        self.post_step_enabled = bool(post_step)
        if not self.post_step_enabled:
            _lib.check(h, self._lib.mpe_set_post_step(h, 0))
        lo = _lib.MpeLayout()
        _lib.check(h, self._lib.mpe_layout(h, C.byref(lo)))
        self.layout = lo
        A, L, B = lo.n_agents, lo.n_landmarks, self.n_envs
        self.n_agents, self.n_landmarks, self.n_act = A, L, lo.n_act
        # number of choices of every action column, in the reference's action_n order
        self.action_dims = []
        for i in range(A):
            self.action_dims += ([5] if self.spec.movable[i] else []) + ([lo.dim_c] if not self.spec.silent[i] else [])
        self.obs_dims = [lo.obs_dim[i] for i in range(A)]
        self.obs_offsets = [lo.obs_offset[i] for i in range(A)]
        # Scripted agents (Agent.action_callback, multiagent/core.py:144-147: World.step asks the callback for the
        # agent's action, and MultiAgentEnv neither counts the agent in env.n nor takes an action for it,
        # core.py:136-139, environment.py:21-23).  Batched form: scripted = {agent index: fn(env) -> int32 [B] or
        # [B, k]}, the agent's action column(s) for every world, evaluated on the device tensors right before each
        # step; step() then takes the policy agents' columns only.
        self.scripted = dict(scripted or {})
        cols, c = [], 0
        for i in range(A):
            k = (1 if self.spec.movable[i] else 0) + (0 if self.spec.silent[i] else 1)
            cols.append(list(range(c, c + k)))
            c += k
        self.agent_cols = cols
        for i in self.scripted:
            if not 0 <= i < A or not cols[i]:
                raise ValueError("scripted agent %r does not exist or takes no action" % (i,))
        self.policy_cols = [c for i in range(A) if i not in self.scripted for c in cols[i]]
        self.n_policy_act = len(self.policy_cols)
        self._act_full = None
        # Every persistent buffer is a view into ONE zero-initialised device allocation (the arena), each on
        # a 256-byte boundary: inputs first (state, aux and -- for a one-world batch, whose caller stages
        # host data -- the action / force / comm-action slots), then the outputs.  A one-world caller can
        # then move all inputs with one copy and all results with another (mpe_hip/oneworld.py).
        self.env_major = bool(lo.state_env_major)
        st = (lambda rows: (B, rows)) if self.env_major else (lambda rows: (rows, B))
        comm_rows = lo.n_speakers * lo.dim_c + lo.n_hist_rows       # consensus keeps its history ring in the comm buffer
        if info and lo.n_info_i == 0:
            raise ValueError("this scenario has no benchmark_data: info is not available")
        self.max_episode_steps = max_episode_steps
        self.seed = int(seed)
        self._episode = 0
        self._auto_since_reset = False
        i32, u8 = torch.int32, torch.uint8
        fields = [("pos", st(2 * (A + L)), dtype), ("vel", st(2 * A), dtype)]
        fields += [("comm", st(comm_rows), dtype)] if comm_rows > 0 else []
        fields += [("aux", (lo.n_aux, B), i32)] if lo.n_aux > 0 else []
        if B == 1:
            fields += [("act_buf", (1, max(1, lo.n_act)), i32), ("u_buf", (1, 2 * A), dtype)]
            fields += [("c_buf", (1, lo.n_speakers * lo.dim_c), dtype)] if lo.n_speakers > 0 else []
        self._n_input_fields = len(fields)
        fields += [("obs", (B, lo.obs_total), dtype), ("rew", (B, A), dtype), ("done", (B,), u8)]
        if info:
            fields += [("info_i", (B, lo.n_info_i), i32)]
            fields += [("info_f", (B, lo.n_info_f), dtype)] if lo.n_info_f > 0 else []
        if max_episode_steps is not None:
            # episode bookkeeping on device (mpe_step_auto): time-limit truncation + auto-reset + returns
            fields += [("step_count", (B,), i32), ("episode_count", (B,), i32), ("ep_return", (B, A), dtype), ("last_return", (B, A), dtype)]
        esize = {torch.float32: 4, torch.float64: 8, i32: 4, u8: 1}
        self.arena_layout, off = {}, 0
        for name, shape, dt in fields:
            nbytes = int(np.prod(shape)) * esize[dt]
            self.arena_layout[name] = (off, nbytes, tuple(shape), dt)
            off = (off + nbytes + 255) // 256 * 256
        self.arena = torch.zeros(max(off, 256), dtype=u8, device=self.device)
        self.arena_input_bytes = max((o + n for k, (o, n, _, _) in list(self.arena_layout.items())[:self._n_input_fields]), default=0)
        for name in ("comm", "aux", "info_i", "info_f"):
            setattr(self, name, None)
        for name, (o, n, shape, dt) in self.arena_layout.items():
            setattr(self, name, self.arena[o:o + n].view(dt).view(*shape))
        # raw device pointers of the persistent buffers, resolved once (the tensors are never reallocated)
        self._p = {k: _ptr(getattr(self, k, None)) for k in ("pos", "vel", "comm", "aux", "obs", "rew", "done", "info_i", "info_f",
                                                             "step_count", "episode_count", "ep_return", "last_return")}

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.mpe_destroy(h)
            self._h = None

    # ------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def obs_n(self):
        """Per-agent observation views ``[B, obs_dim[i]]`` (the reference's obs_n list)."""
        return [self.obs[:, o:o + d] for o, d in zip(self.obs_offsets, self.obs_dims)]

    def observe(self, score=False):
        """Scenario.observation for every agent of every env on the current state; with ``score``
        also Scenario.reward (into ``rew``) and benchmark_data (into ``info_i`` / ``info_f``)."""
        p = self._p
        _lib.check(self._h, self._lib.mpe_observe(self._h, p["pos"], p["vel"], p["comm"], p["aux"], p["obs"], p["rew"] if score else None,
                                                  p["info_i"] if score else None, p["info_f"] if score else None, self._stream()))
        return self.obs

    def reset(self, seed=0, mask=None, observe=True):
        """Scenario.reset_world on device (all envs, or those with a non-zero mask byte).  Returns
        the fresh observations like MultiAgentEnv.reset unless ``observe`` is False."""
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        if self.max_episode_steps is not None and mask is not None and self._auto_since_reset and int(seed) != self.seed:
            # mpe_step_auto takes ONE seed for the whole batch: envs left out of this reset would otherwise draw their
            # later automatic resets from a seed other than the one their episode numbering started under
            raise ValueError("a masked reset under max_episode_steps keeps the batch's seed (%d); reset every env to change it" % self.seed)
        if self.max_episode_steps is not None and self._auto_since_reset:
            # automatic resets on the device have drawn episodes up to max(episode_count) of the seed: start
            # beyond them, so that no env replays a world it (or a neighbour) has already played
            self._episode = max(self._episode, int(self.episode_count.max().item()) + 1)
        p = self._p
        _lib.check(self._h, self._lib.mpe_reset(self._h, int(seed), self._episode, _ptr(mask), p["pos"], p["vel"], p["comm"],
                                                p["aux"], self._stream()))
        if self.max_episode_steps is not None:
            # the device-side counters restart from this reset: later automatic resets draw episodes
            # self._episode + 1, + 2, ... of the same seed
            self.seed = int(seed)
            keep = None if mask is None else (mask == 0)
            for t, v in ((self.step_count, 0), (self.episode_count, self._episode), (self.ep_return, 0)):
                if keep is None:
                    t.fill_(v)
                else:
                    t[~keep] = v
            self._auto_since_reset = True
        self._episode += 1
        return self.observe() if observe else None

    def _actions(self, actions, lead):
        a = actions
        if not (torch.is_tensor(a) and a.dtype == torch.int32 and a.device == self.device and a.is_contiguous()):
            a = torch.as_tensor(actions, device=self.device).to(torch.int32).contiguous()
        if self.scripted:
            if lead:
                raise ValueError("a multi-step launch takes pre-computed actions; scripted agents act on the state of every step")
            if tuple(a.shape) != (self.n_envs, self.n_policy_act):
                raise ValueError("actions must have shape %s (policy agents only), got %s" % ((self.n_envs, self.n_policy_act), tuple(a.shape)))
            if self._act_full is None:
                self._act_full = torch.zeros(self.n_envs, self.n_act, dtype=torch.int32, device=self.device)
                self._policy_idx = torch.tensor(self.policy_cols, dtype=torch.long, device=self.device)
            full = self._act_full
            full[:, self._policy_idx] = a
            for i, fn in self.scripted.items():
                full[:, self.agent_cols[i]] = torch.as_tensor(fn(self), device=self.device).to(torch.int32).reshape(self.n_envs, -1)
            return full
        want = tuple(lead) + (self.n_envs, self.n_act)
        if tuple(a.shape) != want:
            raise ValueError("actions must have shape %s, got %s" % (want, tuple(a.shape)))
        return a

    def step(self, actions):
        """MultiAgentEnv.step for the whole batch.  Returns (obs, rew, done, info) tensors that are
        overwritten by the next call."""
        a = self._actions(actions, ())
        p = self._p
        if self.max_episode_steps is not None:
            # envs that reach max_episode_steps are truncated (done = 1) and reset on device; obs is then the new
            # episode's first observation, last_return the finished episode's return
            _lib.check(self._h, self._lib.mpe_step_auto(self._h, _ptr(a), p["pos"], p["vel"], p["comm"], p["aux"], p["obs"], p["rew"], p["done"],
                                                        p["info_i"], p["info_f"], p["step_count"], p["episode_count"], p["ep_return"],
                                                        p["last_return"], int(self.max_episode_steps), self.seed, self._stream()))
        else:
            _lib.check(self._h, self._lib.mpe_step(self._h, _ptr(a), p["pos"], p["vel"], p["comm"], p["aux"], p["obs"], p["rew"], p["done"],
                                                   p["info_i"], p["info_f"], self._stream()))
        return self.obs, self.rew, self.done, (self.info_i, self.info_f)

    def rollout(self, actions, out=None, reset_seed=None, episodes=1, obs0=False):
        """``T`` steps in one launch for pre-computed actions ``[T, B, n_act]``.  Returns a dict of
        ``[T, ...]`` tensors (pass the previous result as ``out`` to reuse its buffers).  With
        ``reset_seed`` every env is first reset on device inside the same launch (one episode per
        launch, like ``reset(seed, observe=False)`` followed by ``rollout``); with ``episodes`` = K > 1
        the T steps are K consecutive episodes of T / K steps, each starting with its own reset.  With
        ``obs0`` the launch also returns the observation of every reset state, ``out["obs0"]``
        ``[K, B, obs_total]`` (what ``reset()`` returns)."""
        T = int(actions.shape[0])
        a = self._actions(actions, (T,))
        lo, B = self.layout, self.n_envs
        if obs0 and reset_seed is None:
            raise ValueError("obs0 is the observation of an on-device reset: pass reset_seed")
        if out is None or out["obs"].shape[0] != T or (obs0 and (out.get("obs0") is None or out["obs0"].shape[0] != episodes)):
            z = lambda *shape, dt=self.dtype: torch.empty(*shape, dtype=dt, device=self.device)
            out = {"obs": z(T, B, lo.obs_total), "rew": z(T, B, self.n_agents), "done": z(T, B, dt=torch.uint8),
                   "info_i": z(T, B, lo.n_info_i, dt=torch.int32) if self.info_i is not None else None,
                   "info_f": z(T, B, lo.n_info_f) if self.info_f is not None else None,
                   "obs0": z(int(episodes), B, lo.obs_total) if obs0 else None}
        first = _ptr(out.get("obs0")) if obs0 else None
        p = self._p
        if reset_seed is None:
            _lib.check(self._h, self._lib.mpe_rollout(self._h, _ptr(a), p["pos"], p["vel"], p["comm"], p["aux"], _ptr(out["obs"]),
                                                      _ptr(out["rew"]), _ptr(out["done"]), _ptr(out["info_i"]), _ptr(out["info_f"]), T,
                                                      self._stream()))
        elif episodes > 1:
            if T % episodes:
                raise ValueError("%d steps do not split into %d episodes" % (T, episodes))
            _lib.check(self._h, self._lib.mpe_episodes(self._h, int(reset_seed), self._episode, int(episodes), T // episodes, _ptr(a), p["pos"],
                                                       p["vel"], p["comm"], p["aux"], _ptr(out["obs"]), _ptr(out["rew"]), _ptr(out["done"]),
                                                       _ptr(out["info_i"]), _ptr(out["info_f"]), first, self._stream()))
            self._episode += int(episodes)
        else:
            _lib.check(self._h, self._lib.mpe_episode(self._h, int(reset_seed), self._episode, _ptr(a), p["pos"], p["vel"], p["comm"],
                                                      p["aux"], _ptr(out["obs"]), _ptr(out["rew"]), _ptr(out["done"]), _ptr(out["info_i"]),
                                                      _ptr(out["info_f"]), first, T, self._stream()))
            self._episode += 1
        return out

    def set_noise(self, seed=0, noise_u=None, noise_c=None):
        """Source of the standard-normal draws of ``Agent.u_noise`` / ``Agent.c_noise`` (multiagent/core.py:165,211) for a
        spec that sets them.  Without arrays: the device generator keyed by ``seed``, restarted at step 0.  With
        ``noise_u`` [T, B, A, 2] (and ``noise_c`` [T, B, A, dim_c] for speaking agents; T may be omitted for a single
        step): the draws of the NEXT launch only, e.g. the reference's own, after which the generator takes over."""
        draws = None
        if noise_u is not None:
            A, B, lo = self.n_agents, self.n_envs, self.layout
            nu = torch.as_tensor(np.asarray(noise_u), dtype=self.dtype).reshape(-1, B, 2 * A)
            T = nu.shape[0]
            parts = [nu]
            if lo.n_speakers > 0:
                nc = torch.zeros(T, B, A, lo.dim_c, dtype=self.dtype) if noise_c is None else \
                    torch.as_tensor(np.asarray(noise_c), dtype=self.dtype).reshape(T, B, A, lo.dim_c)
                speakers = [i for i in range(A) if not self.spec.silent[i]]
                parts.append(nc[:, :, speakers, :].reshape(T, B, -1))
            row = torch.cat(parts, dim=2)
            if row.shape[2] % 2:
                row = torch.cat([row, torch.zeros(T, B, 1, dtype=self.dtype)], dim=2)
            draws = self._noise_draws = row.contiguous().to(self.device)        # kept alive until the next call
        _lib.check(self._h, self._lib.mpe_set_noise(self._h, int(seed), _ptr(draws)))

    def post_step(self):
        """Scenario.post_step alone (simple_predator_prey, collector): pick-ups, hand-overs and respawns on the
        current state, as ``step`` applies them after the callbacks.  Rewrites landmark positions and ``aux``."""
        p = self._p
        _lib.check(self._h, self._lib.mpe_post_step(self._h, p["pos"], p["aux"], self._stream()))

    # ------------------------------------------------------------------
    def set_state(self, pos, vel, comm=None, goal=None):
        """Inject a state given env-major: pos [B, E, 2], vel [B, A, 2], comm [B, S*dim_c], and the landmark
        indices reset_world draws, [B] or [B, n_aux], of scenarios that have them (kept when None)."""
        B, A, L = self.n_envs, self.n_agents, self.n_landmarks
        lay = (lambda x: x.contiguous()) if self.env_major else (lambda x: x.t().contiguous())
        pos = torch.as_tensor(np.asarray(pos), dtype=self.dtype).reshape(B, 2 * (A + L))
        vel = torch.as_tensor(np.asarray(vel), dtype=self.dtype).reshape(B, 2 * A)
        self.pos.copy_(lay(pos))
        self.vel.copy_(lay(vel))
        if self.comm is not None:
            if comm is None:
                self.comm.zero_()
            else:
                self.comm.copy_(lay(torch.as_tensor(np.asarray(comm), dtype=self.dtype).reshape(B, -1)))
        if goal is not None:
            if self.aux is None:
                raise ValueError("this scenario has no goal landmark")
            self.aux.copy_(torch.as_tensor(np.asarray(goal), dtype=torch.int32).reshape(B, self.aux.shape[0]).t())

    def get_state(self):
        """Returns env-major numpy copies (pos [B, E, 2], vel [B, A, 2], comm [B, S*dim_c] or None)."""
        B, A, L = self.n_envs, self.n_agents, self.n_landmarks
        lay = (lambda x: x) if self.env_major else (lambda x: x.t())
        pos = lay(self.pos).reshape(B, A + L, 2).cpu().numpy()
        vel = lay(self.vel).reshape(B, A, 2).cpu().numpy()
        comm = lay(self.comm).cpu().numpy() if self.comm is not None else None
        return pos, vel, comm

    _STATE_FIELDS = ("pos", "vel", "comm", "aux", "step_count", "episode_count", "ep_return", "last_return")

    def _fingerprint(self):
        """sha256 over everything a resume depends on besides the tensors: the world description (every spec array and
        scalar), the time limit, the state layout and the batch's place in the job's env index."""
        import dataclasses
        import hashlib
        h = hashlib.sha256()
        for f in dataclasses.fields(self.spec):
            v = getattr(self.spec, f.name)
            h.update(f.name.encode() + b"=")
            h.update(np.asarray(v, dtype=np.float64).tobytes() if isinstance(v, np.ndarray) else repr(v).encode())
        h.update(repr((self.max_episode_steps, self.env_major, self.env_base, self.done_bound, self.post_step_enabled)).encode())
        return h.hexdigest()[:32]

    def _noise_state(self, put=None):
        seed, step = C.c_uint64(0 if put is None else put[0]), C.c_uint64(0 if put is None else put[1])
        _lib.check(self._h, self._lib.mpe_noise_state(self._h, 0 if put is None else 1, C.byref(seed), C.byref(step)))
        return int(seed.value), int(step.value)

    def state_dict(self):
        """Checkpoint of the batch: copies (on the device) of everything its future depends on -- positions,
        velocities, comm state or history ring, the scenarios' int state, the episode bookkeeping of
        ``max_episode_steps``, the position of the device's noise generator -- plus the episode numbering of the
        resets.  ``load_state_dict`` on an env built with the same arguments resumes bit for bit (an env built with
        other sizes, noise levels, time limit or ``env_base`` is refused); the dict holds tensors, ints and strings
        only (``torch.save`` takes it).  Draw arrays handed to ``set_noise`` for the next launch are not part of it."""
        return {"scenario": self.spec.name, "n_envs": self.n_envs, "dtype": str(self.dtype), "env_major": self.env_major,
                "episode": self._episode, "seed": self.seed, "auto_since_reset": self._auto_since_reset,
                "max_episode_steps": -1 if self.max_episode_steps is None else int(self.max_episode_steps),
                "env_base": self.env_base, "fingerprint": self._fingerprint(), "noise": list(self._noise_state()),
                "tensors": {k: getattr(self, k).clone() for k in self._STATE_FIELDS if getattr(self, k, None) is not None}}

    def load_state_dict(self, sd):
        """Restore a ``state_dict`` taken from an env of the same scenario, world description, size, dtype, time
        limit and ``env_base``."""
        have = [k for k in self._STATE_FIELDS if getattr(self, k, None) is not None]
        if (sd["scenario"], sd["n_envs"], sd["dtype"]) != (self.spec.name, self.n_envs, str(self.dtype)) or sorted(sd["tensors"]) != sorted(have):
            raise ValueError("checkpoint of %s x %d (%s, fields %s) does not fit this env: %s x %d (%s, fields %s)"
                             % (sd["scenario"], sd["n_envs"], sd["dtype"], sorted(sd["tensors"]), self.spec.name, self.n_envs, self.dtype, sorted(have)))
        mine = {"max_episode_steps": -1 if self.max_episode_steps is None else int(self.max_episode_steps), "env_major": self.env_major,
                "env_base": self.env_base, "fingerprint": self._fingerprint()}
        for k, v in mine.items():
            if k in sd and sd[k] != v:
                raise ValueError("checkpoint was taken from an env with another %s (%r, this env: %r): it would not resume bit for bit"
                                 % (k, sd[k], v))
        for k in have:
            t = sd["tensors"][k]
            if tuple(t.shape) != tuple(getattr(self, k).shape):
                raise ValueError("checkpoint field %s has shape %s, expected %s" % (k, tuple(t.shape), tuple(getattr(self, k).shape)))
            getattr(self, k).copy_(t)
        self._episode, self.seed, self._auto_since_reset = int(sd["episode"]), int(sd["seed"]), bool(sd["auto_since_reset"])
        if "noise" in sd:
            self._noise_state(put=(int(sd["noise"][0]), int(sd["noise"][1])))

    def set_history(self, his_pos, his_vel):
        """consensus: the agents' last ten positions and velocities, [B, A, 10, 2] each, oldest first
        (state.p_his_pos / p_his_vel); stored as the history ring with its head at slot 0."""
        if self.layout.n_hist_rows == 0:
            raise ValueError("this scenario keeps no history")
        B, A = self.n_envs, self.n_agents
        h = np.concatenate([np.asarray(his_pos, np.float64), np.asarray(his_vel, np.float64)], axis=3)     # [B, A, 10, 4]
        rows = torch.as_tensor(h.transpose(0, 2, 1, 3).reshape(B, -1), dtype=self.dtype)                  # (slot, agent, component)
        self.comm.copy_(rows.t().contiguous())
        self.aux.zero_()

    def get_history(self):
        """consensus: (his_pos, his_vel) [B, A, 10, 2], oldest first."""
        B, A = self.n_envs, self.n_agents
        ring = self.comm.t().reshape(B, 10, A, 4).cpu().numpy()
        head = self.aux[0].cpu().numpy()
        order = (head[:, None] + np.arange(10)[None, :]) % 10
        h = ring[np.arange(B)[:, None], order].transpose(0, 2, 1, 3)
        return h[..., :2].copy(), h[..., 2:].copy()

    def to_world(self, b, world=None):
        """Env ``b`` of the batch as a ``World`` object graph (entities with size / mass / flags and the current
        ``state.p_pos`` / ``p_vel`` / ``c``), for code written against the reference's objects: a viewer
        (multiagent/environment.py:214-328 reads ``entity.state.p_pos``, ``size``, ``color``), a ``Scenario``'s Python
        callbacks, a debugger.  Pass the world of an earlier call to refresh it in place instead of building a new one.
        The copy is a snapshot: stepping the batch does not move it, and stepping it does not touch the batch."""
        from . import world as core, new_core
        sp, A, L = self.spec, self.n_agents, self.n_landmarks
        if not 0 <= b < self.n_envs:
            raise IndexError("env %d out of range for a batch of %d" % (b, self.n_envs))
        if world is None:
            mod = new_core if sp.new_core else core
            world = mod.World()
            world.dim_c = sp.dim_c
            world.dt, world.damping, world.contact_force, world.contact_margin = sp.dt, sp.damping, sp.contact_force, sp.contact_margin
            world.collaborative = bool(sp.collaborative)
            world.device = str(self.device)
            for i in range(A):
                a = mod.Agent()
                a.name, a.i = "agent %d" % i, i
                a.movable, a.collide, a.silent = bool(sp.movable[i]), bool(sp.collide[i]), bool(sp.silent[i])
                a.size, a.initial_mass = float(sp.size[i]), float(sp.mass[i])
                acc = sp.accel[i] if sp.accel is not None else sp.sensitivity[i]
                a.accel = None if (np.isnan(acc) or (sp.accel is None and acc == 5.0)) else float(acc)
                a.max_speed = None if np.isinf(sp.max_speed[i]) else float(sp.max_speed[i])
                a.adversary, a.leader = bool(sp.adversary[i]), bool(sp.leader[i])
                a.ghost = bool(sp.ghost[i]) if sp.ghost is not None else False
                a.color = np.array([0.85, 0.35, 0.35]) if a.adversary else np.array([0.35, 0.35, 0.85])
                a.action.u, a.action.c = np.zeros(2), np.zeros(sp.dim_c)
                world.agents.append(a)
            for l in range(L):
                lm = mod.Landmark()
                lm.name, lm.i = "landmark %d" % l, A + l
                lm.movable, lm.collide = False, bool(sp.collide[A + l])
                lm.size, lm.initial_mass = float(sp.size[A + l]), float(sp.mass[A + l])
                lm.color = np.array([0.25, 0.25, 0.25])
                world.landmarks.append(lm)
            world.walls = [core.Wall(orient=w["orient"], axis_pos=w["axis_pos"], endpoints=tuple(w["endpoints"]), width=w["width"], hard=w["hard"])
                           for w in sp.walls]
        elif len(world.agents) != A or len(world.landmarks) != L:
            raise ValueError("the world passed in does not have this batch's entities")
        one = (lambda t: t[b]) if self.env_major else (lambda t: t[:, b])
        pos = one(self.pos).cpu().numpy().astype(np.float64).reshape(A + L, 2)
        vel = one(self.vel).cpu().numpy().astype(np.float64).reshape(A, 2)
        n_c = self.layout.n_speakers * self.layout.dim_c
        comm = one(self.comm)[:n_c].cpu().numpy().astype(np.float64) if (self.comm is not None and n_c > 0) else None
        for e, p in zip(world.entities, pos):
            e.state.p_pos = p.copy()
            e.state.p_vel = np.zeros(2)
        world.unpack_state(pos[None], vel[None], None if comm is None else comm[None])
        if self.layout.n_hist_rows > 0:         # consensus: the sliding history, oldest first
            ring = self.comm[:, b].reshape(10, A, 4).cpu().numpy().astype(np.float64)
            head = int(self.aux[0, b])
            order = (head + np.arange(10)) % 10
            for i, a in enumerate(world.agents):
                a.state.p_his_pos, a.state.p_his_vel = ring[order, i, :2].copy(), ring[order, i, 2:].copy()
        if getattr(world, "cache_dists", False):
            world.calculate_distances()
        return world

    @property
    def goal(self):
        """Landmark indices drawn at reset: a [B] int32 view for scenarios with one goal landmark,
        [B, n_aux] for those with two."""
        if self.aux is None:
            return None
        return self.aux[0] if self.aux.shape[0] == 1 else self.aux.t()

    def world_step(self, u, c=None):
        """World.step alone (multiagent/core.py:142-158) for a physics-only handle (scenario id
        SCN_NONE): ``u`` [B, 2A] are the agents' already-scaled action forces (agent.action.u) and
        ``c`` [B, S*dim_c] the speakers' comm actions.  State is updated in place."""
        u = torch.as_tensor(u, dtype=self.dtype, device=self.device).reshape(self.n_envs, 2 * self.n_agents).contiguous()
        if self.comm is not None:
            c = torch.as_tensor(c, dtype=self.dtype, device=self.device).reshape(self.n_envs, -1).contiguous()
        else:
            c = None
        p = self._p
        _lib.check(self._h, self._lib.mpe_world_step(self._h, _ptr(u), _ptr(c), p["pos"], p["vel"], p["comm"], self._stream()))
