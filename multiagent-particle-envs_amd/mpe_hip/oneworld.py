"""One world on the GPU for the single-env facade: all inputs up in one copy, all results down in another.

``MultiAgentEnv`` and ``World.step`` keep the reference's contract that the Python objects are the state
(a caller may edit ``agent.state.p_pos`` between steps), so every call carries the state to the device and
the results back.  A one-world ``BatchedEnv`` lays its buffers out in one arena, inputs first; this class
mirrors the arena in pinned host memory and exposes numpy views of every field, so a step is: fill the
views, one host-to-device copy of the input prefix, the launch, one device-to-host copy of the arena, one
stream synchronisation.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .batch import BatchedEnv

_NP = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32, torch.uint8: np.uint8}


class OneWorld:
    def __init__(self, spec, device="cuda:0", info=False, post_step=True):
        self.env = env = BatchedEnv(spec, 1, dtype=torch.float64, device=device, info=info, post_step=post_step)
        self.host = torch.zeros(env.arena.numel(), dtype=torch.uint8).pin_memory()
        raw = self.host.numpy()
        self.v = {}                     # numpy views of the host mirror, same names and shapes as the device tensors
        for name, (o, n, shape, dt) in env.arena_layout.items():
            self.v[name] = raw[o:o + n].view(_NP[dt]).reshape(shape)
        self._in = env.arena_input_bytes
        self._stream = torch.cuda.current_stream(env.device)
        A, L = env.n_agents, env.n_landmarks
        self.n_state = (A + L, A)

    # ---- host views in world terms (a one-world batch: env-major and env-minor coincide up to shape) ----
    def put_state(self, pos, vel, comm=None, aux=None):
        """pos [E, 2] (or anything of that size), vel [A, 2], comm [S*dim_c] or history rows, aux [n_aux]."""
        v = self.v
        v["pos"].reshape(-1)[:] = np.asarray(pos, np.float64).reshape(-1)
        v["vel"].reshape(-1)[:] = np.asarray(vel, np.float64).reshape(-1)
        if "comm" in v:
            v["comm"].reshape(-1)[:] = 0.0 if comm is None else np.asarray(comm, np.float64).reshape(-1)
        if aux is not None:
            v["aux"].reshape(-1)[:] = np.asarray(aux, np.int64).reshape(-1)

    def _up(self):
        self.env.arena[:self._in].copy_(self.host[:self._in], non_blocking=True)

    def _down(self):
        self.host.copy_(self.env.arena, non_blocking=True)
        self._stream.synchronize()

    def _sp(self):
        return C.c_void_p(self._stream.cuda_stream)

    def _check_stream(self):
        cur = torch.cuda.current_stream(self.env.device)
        if cur != self._stream:
            self._stream = cur

    # ---- the launches ----
    def step(self, actions):
        """MultiAgentEnv.step for the staged state: actions = flat list of ints ([move][, comm] per agent)."""
        env, p = self.env, self.env._p
        self._check_stream()
        a = self.v["act_buf"].reshape(-1)
        a[:len(actions)] = actions
        self._up()
        _lib.check(env._h, env._lib.mpe_step(env._h, C.c_void_p(env.act_buf.data_ptr()), p["pos"], p["vel"], p["comm"], p["aux"], p["obs"],
                                             p["rew"], p["done"], p["info_i"], p["info_f"], self._sp()))
        self._down()

    def observe(self, score=True, upload=True):
        """The callbacks on the staged state, or with ``upload=False`` on the state the device already holds
        (right after a step)."""
        env, p = self.env, self.env._p
        self._check_stream()
        if upload:
            self._up()
        _lib.check(env._h, env._lib.mpe_observe(env._h, p["pos"], p["vel"], p["comm"], p["aux"], p["obs"], p["rew"] if score else None,
                                                p["info_i"] if score else None, p["info_f"] if score else None, self._sp()))
        self._down()

    def world_step(self, u, c=None):
        """World.step for the staged state; u [2A] scaled action forces, c [S*dim_c] comm actions."""
        env, p = self.env, self.env._p
        self._check_stream()
        self.v["u_buf"].reshape(-1)[:] = np.asarray(u, np.float64).reshape(-1)
        cp = None
        if "c_buf" in self.v:
            self.v["c_buf"].reshape(-1)[:] = np.asarray(c, np.float64).reshape(-1)
            cp = C.c_void_p(env.c_buf.data_ptr())
        self._up()
        _lib.check(env._h, env._lib.mpe_world_step(env._h, C.c_void_p(env.u_buf.data_ptr()), cp, p["pos"], p["vel"], p["comm"], self._sp()))
        self._down()

    def post_step(self):
        env, p = self.env, self.env._p
        self._check_stream()
        self._up()
        _lib.check(env._h, env._lib.mpe_post_step(env._h, p["pos"], p["aux"], self._sp()))
        self._down()

    # ---- results (views of the host mirror: copy what must outlive the next call) ----
    def state(self):
        """(pos [E, 2], vel [A, 2], comm flat or None) after the last launch."""
        E, A = self.n_state
        v = self.v
        return v["pos"].reshape(E, 2), v["vel"].reshape(A, 2), (v["comm"].reshape(-1) if "comm" in v else None)
