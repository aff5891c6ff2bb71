"""mpe_world_step_static, the entry point that allocates nothing (its per-entity table lives in a static
device array of the library), against mpe_world_step on a handle: same kernel, same inputs, so the
results must be bit-identical -- on two calls in a row, the second for a smaller world with fewer envs,
whose table must not inherit anything from the first."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _walls(g):
    return [dict(orient="V" if v else "H", axis_pos=float(a), endpoints=(float(e[0]), float(e[1])), width=float(w), hard=bool(h))
            for v, a, e, w, h in zip(g["wall_vertical"], g["wall_axis_pos"], g["wall_endpoints"], g["wall_width"], g["wall_hard"])]


def _big_world():
    """The walled new_core world of the reference fixture (7 agents, 6 landmarks, 4 walls), its 32 envs."""
    import mpe_hip
    g = load_golden("new_core")
    T, B, A = g["actions"].shape
    L = g["size"].shape[0] - A
    spec = mpe_hip.physics_spec(A, L, 3, g["size"], g["mass"], g["collide"], np.where(np.isnan(g["accel"]), 5.0, g["accel"]), g["max_speed"],
                                np.ones(A, bool), walls=_walls(g), ghost=g["ghost"], movable=g["movable"], new_core=True, accel=g["accel"])
    rng = np.random.RandomState(0)
    u = rng.uniform(-5, 5, (3, B, 2 * A))
    return spec, B, g["pos0"], g["vel0"], u, None


def _small_world():
    """3 agents (two speak, dim_c 4), 2 landmarks, no walls, core.py physics, 5 envs, some in contact."""
    import mpe_hip
    A, L, B = 3, 2, 5
    spec = mpe_hip.physics_spec(A, L, 4, [0.15, 0.1, 0.1, 0.2, 0.05], [1.0, 2.0, 1.0, 1.0, 1.0], [1, 1, 1, 1, 0], [5.0, 3.0, 4.0],
                                [np.inf, 1.0, 1.3], [False, True, False])
    rng = np.random.RandomState(1)
    pos = rng.uniform(-1, 1, (B, A + L, 2))
    pos[:3, 1] = pos[:3, 0] + rng.uniform(-0.15, 0.15, (3, 2))
    vel = rng.uniform(-0.5, 0.5, (B, A, 2))
    u = rng.uniform(-5, 5, (3, B, 2 * A))
    c = (rng.uniform(0, 1, (3, B, 2 * 4)) > 0.5).astype(np.float64)
    return spec, B, pos, vel, u, c


def _run_both(world, dtype):
    """Steps the world through a handle (mpe_world_step) and through the static entry point; returns both final states."""
    import mpe_hip
    from mpe_hip import _lib
    spec, B, pos0, vel0, u, c = world
    lib = _lib.load()
    env = mpe_hip.BatchedEnv(spec, B, dtype=dtype, device="cuda:0")
    env.set_state(pos0, vel0)
    pos = env.pos.clone()
    vel = env.vel.clone()
    comm = env.comm.clone() if env.comm is not None else None
    cspec, keep = spec.to_c()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    for t in range(u.shape[0]):
        ut = torch.as_tensor(u[t], dtype=dtype, device="cuda").contiguous()
        ct = torch.as_tensor(c[t], dtype=dtype, device="cuda").contiguous() if c is not None else None
        env.world_step(ut, ct)
        rc = lib.mpe_world_step_static(C.byref(cspec), B, _lib.MPE_F64 if dtype == torch.float64 else _lib.MPE_F32, 0, ptr(ut), ptr(ct),
                                       ptr(pos), ptr(vel), ptr(comm), stream)
        assert rc == 0, lib.mpe_last_error(None)
    torch.cuda.synchronize()
    return (env.pos, env.vel, env.comm), (pos, vel, comm)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_static_entry_point_equals_the_handle_path_twice_in_a_row(dtype):
    for world in (_big_world(), _small_world()):        # the second call is the smaller one
        want, got = _run_both(world, dtype)
        for w, g in zip(want, got):
            if w is None:
                assert g is None
                continue
            assert not torch.isnan(w).any()
            assert torch.equal(w, g)
    # and the large one once more after the small one
    want, got = _run_both(_big_world(), dtype)
    assert torch.equal(want[0], got[0]) and torch.equal(want[1], got[1])


def test_static_entry_point_moves_something():
    """Guard against comparing two untouched buffers: the step changes positions and copies the comm actions."""
    spec, B, pos0, vel0, u, c = _small_world()
    want, got = _run_both((spec, B, pos0, vel0, u, c), torch.float64)
    assert np.abs(got[0].cpu().numpy().reshape(B, -1, 2)[:, :3] - pos0[:, :3]).max() > 1e-3
    assert torch.equal(got[2], torch.as_tensor(c[-1], dtype=torch.float64, device="cuda"))


def test_static_entry_point_refuses_what_it_cannot_hold():
    import mpe_hip
    from mpe_hip import _lib
    lib = _lib.load()
    spec, B, pos0, vel0, u, c = _small_world()
    cspec, keep = spec.to_c()
    buf = torch.zeros(1024, dtype=torch.float64, device="cuda")
    p = C.c_void_p(buf.data_ptr())
    cspec.n_agents = _lib.MAX_AGENTS + 1
    assert lib.mpe_world_step_static(C.byref(cspec), B, _lib.MPE_F64, 0, p, p, p, p, p, None) == -1
    assert b"agents" in lib.mpe_last_error(None)
    cspec.n_agents = 3
    cspec.n_walls = _lib.MAX_WALLS + 1
    assert lib.mpe_world_step_static(C.byref(cspec), B, _lib.MPE_F64, 0, p, p, p, p, p, None) == -1
    assert b"walls" in lib.mpe_last_error(None)
    cspec.n_walls = 0
    assert lib.mpe_world_step_static(C.byref(cspec), 2 ** 31, _lib.MPE_F64, 0, p, p, p, p, p, None) == -1
    assert b"2^31-1" in lib.mpe_last_error(None)
    tag, keep2 = mpe_hip.make_spec("simple_tag").to_c()
    assert lib.mpe_world_step_static(C.byref(tag), B, _lib.MPE_F64, 0, p, p, p, p, p, None) == -1
    assert b"MPE_SCN_NONE" in lib.mpe_last_error(None)


def test_static_entry_point_runs_under_graph_capture():
    """What the entry point is for: a caller that cannot allocate.  Captured into a hipGraph and replayed, it
    gives the handle path's result (the table travels as kernel arguments, frozen at capture)."""
    import mpe_hip
    from mpe_hip import _lib
    lib = _lib.load()
    spec, B, pos0, vel0, u, c = _small_world()
    dtype = torch.float64
    env = mpe_hip.BatchedEnv(spec, B, dtype=dtype, device="cuda:0")
    env.set_state(pos0, vel0)
    pos, vel, comm = env.pos.clone(), env.vel.clone(), env.comm.clone()
    ut = torch.as_tensor(u[0], dtype=dtype, device="cuda").contiguous()
    ct = torch.as_tensor(c[0], dtype=dtype, device="cuda").contiguous()
    cspec, keep = spec.to_c()
    ptr = lambda t: C.c_void_p(t.data_ptr())
    call = lambda: lib.mpe_world_step_static(C.byref(cspec), B, _lib.MPE_F64, 0, ptr(ut), ptr(ct), ptr(pos), ptr(vel), ptr(comm),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream))
    # one call outside capture resolves the array's address on this device; undo its effect on the state
    assert call() == 0
    torch.cuda.synchronize()
    pos.copy_(env.pos); vel.copy_(env.vel); comm.copy_(env.comm)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        assert call() == 0
    for _ in range(3):
        graph.replay()
        env.world_step(ut, ct)
    torch.cuda.synchronize()
    assert torch.equal(pos, env.pos) and torch.equal(vel, env.vel) and torch.equal(comm, env.comm)


_COLD_CAPTURE = r"""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, %(pkg)r)
from mpe_hip import _lib, physics_spec
lib = _lib.load()
A, L, B = 3, 2, 5
spec = physics_spec(A, L, 4, [0.15, 0.1, 0.1, 0.2, 0.05], [1.0, 2.0, 1.0, 1.0, 1.0], [1, 1, 1, 1, 0], [5.0, 3.0, 4.0], [np.inf, 1.0, 1.3],
                    [True, True, True])
cspec, keep = spec.to_c()
rng = np.random.RandomState(1)
pos0 = torch.as_tensor(rng.uniform(-1, 1, (B, 2 * (A + L))), device="cuda")
vel0 = torch.as_tensor(rng.uniform(-0.5, 0.5, (B, 2 * A)), device="cuda")
u = torch.as_tensor(rng.uniform(-5, 5, (B, 2 * A)), device="cuda")
pos, vel = pos0.clone(), vel0.clone()
ptr = lambda t: C.c_void_p(t.data_ptr())
call = lambda: lib.mpe_world_step_static(C.byref(cspec), B, _lib.MPE_F64, 0, ptr(u), None, ptr(pos), ptr(vel), None,
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream))
# no handle was ever created and the entry point never ran: the static array's address is unknown on this device
marker = torch.zeros(1, device="cuda")
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    rc_cold = call()
    msg = lib.mpe_last_error(None).decode()
    marker.add_(1)
graph.replay()                      # a refused call enqueues nothing and leaves the capture valid: the rest of it replays
torch.cuda.synchronize()
assert float(marker) == 1.0
assert rc_cold == -1 and "capture" in msg, (rc_cold, msg)
assert torch.equal(pos, pos0) and torch.equal(vel, vel0)
assert call() == 0                  # outside capture: resolves the address and steps
torch.cuda.synchronize()
want_pos, want_vel = pos.clone(), vel.clone()
assert not torch.equal(want_pos, pos0)
pos.copy_(pos0); vel.copy_(vel0)
graph2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph2):
    assert call() == 0              # now the same call is capturable
graph2.replay()
torch.cuda.synchronize()
assert torch.equal(pos, want_pos) and torch.equal(vel, want_vel)
print("COLD-CAPTURE-OK")
"""


def test_first_call_under_capture_is_refused_cleanly():
    """A fresh process in which nothing has resolved the static array's address yet: the first call, made under
    stream capture, must be refused (MPE_ERR_INVALID, nothing enqueued, the capture still valid); after one call
    outside capture the same call captures and replays."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _COLD_CAPTURE % {"pkg": os.path.join(root, "multiagent-particle-envs_amd")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "COLD-CAPTURE-OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_after_mpe_create_the_first_call_may_come_under_capture():
    """mpe_create is the library's set-up on a device: it resolves the address, so a process whose first
    mpe_world_step_static comes under capture works when any handle was created before."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (_COLD_CAPTURE % {"pkg": os.path.join(root, "multiagent-particle-envs_amd")}).replace(
        "graph = torch.cuda.CUDAGraph()",
        "import mpe_hip\nkeep_env = mpe_hip.BatchedEnv('simple_tag', 8, device='cuda:0')\ngraph = torch.cuda.CUDAGraph()", 1).replace(
        'assert rc_cold == -1 and "capture" in msg, (rc_cold, msg)\nassert torch.equal(pos, pos0) and torch.equal(vel, vel0)',
        'assert rc_cold == 0, (rc_cold, msg)\nassert not torch.equal(pos, pos0)\npos.copy_(pos0); vel.copy_(vel0)')
    assert "keep_env" in code and "rc_cold == 0" in code
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "COLD-CAPTURE-OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
