#!/usr/bin/env python3
"""Diagnostic: per-workgroup timeline of ONE mpe_step launch (the launch a policy-in-the-loop trainer makes per step).

Needs a library built with -DMPE_STAMPS (see README.md next to this file) named by MPE_HIP_LIB.  Prints, over all
workgroups, when (us after the earliest workgroup start) each wave started, had its state and actions, had issued its
output rows, had issued its state write-back and had every store acknowledged -- i.e. how a single step's launch
divides into dispatch ramp, load latency, arithmetic + store issue, and store drain -- next to the launch's duration
by HIP events (inside a hipGraph of 25 steps).
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multiagent-particle-envs_amd"))
import mpe_hip  # noqa: E402
from mpe_hip import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
scn = sys.argv[2] if len(sys.argv) > 2 else "simple_spread"
env = mpe_hip.BatchedEnv(scn, B, dtype=torch.float32, info=False)
actions = torch.randint(0, 5, (25, B, env.layout.n_act), dtype=torch.int32, device="cuda")
env.reset(seed=3)
for t in range(25):
    env.step(actions[t])
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for t in range(25):
        env.step(actions[t])
for _ in range(20):
    g.replay()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40):
    g.replay()
e1.record()
torch.cuda.synchronize()
print("launch duration inside a 25-step hipGraph: %.2f us per step" % (e0.elapsed_time(e1) * 1e3 / (40 * 25)))
lib = _lib.load()
G = min(4096, B // 64)
st = np.zeros((G, 64), dtype=np.int64)
lib.mpe_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
rc = lib.mpe_debug_stamps(st.ctypes.data, st.size)          # the stamps of the last launch
assert rc == 0, rc
t0 = st[:, 0].min()
us = (st - t0) / 100.0


def q(col):
    v = us[:, col]
    return "min %5.2f  p5 %5.2f  med %5.2f  p95 %5.2f  max %5.2f" % (v.min(), np.percentile(v, 5), np.median(v), np.percentile(v, 95), v.max())


print("workgroups", G, "(100 MHz timestamps: 0.01 us resolution)")
print("wave started             ", q(0))
print("state + actions ready    ", q(1))
print("output rows issued       ", q(60))
print("state write-back issued  ", q(61))
print("all stores acknowledged  ", q(62))
for name, a, b in (("load wait", 0, 1), ("arithmetic + row issue", 1, 60), ("write-back issue", 60, 61), ("store drain", 61, 62), ("lifetime", 0, 62)):
    d = us[:, b] - us[:, a]
    print("%-24s med %5.2f  p95 %5.2f  max %5.2f us" % (name, np.median(d), np.percentile(d, 95), d.max()))
print("by blockIdx % 8 (XCD): start med / end med")
for x in range(8):
    print("  xcd %d: %5.2f / %5.2f" % (x, np.median(us[x::8, 0]), np.median(us[x::8, 62])))
