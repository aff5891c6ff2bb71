#!/usr/bin/env python3
"""Diagnostic: per-workgroup timeline of one mpe_episode launch.

Needs a library built with -DMPE_STAMPS (see README.md next to this file) named by MPE_HIP_LIB.
Prints, over all workgroups, when (us after the earliest workgroup start) each wave had computed a
step and issued its stores, and the spread of wave lifetimes by XCD.
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
T = 25
actions = torch.randint(0, 5, (T, B, env.layout.n_act), dtype=torch.int32, device="cuda")
out = None
for _ in range(5):
    out = env.rollout(actions, out, reset_seed=7)
torch.cuda.synchronize()
lib = _lib.load()
G = min(4096, B // 64)
st = np.zeros((G, 64), dtype=np.int64)
lib.mpe_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
rc = lib.mpe_debug_stamps(st.ctypes.data, st.size)
assert rc == 0, rc
t0 = st[:, 0].min()
us = (st - t0) / 100.0


def q(col):
    v = us[:, col]
    return "min %6.2f  med %6.2f  p95 %6.2f  max %6.2f" % (v.min(), np.median(v), np.percentile(v, 95), v.max())


print("workgroups", G)
print("start        ", q(0))
print("state ready  ", q(1))
for t in (0, 1, 2, 6, 12, 13, 14, 18, 23, 24):
    print("step %2d issued %s" % (t, q(2 + t)))
dur = us[:, 60] - us[:, 0]
print("wave lifetime (start -> last step issued):", "min %.2f med %.2f max %.2f" % (dur.min(), np.median(dur), dur.max()))
print("lifetime by blockIdx % 8 (XCD under round-robin dispatch):")
for x in range(8):
    d = dur[x::8]
    print("  xcd %d: min %.1f med %.1f max %.1f" % (x, d.min(), np.median(d), d.max()))
print("lifetime by (blockIdx // 8) quartile of dispatch order:")
idx = np.arange(G) // 8
for lo, hi in ((0, 32), (32, 64), (64, 96), (96, 128)):
    d = dur[(idx >= lo) & (idx < hi)]
    print("  slots %3d-%3d: min %.1f med %.1f max %.1f" % (lo, hi, d.min(), np.median(d), d.max()))
np.save(os.path.join(ROOT, "gpurun_out", "stamps.npy"), st)
end = us[:, 60]
print("end of the last wave by blockIdx % 8 (launches with several rounds of waves end with the slowest XCD):")
for x in range(8):
    print("  xcd %d: last wave ends at %.1f us, mean lifetime %.1f us" % (x, end[x::8].max(), dur[x::8].mean()))
