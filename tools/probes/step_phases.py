#!/usr/bin/env python3
"""Diagnostic: where one step of an episode launch spends its time, per wave (cooperative-pass scenarios).

Needs a library built with -DMPE_STAMPS on the scenario's translation unit, named by MPE_HIP_LIB.
Slots written at step 10: 30 physics done, 31 rewards done, 32 state staged, 33..40 pass k flushed,
41 post_step done; slot 2 + t = step t complete.
usage: step_phases.py n_envs scenario
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

B = int(sys.argv[1])
scn = sys.argv[2]
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
assert lib.mpe_debug_stamps(st.ctypes.data, st.size) == 0
us = st / 100.0
names = [(2 + 9, "step 9 complete"), (30, "physics"), (31, "rewards"), (32, "staged + prepare")] + \
        [(33 + k, "pass %d" % k) for k in range(8)] + [(41, "post_step"), (2 + 10, "end of step 10")]
prev = us[:, names[0][0]]
print("workgroups", G, " step 10, us per phase (median / p10 / p90 over waves)")
for slot, name in names[1:]:
    d = us[:, slot] - prev
    print("%-34s %6.2f %6.2f %6.2f" % (name, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    prev = us[:, slot]
d = us[:, 12] - us[:, 11]
print("%-34s %6.2f %6.2f %6.2f" % ("whole step", np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
life = us[:, 60] - us[:, 0]
print("wave lifetime med %.1f us; launch span %.1f us" % (np.median(life), us[:, 60].max() - us[:, 0].min()))
