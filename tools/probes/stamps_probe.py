"""Diagnostic: run the stamped build of the rollout kernel and print per-phase shader cycles."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multiagent-particle-envs_amd"))
import mpe_hip
from mpe_hip import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = mpe_hip.BatchedEnv("simple_spread", B)
act = torch.randint(0, 5, (25, B, 3), device="cuda:0", dtype=torch.int32)
out = None
for _ in range(20):
    env.reset(seed=1, observe=False)
    out = env.rollout(act, out)
torch.cuda.synchronize()
lib = _lib.load()
n = min(B // 64, 4096)
buf = (C.c_longlong * (n * 8))()
lib.mpe_debug_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.mpe_debug_stamps(buf, n * 8) == 0
a = np.array(buf[:]).reshape(n, 8).astype(np.float64)
cyc, rt = a[:, 0], a[:, 1]
print("waves", n, "cycles/wave median %.0f" % np.median(cyc), "us/wave %.1f" % np.median(rt / 100.0), "clock MHz %.0f" % np.median(cyc / (rt / 100.0)))
for q in (0, 10, 50, 90, 99, 100):
    print("  wave lifetime p%d: %.1f us" % (q, np.percentile(rt / 100.0, q)))
print("per step cycles: physics %.0f  callbacks+tile writes %.0f  flush %.0f  (sum %.0f)" % tuple(np.median(a[:, k]) / 25 for k in (2, 3, 4)) + (np.median(a[:, 2:5].sum(1)) / 25,) if False else
      "per step cycles: physics %.0f  callbacks+tile writes %.0f  flush %.0f" % tuple(np.median(a[:, k]) / 25 for k in (2, 3, 4)))
