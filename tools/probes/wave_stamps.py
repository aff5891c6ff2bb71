"""Per-wave timeline of one 25-step episode launch of simple_spread 64/64 x 2 048 worlds (library built with
-DMPE_WSTAMPS on mpe_wave.hip):  MPE_HIP_LIB=$PWD/variants/libmpe_ws.so python3 tools/probes/wave_stamps.py
Slots per step and wave: 0 step begins, 1 new positions in LDS, 2 rewards done, 3 observation stores issued."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multiagent-particle-envs_amd"))
import mpe_hip  # noqa: E402
from mpe_hip import _lib  # noqa: E402

B, T, A = 2048, 25, 64
env = mpe_hip.BatchedEnv("simple_spread", B, n_agents=A, n_landmarks=A, dtype=torch.float32, device="cuda:0")
act = torch.randint(0, 5, (T, B, A), dtype=torch.int32, device="cuda:0")
out = env.rollout(act, reset_seed=3, obs0=True)
for _ in range(5):
    env.rollout(act, out=out, reset_seed=3, obs0=True)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["MPE_HIP_LIB"])
st = np.zeros(2048 * 2 * 32 * 4, np.int64)
lib.mpe_debug_wstamps.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
assert lib.mpe_debug_wstamps(st.ctypes.data, st.size) == 0
st = st.reshape(2048, 2, 32, 4)[:, :, :T].astype(np.float64) / 100.0       # us
t0 = st[:, :, 0, 0].min()
st -= t0
print("launch: first wave starts 0, last stamp %.1f us" % st.max())
print("step | begin (mean, spread over worlds) | physics | rewards | emission issue | period     [agent wave; us]")
for t in range(T):
    w0 = st[:, 0, t]
    nxt = st[:, 0, t + 1, 0] if t + 1 < T else w0[:, 3]
    print("%4d | %7.1f +- %5.1f (min %6.1f max %6.1f) | %5.2f | %5.2f | %5.2f | %5.2f"
          % (t, w0[:, 0].mean(), w0[:, 0].std(), w0[:, 0].min(), w0[:, 0].max(), (w0[:, 1] - w0[:, 0]).mean(),
             (w0[:, 2] - w0[:, 1]).mean(), (w0[:, 3] - w0[:, 2]).mean(), (nxt - w0[:, 0]).mean()))
w1 = st[:, 1]
print("second wave: physics wait %.2f, rewards %.2f, emission issue %.2f us (means over steps)"
      % ((w1[:, :, 1] - w1[:, :, 0]).mean(), (w1[:, :, 2] - w1[:, :, 1]).mean(), (w1[:, :, 3] - w1[:, :, 2]).mean()))
xcd = np.arange(B) % 8
if os.environ.get("MPE_WAVE_SKEW"):
    ne, no = (int(v) for v in os.environ["MPE_WAVE_SKEW"].split(","))
    xcd = 2 * (np.arange(B) // (ne + no)) + (np.arange(B) % (ne + no) >= ne)
end = st[:, :, T - 1, 3].max(axis=1)
print("last stamp by XCD:", " ".join("%.0f" % end[xcd == x].max() for x in range(8)))
