"""Per-wave timeline of one 25-step episode launch of simple_spread 64/64 x 2 048 worlds (library built with
-DMPE_WSTAMPS on mpe_wave.hip):  MPE_HIP_LIB=$PWD/variants/libmpe_ws.so python3 tools/probes/wave_stamps.py
Slots per step and wave: 0 step begins, 1 new positions in LDS, 2 rewards done, 3 observation stores issued (the two-role
kernel of the multi-step launches: stepping wave 0-2, streaming wave 0 and 3, per state).  With -DMPE_WAVE_NO_STREAM as
well the launch runs on the shared kernel."""
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
raw = st.reshape(2048, 2, 32, 4)
stream = not raw[:, 0, :T, 3].any()       # the two-role kernel: the stepping wave writes slots 0-2, the streaming wave 0 and 3
NS = T + 1 if stream else T               # states of the launch (the reset state's observation is one of them)
st = raw[:, :, :NS].astype(np.float64) / 100.0       # us
st -= st[:, :, 0, 0].min()
if stream:
    w0, w1 = st[:, 0], st[:, 1]
    print("two-role kernel: last stamp %.1f us" % w1[:, :, 3].max())
    print("state | stepping wave: begin (mean +- spread) | step | rewards | streaming wave: begin | stores issued after  [us]")
    for j in range(NS):
        step = (w0[:, j, 1] - w0[:, j, 0]).mean() if j < T else float("nan")
        rew = (w0[:, j, 2] - w0[:, j, 1]).mean() if j < T else float("nan")
        print("%5d | %7.1f +- %5.1f | %5.2f | %5.2f | %7.1f +- %5.1f | %5.2f"
              % (j, w0[:, j, 0].mean(), w0[:, j, 0].std(), step, rew, w1[:, j, 0].mean(), w1[:, j, 0].std(),
                 (w1[:, j, 3] - w1[:, j, 0]).mean()))
    end = w1[:, NS - 1, 3]
else:
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
    end = st[:, :, T - 1, 3].max(axis=1)
xcd = np.arange(B) % 8
if stream:      # 2 048 worlds of 64 + 64: 288 worlds to an even XCD, 224 to an odd one (launch_wave_t)
    xcd = 2 * (np.arange(B) // 512) + (np.arange(B) % 512 >= 288)
print("last stamp by XCD:", " ".join("%.0f" % end[xcd == x].max() for x in range(8)))
