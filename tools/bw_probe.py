"""Achievable HBM bandwidth on this device for plain torch fill / copy kernels (calibration only)."""
import torch, time
dev = "cuda:0"
n = 1 << 28  # 1 GiB of float32
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
dt = t(lambda: x.fill_(1.0)); print("fill  1 GiB: %.1f GB/s written" % (n * 4 / dt / 1e9))
dt = t(lambda: y.copy_(x)); print("copy  1 GiB: %.1f GB/s read+written" % (2 * n * 4 / dt / 1e9))
dt = t(lambda: x.sum()); print("sum   1 GiB: %.1f GB/s read" % (n * 4 / dt / 1e9))
