"""Achievable HBM bandwidth on this device for plain torch fill / copy / sum kernels (calibration only).

The fill is repeated on buffers from 256 MiB to 8 GiB: the 256 MiB Infinity Cache absorbs part of a
small buffer's writes, so only the large sizes show what a long output stream can sustain."""
import torch

dev = "cuda:0"


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


for mib in (256, 1024, 4096, 8192):
    n = mib << 18                       # float32 elements
    x = torch.empty(n, dtype=torch.float32, device=dev)
    dt = t(lambda: x.fill_(1.0))
    print("fill  %5d MiB: %.0f GB/s written" % (mib, n * 4 / dt / 1e9))
    if mib in (1024, 4096):
        y = torch.empty(n, dtype=torch.float32, device=dev)
        dt = t(lambda: y.copy_(x))
        print("copy  %5d MiB: %.0f GB/s read+written" % (mib, 2 * n * 4 / dt / 1e9))
        dt = t(lambda: x.sum())
        print("sum   %5d MiB: %.0f GB/s read" % (mib, n * 4 / dt / 1e9))
        del y
    del x
