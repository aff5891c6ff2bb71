"""Diagnostic driver of xcc_id.hip: the XCD of workgroup i, launch after launch, with other launches in between."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "variants", "libxccid.so"))
lib.xcc_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")


def probe(n_wg, threads=128, stream=None):
    out = torch.full((n_wg,), -1, dtype=torch.int32, device=dev)
    st = stream or torch.cuda.current_stream()
    assert lib.xcc_probe(out.data_ptr(), n_wg, threads, C.c_void_p(st.cuda_stream)) == 0
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    shift = [(int(o[i]) - i) % 8 for i in range(min(n_wg, 64))]
    return o[:12].tolist(), sorted(set(shift))


print("fresh, 2304 workgroups:", probe(2304))
print("again:", probe(2304))
print("after a launch of 3 workgroups:", (probe(3), probe(2304))[1])
print("again:", probe(2304))
x = torch.randn(1000003, device=dev)
y = (x * 2).sum()
print("after torch kernels:", probe(2304))
print("13 workgroups:", probe(13), " then 2304:", probe(2304))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    print("on another stream:", probe(2304, stream=s))
    print("again on it:", probe(2304, stream=s))
print("back on the first stream:", probe(2304))
g = torch.cuda.CUDAGraph()
out = torch.full((2304,), -1, dtype=torch.int32, device=dev)
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        lib.xcc_probe(out.data_ptr(), 2304, 128, C.c_void_p(s.cuda_stream))
for k in range(3):
    g.replay()
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    print("graph replay", k, o[:12].tolist(), sorted(set((int(o[i]) - i) % 8 for i in range(64))))
