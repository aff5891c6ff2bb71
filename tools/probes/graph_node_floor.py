#!/usr/bin/env python3
"""Diagnostic: what ONE node of a hipGraph of 25 dependent kernels costs on this device when the kernel does next to
nothing, and when it only moves a single step's bytes (a fill of 15 MB, a copy reading 5.5 MB and writing 15 MB) --
the floor under "one mpe_step launch per step" at 65 536 worlds of simple_spread (20.5 MB per step)."""
import torch

dev = "cuda"
tiny = torch.zeros(64, device=dev)
out15 = torch.empty(15 * 1024 * 1024 // 4, device=dev)
src = torch.randn(out15.numel(), device=dev)
small = torch.randn(int(5.5 * 1024 * 1024) // 4, device=dev)
acc = torch.zeros_like(small)


def per_node(fn, nodes=25, replays=200):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(nodes):
            fn()
    for _ in range(20):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (replays * nodes)


print("empty node (64-element add)              : %.2f us per node" % per_node(lambda: tiny.add_(1.0)))
print("fill 15 MB                                : %.2f us per node" % per_node(lambda: out15.fill_(1.0)))
print("copy 15 MB (reads 15, writes 15)          : %.2f us per node" % per_node(lambda: out15.copy_(src)))


def step_like():
    acc.add_(small)          # reads 5.5 + 5.5 MB, writes 5.5 MB: stands for state + actions in, state out
    out15.fill_(2.0)         # the observation rows out


print("two nodes: add 5.5 MB in place + fill 15 MB: %.2f us per pair" % (2 * per_node(step_like, nodes=24)))
