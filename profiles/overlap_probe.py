#!/usr/bin/env python3
"""profiles/overlap_probe.py -- C2 steps on ONE stream vs alternating between two engines on two
streams (the tail kernels of step i under the head of step i+1).  Experiment, not the bench."""
import sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from expann_amd import GpuBruteForceEngine

n, d, m, k = 1_000_000, 128, 10_000, 10
g = torch.Generator(device="cuda"); g.manual_seed(1234)
base = torch.randn(n, d, device="cuda", generator=g)
g.manual_seed(4321)
q = [torch.randn(m, d, device="cuda", generator=g) for _ in range(2)]
engs, streams, outs = [], [], []
for i in range(2):
    e = GpuBruteForceEngine(d, "l2")
    e.set_base_device(base.data_ptr(), n, 0)
    e.set_option("async_search", 1)
    engs.append(e)
    streams.append(torch.cuda.Stream())
    outs.append((torch.empty(m, k, dtype=torch.int64, device="cuda"), torch.empty(m, k, device="cuda")))
torch.cuda.synchronize()

def run(n_eng, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        i = s % n_eng
        engs[i].search_device(q[i].data_ptr(), m, k, outs[i][0].data_ptr(), outs[i][1].data_ptr(),
                              streams[i].cuda_stream)
    for i in range(n_eng):
        engs[i].sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for n_eng in (1, 2, 1, 2):
    run(n_eng, 4)
    print(f"engines/streams {n_eng}: {run(n_eng, 40):.3f} ms per step", flush=True)
a = outs[0][0].clone()
run(1, 2)
print("same ids:", bool((a == outs[0][0]).all()))
