#!/usr/bin/env python3
"""Where the HOST time of an eager training step goes at the reference's own recipe size (B = 64, N = 750;
main_1v.py:72-76): cProfile of 300 steps + wall / GPU time per step.  usage: python tools/host_profile.py [B N]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import bench
from pointnetgpd_amd.optim import FlatAdam

dev = torch.device("cuda:0")
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 750)
k = 2
m = bench.build_model(N, k, dev).train()
opt = FlatAdam(m.parameters(), lr=0.005)
x = bench.synth_clouds(B, N, 1, dev); y = (torch.arange(B, device=dev) % k).long()


def step():
    opt.zero_grad()
    lp, _ = m(x)
    F.nll_loss(lp, y).backward()
    opt.step()


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    step()
t1 = time.perf_counter()          # host enqueue only
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B {B} N {N}: host enqueue {1e3 * (t1 - t0) / 300:.3f} ms/step, wall {1e3 * (t2 - t0) / 300:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
