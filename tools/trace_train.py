#!/usr/bin/env python3
"""N eager training steps at the bench shape (B = N = 1024) and nothing else — the workload of the per-step kernel
trace (profiles/*_train_step_trace.md: every kernel of a step, torch's own included)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"      # fp32 | bf16x3 | bf16
from pointnetgpd_amd import train as _train
_train.set_train_precision(prec)
dev = torch.device("cuda:0")
B, N, k = (int(os.environ.get(v, d)) for v, d in (("TRACE_B", 1024), ("TRACE_N", 1024), ("TRACE_K", 2)))
m = bench.build_model(N, k, dev).train()
from pointnetgpd_amd.optim import FlatAdam
opt = FlatAdam(m.parameters(), lr=0.005)
x = bench.synth_clouds(B, N, 1234, dev)
y = (torch.arange(B, device=dev) % k).long()
for _ in range(steps):
    opt.zero_grad()
    loss, _, _ = m.forward_loss(x, y)          # mains.py's step (main_1v.py:72-76): the loss inside the head's calls
    _train.loss_backward(loss)
    opt.step()
torch.cuda.synchronize()
print("steps", steps, prec, "B", B, "N", N, "k", k)
