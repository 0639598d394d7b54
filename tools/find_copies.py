#!/usr/bin/env python3
"""Which Python call sites put ATen copies / fills / other non-libpngpd launches into an eager training step?
(torch.profiler with stacks; prints per-step counts of aten ops that launch a device kernel or memcpy.)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import bench
from pointnetgpd_amd.optim import FlatAdam
from pointnetgpd_amd import train
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
B, N, k = int(os.environ.get("B", 128)), int(os.environ.get("N", 1024)), 2
m = bench.build_model(N, k, dev).train()
opt = FlatAdam(m.parameters(), lr=0.005)
x = bench.synth_clouds(B, N, 1, dev); y = (torch.arange(B, device=dev) % k).long()
use_fused_loss = os.environ.get("FUSED_LOSS", "0") == "1"
def step():
    opt.zero_grad()
    if use_fused_loss:
        loss, _, _ = m.forward_loss(x, y); train.loss_backward(loss)
    else:
        lp, _ = m(x); F.nll_loss(lp, y).backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
STEPS = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(STEPS): step()
    torch.cuda.synchronize()
ev = prof.events()
rows = {}
for e in ev:
    if e.device_type.name != "CPU" or not e.name.startswith("aten::"):
        continue
    kids = [k for k in e.kernels] if hasattr(e, "kernels") else []
    if not kids:
        continue
    # innermost aten op only
    if any(c.name.startswith("aten::") and getattr(c, "kernels", []) for c in e.cpu_children):
        continue
    stack = [s for s in (e.stack or []) if "pointnetgpd_amd" in s or "bench" in s or "find_copies" in s][:3]
    key = (e.name, tuple(kn.name[:60] for kn in kids), tuple(stack))
    rows[key] = rows.get(key, 0) + 1
for (name, kn, stack), c in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f"{c / STEPS:5.2f}/step  {name:28s} {kn}")
    for s in stack:
        print("            ", s)
