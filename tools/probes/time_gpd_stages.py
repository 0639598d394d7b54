#!/usr/bin/env python3
"""Per-launch times of the GPD comparator's training kernels (HIP events on the current stream)."""
import os
import sys
import json
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pointnetgpd_amd import gpd_ops, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
C = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.rand(B, C, 60, 60, device=dev) * (torch.rand(B, 1, 60, 60, device=dev) < 0.3)
w1 = torch.randn(20, C, 5, 5, device=dev) * 0.1; b1 = torch.randn(20, device=dev)
w2 = torch.randn(50, 20, 5, 5, device=dev) * 0.05; b2 = torch.randn(50, device=dev)
fw1 = torch.randn(500, 7200, device=dev) * 0.01; fb1 = torch.randn(500, device=dev)
p1, a1 = gpd_ops.conv5_pool2_arg(x, w1, b1)
p2, a2 = gpd_ops.conv5_pool2_arg(p1, w2, b2)
g2 = torch.randn_like(p2)
g1 = torch.randn_like(p1)
flat = p2.view(B, -1)
gh = torch.randn(B, 500, device=dev)


def t(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1000, 1)


out = {"B": B, "C": C,
       "fwd1_us": t(lambda: gpd_ops.conv5_pool2_arg(x, w1, b1)),
       "fwd2_us": t(lambda: gpd_ops.conv5_pool2_arg(p1, w2, b2)),
       "bwd2_w_and_x_us": t(lambda: gpd_ops.conv5_pool2_bwd(p1, w2, g2, a2, True)),
       "bwd2_w_us": t(lambda: gpd_ops.conv5_pool2_bwd(p1, w2, g2, a2, False)),
       "bwd1_w_us": t(lambda: gpd_ops.conv5_pool2_bwd(x, w1, g1, a1, False)),
       "fc1_fwd_us": t(lambda: ops.fc_fwd(flat, fw1, fb1, ops.EPI_RELU)),
       "fc1_fwd_splitk_us": t(lambda: gpd_ops.fc_fwd_splitk(flat, fw1, fb1, True)),
       "fc1_bwd_us": t(lambda: ops.fc_bwd(gh, flat, fw1))}
print(json.dumps(out))
