"""10 eager training steps at the reference's own batch shape (B=64, N=750) — for a rocprofv3 kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import bench
from pointnetgpd_amd.model import pointnet as pn
from pointnetgpd_amd import train as _tr
dev = torch.device("cuda:0")
B, N, k = int(os.environ.get("B", 64)), int(os.environ.get("N", 750)), 2
torch.manual_seed(0)
m = pn.PointNetCls(N, 3, k).to(dev).train()
x = bench.synth_clouds(B, N, 1, dev); y = torch.randint(0, k, (B,), device=dev)
from pointnetgpd_amd.optim import FlatAdam
opt = FlatAdam(m.parameters(), lr=0.005)
for i in range(12):
    opt.zero_grad()
    loss, _, _ = m.forward_loss(x, y); _tr.loss_backward(loss); opt.step()      # mains.py's step
torch.cuda.synchronize()
