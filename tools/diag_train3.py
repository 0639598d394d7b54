"""GPU diagnostic: golden n64_k2-like case, per-parameter errors hip vs cpu-fp32 vs fp64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from oracle import pointnet_oracle as po
from tests.helpers import build_model, state_dict_cpu

def rel(a, b):
    a = a.double().flatten().cpu(); b = b.double().flatten().cpu()
    return (a - b).norm().item() / max(b.norm().item(), 1e-30)
dev = torch.device("cuda:0")
fx = np.load("tests/golden/pointnet_train_n64_k2.npz")
m = build_model(fx["num_points"], fx["k"], fx["seed_w"], fx["seed_bn"]).train()
sd = state_dict_cpu(m)
x = torch.from_numpy(fx["x"]); y = torch.from_numpy(fx["y"])
l64, lp64, tr64, g64, _ = po.train_step_torch(sd, x, y, dtype=torch.float64)
l32, lp32, tr32, g32, _ = po.train_step_torch(sd, x, y, dtype=torch.float32)
m = m.to(dev)
lp, tr = m(x.to(dev)); loss = F.nll_loss(lp, y.to(dev)); loss.backward()
print("loss", loss.item(), l32.item(), l64.item())
print("trans err hip", (tr.cpu().double()-tr64).abs().max().item(), "cpu32", (tr32.double()-tr64).abs().max().item())
print("logp err hip", (lp.cpu().double()-lp64).abs().max().item(), "cpu32", (lp32.double()-lp64).abs().max().item())
for n, p in m.named_parameters():
    if g64[n].norm().item() < 1e-9: continue
    print(f"   {n:28s} hip {rel(p.grad, g64[n]):.2e}   cpu-fp32 {rel(g32[n], g64[n]):.2e}")
