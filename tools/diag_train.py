"""GPU diagnostic: per-parameter gradient error of the HIP train path vs the fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import pointnet_oracle as po
from tests.helpers import build_model, state_dict_cpu, synth_cloud

def rel(a, b):
    a = a.double().flatten().cpu(); b = b.double().flatten().cpu()
    return (a - b).norm().item() / max(b.norm().item(), 1e-30)

dev = torch.device("cuda:0")
for (B, N, k, kind, scale) in [(16, 750, 2, "box", 1.0), (16, 750, 2, "box", 10.0), (5, 100, 3, "box", 1.0), (2, 64, 2, "box", 1.0)]:
    m = build_model(N, k, 80 + B, 4500 + B).train()
    sd = state_dict_cpu(m)
    x = synth_cloud(B, N, 900 + B, kind) * scale
    y = (torch.arange(B) * 7 % k).long()
    loss_ref, logp_ref, trans_ref, grads_ref, stats_ref = po.train_step_torch(sd, x, y, dtype=torch.float64)
    # fp32 CPU reference error for calibration
    _, _, _, grads32, _ = po.train_step_torch(sd, x, y, dtype=torch.float32)
    m = m.to(dev)
    logp, trans = m(x.to(dev)); loss = F.nll_loss(logp, y.to(dev)); loss.backward()
    print(f"--- B={B} N={N} k={k} scale={scale}: loss err {abs(loss.item()-loss_ref.item()):.2e}")
    for n, p in m.named_parameters():
        ref = grads_ref[n]
        if ref.double().norm().item() < 1e-9: continue
        print(f"   {n:28s} hip {rel(p.grad, ref):.2e}   cpu-fp32 {rel(grads32[n], ref):.2e}   |ref| {ref.norm().item():.2e}")
