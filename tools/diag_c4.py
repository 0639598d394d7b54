import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import pointnet_oracle as po
from tests.helpers import build_model, state_dict_cpu, synth_cloud
dev = torch.device("cuda:0")
B, N = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = build_model(N, 2, 405, 4805).train(); sd = state_dict_cpu(m)
x = synth_cloud(B, N, 4050, "box"); y = (torch.arange(B) % 2).long()
loss_ref, logp_ref, _, g64, st64 = po.train_step_torch(sd, x, y, dtype=torch.float64)
_, _, _, g32, _ = po.train_step_torch(sd, x, y, dtype=torch.float32)
mg = m.to(dev); logp, _ = mg(x.to(dev)); loss = F.nll_loss(logp, y.to(dev)); loss.backward()
rel = lambda a, b: (a.double().flatten().cpu() - b.double().flatten()).norm().item() / max(b.double().norm().item(), 1e-30)
for n, p in mg.named_parameters():
    if g64[n].double().norm().item() < 1e-9: continue
    r, r32 = rel(p.grad, g64[n]), rel(g32[n], g64[n])
    flag = "  <-- FAIL" if r > 4 * r32 + 2e-3 else ""
    print(f"{n:26s} hip {r:.2e} cpu32 {r32:.2e}{flag}")
cur = mg.state_dict()
for n in ["feat.stn.bn3.running_var", "feat.bn3.running_var", "feat.stn.bn2.running_var", "feat.stn.bn1.running_var"]:
    print(n, rel(cur[n], st64[n]))
