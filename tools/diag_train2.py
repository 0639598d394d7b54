"""GPU diagnostic: feat-trunk backward intermediates (HIP) vs the fp64 prototype in a model-level scenario."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from tests.helpers import build_model, synth_cloud
from tests.train_algo_prototype import trunk_fwd, trunk_bwd
from pointnetgpd_amd import train

def rel(a, b):
    a = a.double().flatten().cpu(); b = b.double().flatten().cpu()
    return (a - b).norm().item() / max(b.norm().item(), 1e-30)

dev = torch.device("cuda:0")
B, N, k, scale = 16, 750, 2, float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
m = build_model(N, k, 80 + B, 4500 + B).train()
x = synth_cloud(B, N, 900 + B, "box") * scale
y = (torch.arange(B) * 7 % k).long()
mod_cpu = m.feat
P = {}
for i in (1, 2, 3):
    conv, bn = getattr(mod_cpu, f"conv{i}"), getattr(mod_cpu, f"bn{i}")
    P[f"W{i}"] = conv.weight.detach()[:, :, 0].double(); P[f"b{i}"] = conv.bias.detach().double()
    P[f"g{i}"] = bn.weight.detach().double(); P[f"be{i}"] = bn.bias.detach().double()
m = m.to(dev)
train.DEBUG_STASH = {}
stash_all = []
orig = train.TrunkTrainFn.backward
logp, trans = m(x.to(dev)); loss = F.nll_loss(logp, y.to(dev))
# run backward; the LAST trunk backward executed is the STN one, the FIRST is the feat one: capture the first
caps = []
def wrapped(ctx, dp):
    train.DEBUG_STASH = {}
    out = orig(ctx, dp)
    caps.append(dict(train.DEBUG_STASH))
    return out
train.TrunkTrainFn.backward = staticmethod(wrapped)
loss.backward()
feat = caps[0]
T = trans.detach().double().cpu()
dp = feat["dp"].cpu()
pooled_ref, sv = trunk_fwd(x.double(), T, P, relu_last=False)
g = trunk_bwd(dp, P, sv)
dbg = g["_dbg"]
print("idx mismatch frac", (feat["idx"].cpu().long() != dbg["idx"]).double().mean().item())
for kx in ["dg3", "dbe3", "S2", "S1", "sh", "sh1", "G", "A", "cvec", "a1", "a2", "Pm", "c1", "c2", "Rb", "g2buf"]:
    print(f"  {kx:6s} rel err {rel(feat[kx], dbg[kx]):.3e}")
for kx, ky in [("dW1", "W1"), ("dW2", "W2"), ("dW3", "W3"), ("dT", "T")]:
    print(f"  {kx:6s} rel err {rel(feat[kx], g[ky]):.3e}")
