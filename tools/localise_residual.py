#!/usr/bin/env python3
"""Where does the residual of the 'residue' gradients come from?  (VERDICT r4 P1 / next-round #9.)

tests/test_gpu_grad_gate.py shows, with the discrete decisions imposed, the STN3d tensors and conv1 / conv2 / bn1 / bn2 of
the feature trunk 3-6e-3 from the fp64 oracle on the headline's iid box clouds (B = N = 1024) while every other tensor
meets 1e-3.  This tool splits that error by HAND-OFF: the training graph is cut at dL/dpooled of the feature trunk (the
FC head's output, ``dp``), at dL/dtrans (``dT``, the feature trunk's output into the STN) and at dL/dpooled of the STN
trunk; at each cut the HIP kernels downstream are run twice — with the HIP path's own upstream gradient, and with the
fp64 oracle's upstream gradient rounded to fp32 — and every parameter gradient is compared with the oracle.  If a group's
error collapses when it is fed the exact upstream, the error was INHERITED through that hand-off; what remains is what
the group's own kernels add.  The error of each hand-off itself and its conditioning floor (the same hand-off of the
fp64 oracle with clouds and weights perturbed by half an fp32 ulp) are printed next to it.

    python tools/localise_residual.py [B N kind]          (default 1024 1024 box; needs ~100 GB of HBM)
"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from oracle import pointnet_oracle as po                       # the checker (diagnostic tool, not a product path)
from pointnetgpd_amd import train
from tests.helpers import build_model, capture_choices, state_dict_cpu, synth_cloud

dev = torch.device("cuda:0")
B, N, kind = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]) if len(sys.argv) > 3 else (1024, 1024, "box")
k, seed = 2, 1


def rel(a, b):
    a = a.double().flatten().cpu(); b = b.double().flatten().cpu()
    return (a - b).norm().item() / max(b.norm().item(), 1e-300)


def oracle_with_handoffs(sd, x, y, choices):
    """forward_torch's body (oracle/pointnet_oracle.py:198-229) in fp64 on the device, keeping the three hand-offs."""
    po.CONV_AS_MATMUL = True
    W = {n: (v.double().to(dev).requires_grad_(True) if v.is_floating_point() and "running" not in n else v.clone().to(dev))
         for n, v in sd.items()}
    for n, v in W.items():
        if "running" in n:
            W[n] = v.double()
    c = {kk: ([m.to(dev) for m in v] if isinstance(v, list) else v.to(dev)) for kk, v in choices.items()}
    fk = c["fc_keep"]
    xx = x.double().to(dev)
    ps = po._trunk_torch(F, xx, W, "feat.stn.", True, True, (c["stn_idx"], c["stn_keep"]))
    ps.retain_grad()
    g = po._relu_choice(F, po._bn_torch(F, F.linear(ps, W["feat.stn.fc1.weight"], W["feat.stn.fc1.bias"]), W, "feat.stn.bn4", True), fk[0])
    g = po._relu_choice(F, po._bn_torch(F, F.linear(g, W["feat.stn.fc2.weight"], W["feat.stn.fc2.bias"]), W, "feat.stn.bn5", True), fk[1])
    g = F.linear(g, W["feat.stn.fc3.weight"], W["feat.stn.fc3.bias"])
    trans = (g + torch.eye(3, dtype=torch.float64, device=dev).view(1, 9)).view(-1, 3, 3)
    trans.retain_grad()
    xt = torch.bmm(xx.transpose(2, 1), trans).transpose(2, 1)
    pf = po._trunk_torch(F, xt, W, "feat.", True, False, (c["feat_idx"], None))
    pf.retain_grad()
    f = po._relu_choice(F, po._bn_torch(F, F.linear(pf, W["fc1.weight"], W["fc1.bias"]), W, "bn1", True), fk[2])
    f = po._relu_choice(F, po._bn_torch(F, F.linear(f, W["fc2.weight"], W["fc2.bias"]), W, "bn2", True), fk[3])
    logp = F.log_softmax(F.linear(f, W["fc3.weight"], W["fc3.bias"]), dim=-1)
    F.nll_loss(logp, y.to(dev)).backward()
    grads = {n: v.grad.detach().cpu() for n, v in W.items() if torch.is_tensor(v) and v.requires_grad}
    out = dict(dp=pf.grad.detach().cpu(), dT=trans.grad.detach().cpu(), dps=ps.grad.detach().cpu(), grads=grads,
               trans=trans.detach().cpu())
    del W, ps, g, trans, xt, pf, f, logp
    torch.cuda.empty_cache()
    return out


m = build_model(N, k, 310 + seed, 5200 + seed).train()
sd = state_dict_cpu(m)
x = synth_cloud(B, N, 1800 + seed, kind)
y = (torch.arange(B) * 7 % k).long()
mg = m.to(dev)
xd, yd = x.to(dev), y.to(dev)
hand = {}


def run_full():
    """The HIP step, recording its own hand-offs."""
    o_trunk = train.trunk_train

    def trunk(mod, xx, trans, relu_last):
        pooled = o_trunk(mod, xx, trans, relu_last)
        pooled.register_hook(lambda gr, key=("dps" if relu_last else "dp"): hand.__setitem__(key, gr.detach().clone()))
        return pooled
    train.trunk_train = trunk
    try:
        logp, trans = mg(xd)
        trans.register_hook(lambda gr: hand.__setitem__("dT", gr.detach().clone()))
        F.nll_loss(logp, yd).backward()
    finally:
        train.trunk_train = o_trunk


_, ch = capture_choices(run_full)
torch.cuda.synchronize()
own = {n: p.grad.detach().cpu().clone() for n, p in mg.named_parameters()}
ref = oracle_with_handoffs(sd, x, y, ch)
# conditioning floor of each hand-off: the fp64 oracle with clouds and weights moved by half an fp32 ulp
gen = torch.Generator().manual_seed(99)
hu = 2.0 ** -24
pert = lambda t: t.double() * (1 + hu * (torch.randint(0, 2, t.shape, generator=gen).double() * 2 - 1))
sdp = {n: (pert(v) if v.is_floating_point() and "running" not in n else v) for n, v in sd.items()}
refp = oracle_with_handoffs(sdp, pert(x), y, ch)


def grads_with(cut, upstream):
    """Parameter gradients of everything DOWNSTREAM of ``cut`` when the HIP kernels are fed ``upstream`` there."""
    for p in mg.parameters():
        p.grad = None
    if cut == "dp":                         # feature trunk (and, through dT, the whole STN)
        pooled, trans = mg.feat(xd)
        pooled.backward(upstream.to(dev).float())
    elif cut == "dT":                       # the STN alone
        trans = mg.feat.stn(xd)
        trans.backward(upstream.to(dev).float())
    else:                                   # "dps": the STN's trunk alone
        pooled = train.trunk_train(mg.feat.stn, xd, None, relu_last=True)
        pooled.backward(upstream.to(dev).float())
    return {n: p.grad.detach().cpu().clone() for n, p in mg.named_parameters() if p.grad is not None}


groups = [("feat trunk conv3 / bn3.weight", ["feat.conv3.weight", "feat.bn3.weight"]),
          ("feat trunk conv2 / bn2 (residue)", ["feat.conv2.weight", "feat.bn2.weight", "feat.bn2.bias"]),
          ("feat trunk conv1 / bn1 (residue)", ["feat.conv1.weight", "feat.bn1.weight", "feat.bn1.bias"]),
          ("STN FC stack (fc1-3, bn4-5)", ["feat.stn.fc1.weight", "feat.stn.fc2.weight", "feat.stn.fc3.weight",
                                            "feat.stn.bn4.weight", "feat.stn.bn5.weight"]),
          ("STN trunk conv3 / bn3.weight", ["feat.stn.conv3.weight", "feat.stn.bn3.weight"]),
          ("STN trunk conv1-2 / bn1-2", ["feat.stn.conv1.weight", "feat.stn.conv2.weight", "feat.stn.bn1.weight",
                                          "feat.stn.bn2.weight"])]
print(f"## localise_residual  B={B} N={N} {kind} clouds, decisions imposed; relative L2 errors against the fp64 oracle")
print("\nhand-off                       | HIP path's own | conditioning floor (fp64, inputs +- half an fp32 ulp)")
for key, name in (("dp", "dL/dpooled (FC head -> feature trunk)"), ("dT", "dL/dtrans  (feature trunk -> STN)"),
                  ("dps", "dL/dpooled (STN FC stack -> STN trunk)")):
    print(f"{name:40s} | {rel(hand[key], ref[key]):.2e} | {rel(refp[key], ref[key]):.2e}")
runs = {"own upstream (the step as it runs)": own,
        "exact dp fed to the feature trunk": grads_with("dp", ref["dp"]),
        "exact dT fed to the STN": grads_with("dT", ref["dT"]),
        "exact dL/dpooled fed to the STN trunk": grads_with("dps", ref["dps"])}
print("\ngroup (worst tensor)                 | " + " | ".join(runs))
for gname, names in groups:
    row = []
    for rn, gr in runs.items():
        vals = [rel(gr[n], ref["grads"][n]) for n in names if n in gr]
        row.append(f"{max(vals):.2e}" if vals else "   -    ")
    print(f"{gname:36s} | " + " | ".join(f"{v:>34s}" for v in row))
print("\nfloor of the same groups (fp64 oracle, inputs +- half an fp32 ulp): " +
      ", ".join(f"{gname}: {max(rel(refp['grads'][n], ref['grads'][n]) for n in names):.1e}" for gname, names in groups))

# ---- inside the feature trunk's backward: every accumulated quantity against the fp64 pass-structured prototype
#      (tests/train_algo_prototype.py, itself checked against autograd) at the HIP run's arg-max points and upstream dp
from tests.test_gpu_train import _trunk_params
from tests.train_algo_prototype import trunk_bwd, trunk_fwd

caps = []
orig_bwd = train.TrunkTrainFn.backward


def wrapped(ctx, dp):
    train.DEBUG_STASH = {}
    out = orig_bwd(ctx, dp)
    caps.append(dict(train.DEBUG_STASH))
    train.DEBUG_STASH = None
    return out


for which in ("feat", "stn"):
    caps.clear()
    for p in mg.parameters():
        p.grad = None
    train.TrunkTrainFn.backward = staticmethod(wrapped)
    train.set_sequencing("passes")
    try:
        logp, trans = mg(xd)
        F.nll_loss(logp, yd).backward()
    finally:
        train.TrunkTrainFn.backward = orig_bwd
        train.set_sequencing("fused")
    cap = caps[0] if which == "feat" else caps[1]          # backward order: the feat trunk first, then the STN trunk
    mod = mg.feat if which == "feat" else mg.feat.stn
    Pd = {n: v.to(dev) for n, v in _trunk_params(mod).items()}
    T = trans.detach().double() if which == "feat" else None
    _, sv = trunk_fwd(xd.double(), T, Pd, relu_last=which == "stn")
    sv["idx"] = cap["idx"].long()
    g = trunk_bwd(cap["dp"], Pd, sv)
    dbg = g["_dbg"]
    print(f"\n{which} trunk, pass by pass (HIP fp32 vs fp64 prototype, same arg-max points and dp), in the order they are produced:")
    for kx, what in [("dg3", "d gamma3"), ("S2", "sum h2 h2^T (Gram, pass D)"), ("G", "gather: sum s3 dp h2[n*]"),
                     ("A", "A = W3^T diag(.) W3"), ("cvec", "cvec"), ("g2buf", "g2 = relu'(.) dh2 (pass D output, per point)"),
                     ("a1", "a1 = sum g2            (pass D)"), ("a2", "a2 = sum g2 zhat2      (pass D)"),
                     ("c1", "c1 = sum g1            (pass E)"), ("c2", "c2 = sum g1 zhat1      (pass E)"),
                     ("Rb", "R_b = sum_n g1 x^T     (pass E)")]:
        if kx in cap and kx in dbg:
            print(f"  {what:46s} {rel(cap[kx], dbg[kx]):.2e}")
    for kx, ky in [("dW3", "W3"), ("dW2", "W2"), ("dW1", "W1")] + ([("dT", "T")] if which == "feat" else []):
        print(f"  {kx:46s} {rel(cap[kx], g[ky]):.2e}")
    # the prototype is the fp64 backward of THIS trunk fed the HIP forward's own fp32 hand-offs (trans, dp); the oracle
    # differentiates the whole model in fp64.  Their distance is what the forward's fp32 rounding of those hand-offs
    # costs once the (ill-conditioned) gradient amplifies it — no backward kernel is involved in this number.
    pre = "feat." if which == "feat" else "feat.stn."
    for ky, name in (("W3", "conv3.weight"), ("W2", "conv2.weight"), ("W1", "conv1.weight")):
        print(f"  fp64 prototype on the HIP forward's hand-offs vs the whole-model fp64 oracle, d{ky}: "
              f"{rel(g[ky].reshape(-1), ref['grads'][pre + name].reshape(-1)):.2e}")
    if which == "feat":
        print(f"  (the hand-off itself: trans of the HIP forward vs the oracle's, relative {rel(trans.detach(), ref['trans']):.2e})")
    del sv, g, dbg
    torch.cuda.empty_cache()
