"""Shared test helpers (weights recipe of the golden fixtures, comparisons)."""
import glob
import os

import numpy as np
import torch

from oracle.pointnet_oracle import randomize_bn_

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def build_model(num_points, k, seed_w, seed_bn, cls=None):
    """The recipe recorded by oracle/make_golden.py, applied to OUR mirror module."""
    if cls is None:
        from pointnetgpd_amd.model.pointnet import PointNetCls as cls
    torch.manual_seed(int(seed_w))
    m = cls(num_points=int(num_points), input_chann=3, k=int(k))
    if int(seed_bn) >= 0:
        randomize_bn_(m.state_dict(), int(seed_bn))
    return m


def assert_checksums(model, fx):
    """The rebuilt weights must be bit-identical to the reference's (sum / abs-sum in fp64)."""
    sd = model.state_dict()
    names = [str(n) for n in fx["names"]]
    assert sorted(k for k in sd if not k.endswith("num_batches_tracked")) == names
    cs = np.array([[sd[k].double().sum().item(), sd[k].double().abs().sum().item()] for k in names])
    np.testing.assert_array_equal(cs, fx["checksums"])


def synth_cloud(b, n, seed, kind="box"):
    g = torch.Generator().manual_seed(seed)
    if kind == "box":
        w = 0.085
        u = torch.rand(b, 3, n, generator=g) - 0.5
        return (u * torch.tensor([w / 2, w, w / 2]).view(1, 3, 1)).float().contiguous()
    if kind == "gauss":
        return (torch.randn(b, 3, n, generator=g) * 0.02).float().contiguous()
    if kind == "diverse":
        # clouds that differ from each other (per-cloud anisotropic scale, rotation, offset, box / gaussian / shell
        # mix): with iid box clouds the pooled features are almost identical across the batch, the FC BatchNorms
        # divide by a vanishing batch variance and ANY arithmetic noise is amplified — not representative of crops
        # of real scenes
        w = 0.085
        base = torch.rand(b, 3, n, generator=g) - 0.5
        gau = torch.randn(b, 3, n, generator=g) * 0.3
        mix = torch.rand(b, 1, 1, generator=g)
        pts = torch.where(mix < 0.5, base, gau)
        shell = pts / pts.norm(dim=1, keepdim=True).clamp_min(1e-3) * 0.5
        pts = torch.where(mix > 0.8, shell, pts)
        scale = (0.4 + 1.2 * torch.rand(b, 3, 1, generator=g)) * torch.tensor([w / 2, w, w / 2]).view(1, 3, 1)
        q = torch.randn(b, 4, generator=g); q = q / q.norm(dim=1, keepdim=True)
        a, bq, c, d = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([1 - 2 * (c * c + d * d), 2 * (bq * c - a * d), 2 * (bq * d + a * c),
                         2 * (bq * c + a * d), 1 - 2 * (bq * bq + d * d), 2 * (c * d - a * bq),
                         2 * (bq * d - a * c), 2 * (c * d + a * bq), 1 - 2 * (bq * bq + c * c)], 1).view(b, 3, 3)
        off = (torch.rand(b, 3, 1, generator=g) - 0.5) * 0.02
        return (torch.bmm(R, pts * scale) + off).float().contiguous()
    return torch.randn(b, 3, n, generator=g).float().contiguous()


def state_dict_cpu(model):
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


def grad_tol(batch, r32):
    """Tolerance for a whole-model gradient vs the fp64 oracle.

    The gradient is a discontinuous function of the inputs: one ReLU of the (B,512) / (B,256) FC
    activations (or one arg-max of the pool) sitting within fp32 round-off of its threshold flips between
    any two fp32 implementations and moves every upstream gradient by about 1/sqrt(#active units).  So the
    bound is: 4x the deviation the reference's own ATen-fp32 run shows on this host (r32), a 2e-3 floor,
    plus the size of ONE such flip.  Tight kernel-level checks (1e-4..2e-3, no flip ambiguity) live in
    test_trunk_train_vs_prototype / test_trunk_backward_intermediates."""
    return 4 * r32 + 2e-3 + 2.5 / (batch * 512) ** 0.5


def oracle_train_step_on_device(sd, x, y, dtype, device, choices=None):
    """The oracle's train step (``oracle.pointnet_oracle.train_step_torch`` — autograd over the reference's op
    sequence) executed through ATen ON THE GPU, for cases too large for the host (B = N = 1024 in fp64 holds
    ~90 GB of activations; it fits the 288 GB of HBM).  Test-only checker.  1x1 convolutions are dispatched as
    matmuls there (MIOpen has no fp64 convolution); see ``_conv1x1_torch``."""
    from oracle import pointnet_oracle as po
    old = po.CONV_AS_MATMUL
    po.CONV_AS_MATMUL = True
    try:
        sdd = {k: v.to(device) for k, v in sd.items()}
        if choices is not None:
            choices = {k: ([m.to(device) for m in v] if isinstance(v, list) else v.to(device))
                       for k, v in choices.items()}
        loss, logp, trans, grads, stats = po.train_step_torch(sdd, x.to(device), y.to(device), dtype=dtype,
                                                              choices=choices)
    finally:
        po.CONV_AS_MATMUL = old
    torch.cuda.synchronize()
    out = (loss.cpu(), logp.cpu(), trans.cpu(), {k: v.cpu() for k, v in grads.items()},
           {k: v.cpu() for k, v in stats.items()})
    del sdd, loss, logp, trans, grads, stats, choices
    torch.cuda.empty_cache()
    return out


def capture_choices(run):
    """Run ``run()`` (a HIP train-mode forward [+ backward]) with the pass-by-pass sequencing and record the DISCRETE
    decisions it took: the arg-max point of every pooled value of both trunks, which pooled values the STN's
    ReLU-before-max kept, and the activation pattern of the four FC ReLUs — the ``choices`` argument of
    ``oracle.pointnet_oracle.forward_torch``.  (Pass sequencing is bit-identical to the fused entries,
    tests/test_gpu_fused.py; it is used here because its autograd nodes expose the arg-max indices.)"""
    from pointnetgpd_amd import train
    ch = {"fc_keep": []}
    o_trunk, o_fc = train.trunk_train, train.fc_bn_relu_train

    def trunk(mod, x, trans, relu_last):
        pooled = o_trunk(mod, x, trans, relu_last)
        idx = pooled.grad_fn.saved_tensors[16]            # TrunkTrainFn.forward's save order
        assert idx.dtype == torch.int32 and tuple(idx.shape) == tuple(pooled.shape)
        if relu_last:
            ch["stn_idx"], ch["stn_keep"] = idx.long().clone(), (pooled.detach() > 0)
        else:
            ch["feat_idx"] = idx.long().clone()
        return pooled

    def fc(lin, bn, inp):
        y = o_fc(lin, bn, inp)
        ch["fc_keep"].append(y.detach() > 0)
        return y

    train.set_sequencing("passes")
    train.trunk_train, train.fc_bn_relu_train = trunk, fc
    try:
        out = run()
    finally:
        train.trunk_train, train.fc_bn_relu_train = o_trunk, o_fc
        train.set_sequencing("fused")
    assert len(ch["fc_keep"]) == 4 and "stn_idx" in ch and "feat_idx" in ch
    return out, ch


def oracle_forward_on_device(sd, x, device, dtype=torch.float64, chunk=128):
    """The oracle's eval-mode forward (``oracle.pointnet_oracle.forward_torch`` — the reference's ATen op sequence)
    over the WHOLE batch, executed through ATen on the GPU in ``dtype`` (fp64: 1x1 convolutions as matmuls, MIOpen has
    no fp64 convolution).  Eval-mode samples are independent, so the batch is walked in chunks to bound the
    (chunk,1024,N) activations.  Test-only checker."""
    from oracle import pointnet_oracle as po
    old = po.CONV_AS_MATMUL
    po.CONV_AS_MATMUL = True
    try:
        sdd = {k: (v.to(device).to(dtype) if v.is_floating_point() else v.to(device)) for k, v in sd.items()}
        lps, trs = [], []
        with torch.no_grad():
            for i in range(0, x.shape[0], chunk):
                lp, tr = po.forward_torch(sdd, x[i:i + chunk].to(device).to(dtype))
                lps.append(lp.cpu()); trs.append(tr.cpu())
    finally:
        po.CONV_AS_MATMUL = old
    torch.cuda.synchronize()
    del sdd
    torch.cuda.empty_cache()
    return torch.cat(lps), torch.cat(trs)


# ---------------------------------------------------------------------------------------------------------------
# FC-stack cases whose ReLUs cannot flip (tests/test_gpu_head_train.py)
# ---------------------------------------------------------------------------------------------------------------
def head_case(B, K0, H1, H2, k, seed, margin=2e-3):
    """Parameters + input of one FC stack (fc1/bn/relu -> fc2/bn/relu -> fc3, pointnet.py:35-37 / :191-193) in fp64
    such that every pre-ReLU activation of the train-mode forward is at least ``margin`` away from zero — no ReLU can
    flip between an fp32 and an fp64 evaluation, so the gradient comparison needs no allowance for flips.

    Random parameters and a random input; then the (few) activations that land too close to a threshold are pushed
    away from it by the smallest change of the INPUT that does it: K0 >= H1, so any change of the first layer's
    pre-activations ``z1`` is realised exactly by ``d inp = d z1 (W1 W1^T)^-1 W1``; a second-layer entry (b, j) is
    moved through row b of the first layer's active units along ``W2[j]``.  At B = 2 train-mode BatchNorm maps every
    column to -/+ 1 whatever the input, so there the offending channel's beta is nudged instead."""
    from oracle.pointnet_oracle import head_stack_torch
    assert K0 >= H1
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
    n = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    lin = lambda o, i: ((r(o, i) * 2 - 1) / i ** 0.5, (r(o) * 2 - 1) / i ** 0.5)
    P = {}
    P["W1"], P["b1"] = lin(H1, K0)
    P["W2"], P["b2"] = lin(H2, H1)
    P["W3"], P["b3"] = lin(k, H2)
    for i, H in ((1, H1), (2, H2)):
        gam = r(H) + 0.5
        P[f"g{i}"] = torch.where(r(H) < 0.15, -gam, gam)          # some negative gammas: the sign matters in the backward
        P[f"be{i}"] = n(H) * 0.3
        P[f"rm{i}"], P[f"rv{i}"] = n(H) * 0.1, r(H) + 0.5
    inp = n(B, K0) * 0.5 + n(1, K0) * 0.2 + 0.1 * r(B, 1)
    pinv = torch.linalg.solve(P["W1"] @ P["W1"].t(), P["W1"])        # (H1,K0): d inp = d z1 @ pinv
    eps = 1e-5
    if B == 2:
        # two samples: BatchNorm's variance is (z_a - z_b)^2 / 4, so a column whose two values nearly coincide is
        # ill-conditioned in ANY fp32 implementation (the reference's included).  Keep the two rows apart in every
        # column of both layers: layer 1 through the input, layer 2 through W2 (y1 is fixed by gamma1 / beta1 there).
        def spread(d, least):
            tgt = torch.where(d >= 0, least, -least).to(d.dtype)
            return torch.where(d.abs() < least, tgt - d, torch.zeros_like(d))
        z1 = inp @ P["W1"].t() + P["b1"]
        fix = spread(z1[0] - z1[1], 0.25) / 2
        inp = inp + torch.stack([fix, -fix]) @ pinv
        z1 = inp @ P["W1"].t() + P["b1"]
        a1 = P["g1"] * (z1 - z1.mean(0)) / (z1.var(0, unbiased=False) + eps).sqrt() + P["be1"]
        P["be1"] = torch.where((a1.abs() < 0.05).any(0), P["be1"] + 0.2, P["be1"])
        a1 = P["g1"] * (z1 - z1.mean(0)) / (z1.var(0, unbiased=False) + eps).sqrt() + P["be1"]
        y1 = a1.clamp_min(0)
        v = y1[0] - y1[1]
        d2 = (y1 @ P["W2"].t())[0] - (y1 @ P["W2"].t())[1]
        P["W2"] = P["W2"] + spread(d2, 0.1)[:, None] * v[None, :] / (v * v).sum()
    for it in range(400):
        z1 = inp @ P["W1"].t() + P["b1"]
        s1 = (z1.var(0, unbiased=False) + eps).sqrt()
        a1 = P["g1"] * (z1 - z1.mean(0)) / s1 + P["be1"]
        z2 = a1.clamp_min(0) @ P["W2"].t() + P["b2"]
        s2 = (z2.var(0, unbiased=False) + eps).sqrt()
        a2 = P["g2"] * (z2 - z2.mean(0)) / s2 + P["be2"]
        bad1, bad2 = a1.abs() < margin, a2.abs() < margin
        if not (bad1.any() or bad2.any()):
            break
        if B == 2:
            P["be1"] = torch.where(bad1.any(0), P["be1"] + 8 * margin, P["be1"])
            P["be2"] = torch.where(bad2.any(0), P["be2"] + 8 * margin, P["be2"])
            continue
        sgn = lambda a: torch.where(a >= 0, 1.0, -1.0).to(a.dtype)
        dz1 = torch.zeros_like(z1)
        if bad1.any():
            dz1 = torch.where(bad1, sgn(a1) * sgn(P["g1"]) * 4 * margin * s1 / P["g1"].abs(), dz1)
        else:
            for b, j in bad2.nonzero().tolist():
                act = (a1[b] > 8 * margin).to(z1.dtype)                   # active units with slack
                d = P["W2"][j] * act
                dz2 = sgn(a2[b, j]) * sgn(P["g2"][j]) * 4 * margin * s2[j] / P["g2"][j].abs()
                da1 = dz2 * d / (d * d).sum()
                dz1[b] += da1 * s1 / P["g1"]
        inp = inp + dz1 @ pinv
    else:
        raise AssertionError("could not build a flip-free FC-stack case")
    Pw = {k_: v.clone() for k_, v in P.items()}
    _, acts = head_stack_torch(inp, Pw, "none", training=True)
    assert min(a.abs().min().item() for a in acts) >= margin * 0.99
    return inp, P


def rel_max(a, b):
    """max |a - b| / max |b| (fp64)."""
    a, b = a.double().cpu(), b.double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-300)
