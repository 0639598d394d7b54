"""Whole-model gradient gate, flip-free: all 44 parameter gradients of a training step against the
fp64 oracle at the benched size (configs[1]: B = N = 1024) and at the full-view geometry (configs[3]: N = 4096), with NO
fp32 yardstick in any bound.

Two things limit a whole-model gradient comparison, and this file separates them:

1. DISCRETE DECISIONS.  An arg-max of the pool or a ReLU of the FC stacks sitting within fp32 round-off of its threshold
   flips between any two correct fp32 implementations and moves whole gradient tensors by percent
   (tests/test_gpu_train_large.py keeps that flip-AWARE comparison).  Here the fp64 oracle (autograd over the reference's
   op sequence, pointnet.py:27-45,137-154,189-194 under main_1v.py:72-75, run through ATen on the device) is evaluated
   with the HIP run's own decisions imposed — max over N -> gather at the HIP run's arg-max points, the STN's
   ReLU-before-max and the four FC ReLUs -> the HIP run's activation patterns
   (``oracle.pointnet_oracle.forward_torch(choices=...)``).  Both sides then evaluate the same smooth function.
2. CANCELLATION.  What is left is rounding — and the gradients of everything UPSTREAM of a batch-statistics BatchNorm are
   residues: ``dz = (g/s)(dy - mean(dy) - zhat mean(dy zhat))`` subtracts the batch mean of a ``dy`` that is nearly
   constant over the batch when the clouds resemble each other.  Measured with the decisions imposed (diagnostic mode,
   PNGPD_GATE_DIAG=1; profiles/r04_gate_diag.txt): on the headline's iid box clouds at B = N = 1024 the STN / trunk
   gradients of these kernels are 3-6e-3 from fp64, the reference's OWN fp32 arithmetic (ATen, same decisions) is
   2-5e-2 from it (1e-1 .. 2 at B = 64), while perturbing clouds and weights by half an fp32 ulp moves the fp64 gradients
   by 3e-4: BASELINE.md's "1e-3 relative" is not attainable by fp32 arithmetic on those tensors.  It IS attained,
   with margin, wherever no such residue is involved: the 14 tensors of the classifier head, conv3 / bn3 of the feature
   trunk (2e-6 .. 1e-4), every FC-stack kernel alone (5e-7, tests/test_gpu_head_train.py) and each trunk's backward
   under a generic upstream gradient (<= 1e-3, test_trunk_intermediates_large).

Bars (all absolute, none relative to another fp32 run):
* every gradient that is not such a residue: <= 1e-3 relative on every input;
* the residue tensors (STN3d, and conv1 / conv2 / bn1 / bn2 of the feature trunk): <= 1e-2 on clouds that differ from
  each other ("diverse": what crops of real scenes look like) and <= 5e-2 on the iid box / gauss clouds — tight enough
  that a kernel defect moving any gradient by a few percent fails;
* gradients that are exactly zero in exact arithmetic (a bias ahead of a train-mode BatchNorm; bn3.bias, whose
  upstream gradient sums to zero over the batch behind the next BatchNorm): <= 1e-4 of the largest gradient entry of the
  same layer's weight.
The forward is NOT excused: loss / log-probs / trans are compared with the oracle's own free-running forward at 1e-3."""
import os

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import build_model, state_dict_cpu, synth_cloud, oracle_train_step_on_device, capture_choices

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return (a - b).norm().item() / max(b.norm().item(), 1e-300)


def _gate(B, N, k, kind, dev, seed):
    m = build_model(N, k, 310 + seed, 5200 + seed).train()
    sd = state_dict_cpu(m)
    x = synth_cloud(B, N, 1800 + seed, kind)
    y = (torch.arange(B) * 7 % k).long()
    mg = m.to(dev)
    xd, yd = x.to(dev), y.to(dev)

    def run():
        logp, trans = mg(xd)
        loss = F.nll_loss(logp, yd)
        loss.backward()
        return loss.detach(), logp.detach(), trans.detach()

    (loss, logp, trans), ch = capture_choices(run)
    torch.cuda.synchronize()
    assert "libpngpd.so" in open("/proc/self/maps").read()
    grads = {n: p.grad.detach().cpu().clone() for n, p in mg.named_parameters()}
    # the same step through the fused entries must give the same bits (what the CLI runs)
    m2 = build_model(N, k, 310 + seed, 5200 + seed).train().to(dev)
    lp2, _ = m2(xd)
    F.nll_loss(lp2, yd).backward()
    for n, p in m2.named_parameters():
        assert torch.equal(p.grad.cpu(), grads[n]), n
    del m2, lp2
    torch.cuda.empty_cache()
    # free-running oracle forward (no choices): the flip-free quantities
    loss_f, logp_f, trans_f, _, _ = oracle_train_step_on_device(sd, x, y, torch.float64, dev)
    assert abs(loss.item() - loss_f.item()) <= 1e-3 * max(1.0, abs(loss_f.item()))
    assert (logp.cpu() - logp_f).abs().max().item() <= 1e-3 and (trans.cpu() - trans_f).abs().max().item() <= 1e-3
    # the oracle at the HIP run's discrete decisions
    loss_r, logp_r, trans_r, g64, _ = oracle_train_step_on_device(sd, x, y, torch.float64, dev, choices=ch)
    # how many decisions differ from the oracle's own is reported, not asserted (it is what makes the plain
    # comparison flip-limited); the imposed evaluation must still reproduce the forward
    assert abs(loss_r.item() - loss_f.item()) <= 1e-5 and (logp_r - logp_f).abs().max().item() <= 1e-4
    worst, rows, zeros = ("", 0.0), [], []
    for n, g in grads.items():
        ref = g64[n]
        if ".bias" in n and ("conv" in n or n.endswith("fc1.bias") or n.endswith("fc2.bias")):
            # a bias ahead of a train-mode BatchNorm: exactly zero in exact arithmetic
            scale = max(v.double().abs().max().item() for v in g64.values())
            assert ref.double().abs().max().item() <= 1e-9 * max(scale, 1.0), n
            zeros.append((n, g.abs().max().item()))
            continue
        r = _rel(g, ref)
        rows.append((n, r))
        if r > worst[1]:
            worst = (n, r)
    short = lambda n: n.replace("feat.", "f.").replace("weight", "w").replace("bias", "b")
    extra = ""
    if os.environ.get("PNGPD_GATE_DIAG") == "1":
        # diagnosis only (never part of the bar): the oracle's own fp32 run at the same imposed decisions
        _, _, _, g32, _ = oracle_train_step_on_device(sd, x, y, torch.float32, dev, choices=ch)
        extra = " | ATen-fp32 at the same decisions: " + " ".join(f"{short(n)}:{_rel(g32[n], g64[n]):.1e}" for n, _ in rows)
        # conditioning of the function itself: the fp64 oracle with clouds and weights perturbed by half an fp32 ulp
        gen = torch.Generator().manual_seed(99)
        hu = 2.0 ** -24
        pert = lambda t: (t.double() * (1 + hu * (torch.randint(0, 2, t.shape, generator=gen).double() * 2 - 1))) \
            if t.is_floating_point() else t
        sdp = {n: (pert(v) if v.is_floating_point() and "running" not in n else v) for n, v in sd.items()}
        _, _, _, g64p, _ = oracle_train_step_on_device(sdp, pert(x), y, torch.float64, dev, choices=ch)
        extra += " | fp64 oracle, inputs +- half an fp32 ulp: " + " ".join(f"{short(n)}:{_rel(g64p[n], g64[n]):.1e}" for n, _ in rows)
    print(f"[gate B={B} N={N} k={k} {kind}] loss {loss.item():.6f} (oracle {loss_f.item():.6f}); "
          f"worst gradient {worst[0]} rel {worst[1]:.2e}; " + " ".join(f"{short(n)}:{r:.1e}" for n, r in rows) + extra)
    for n, v in zeros:
        assert v <= 1e-4, (n, v)
    iid = kind != "diverse"
    for n, r in rows:
        if n.endswith("bn3.bias"):
            # sum over the batch of dL/dpooled: zero behind the next BatchNorm unless the STN's ReLU clamped some entries
            wmax = g64[n.replace("bias", "weight")].double().abs().max().item()
            err = (grads[n].double() - g64[n].double()).abs().max().item()
            rbar = (5e-2 if iid else 1e-2) if n.startswith("feat.stn.") else 1e-3
            assert err <= 1e-4 * wmax + rbar * g64[n].double().abs().max().item(), (n, err, wmax)
            continue
        residue = n.startswith("feat.stn.") or n.startswith("feat.conv1") or n.startswith("feat.conv2") or \
            n.startswith("feat.bn1") or n.startswith("feat.bn2")
        bar = 1e-3 if not residue else (5e-2 if iid else 1e-2)
        assert r <= bar, (n, r, bar)


def test_gradient_gate_bench_size(cuda_device):
    """BASELINE configs[1]: B = N = 1024, the headline's iid box clouds."""
    free, _ = torch.cuda.mem_get_info()
    assert free > 150e9, "needs ~100 GB of HBM for the fp64 oracle's activations"
    _gate(1024, 1024, 2, "box", cuda_device, 1)


def test_gradient_gate_fullview_geometry(cuda_device):
    """BASELINE configs[3]'s geometry (N = 4096, 64 tiles per cloud) at the batch the device-side fp64 oracle holds."""
    free, _ = torch.cuda.mem_get_info()
    assert free > 150e9
    _gate(256, 4096, 2, "box", cuda_device, 2)


@pytest.mark.parametrize("B,N,k,kind", [(64, 750, 2, "box"), (128, 1024, 3, "diverse"), (33, 200, 3, "gauss"),
                                        (1024, 1024, 2, "diverse"), (256, 4096, 2, "diverse")])
def test_gradient_gate_small(B, N, k, kind, cuda_device):
    """The reference's own recipe (B = 64, N = 750), the 3-class variants, ragged tiles."""
    _gate(B, N, k, kind, cuda_device, 3 + B)
