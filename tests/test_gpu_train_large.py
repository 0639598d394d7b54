"""GPU parity of the training path at the sizes the bench runs and in every workgroup-split regime
(S = 1, 1 < S < T, S = T), against the fp64 oracle executed on the device through ATen.

Why a separate file: round 1's parity cases stopped at B = 32, where the launch geometry differs from the benched
B = N = 1024 (one workgroup per cloud, 16 tiles per workgroup).  Here:

* flip-free quantities (loss, log-probs, trans, running statistics) are held to BASELINE.md §4's 1e-3;
* whole-model gradients keep the flip-aware bound of tests/helpers.grad_tol (an arg-max / ReLU sitting within fp32
  round-off of its threshold legitimately flips between two fp32 implementations), with the fp32 yardstick being the
  oracle's OWN fp32 run on the same device; the observed worst ratio is printed.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import build_model, state_dict_cpu, synth_cloud, grad_tol, oracle_train_step_on_device

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return (a - b).norm().item() / max(b.norm().item(), 1e-30)


def _run_case(B, N, k, dev, target_blocks=None, expect=None):
    from pointnetgpd_amd import ops
    m = build_model(N, k, 300 + B % 97, 5100 + N).train()
    sd = state_dict_cpu(m)
    x = synth_cloud(B, N, 1700 + B, "box")
    y = (torch.arange(B) * 7 % k).long()
    loss_ref, logp_ref, trans_ref, g64, stats_ref = oracle_train_step_on_device(sd, x, y, torch.float64, dev)
    _, _, _, g32, _ = oracle_train_step_on_device(sd, x, y, torch.float32, dev)
    old = ops.TRAIN_TARGET_BLOCKS
    if target_blocks is not None:
        ops.TRAIN_TARGET_BLOCKS = target_blocks
    try:
        S, T = ops.train_splits(B, N), (N + 63) // 64
        if expect == "one":
            assert S == 1 and T >= 1
        elif expect == "mid":
            assert 1 < S < T, (S, T)
        elif expect == "all":
            assert S == T and T > 1, (S, T)
        m = m.to(dev)
        logp, trans = m(x.to(dev))
        loss = F.nll_loss(logp, y.to(dev))
        loss.backward()
        torch.cuda.synchronize()
    finally:
        ops.TRAIN_TARGET_BLOCKS = old
    assert "libpngpd.so" in open("/proc/self/maps").read()
    # flip-free quantities: 1e-3
    assert abs(loss.item() - loss_ref.item()) <= 1e-3 * max(1.0, abs(loss_ref.item()))
    np.testing.assert_allclose(logp.detach().cpu().numpy(), logp_ref.numpy(), atol=1e-3, rtol=1e-3)
    np.testing.assert_allclose(trans.detach().cpu().numpy(), trans_ref.numpy(), atol=1e-3, rtol=1e-3)
    assert (logp.argmax(1).cpu() == logp_ref.argmax(1)).float().mean().item() >= 0.999
    cur = m.state_dict()
    for n, v in stats_ref.items():
        np.testing.assert_allclose(cur[n].cpu().numpy(), v.float().numpy(), atol=2e-5, rtol=1e-3, err_msg=n)
    worst = ("", 0.0, 0.0)
    for n, p in m.named_parameters():
        ref = g64[n]
        if ref.double().norm().item() < 1e-9:
            assert p.grad.abs().max().item() < 1e-4, n
            continue
        r, r32 = _rel(p.grad.cpu(), ref), _rel(g32[n], ref)
        tol = grad_tol(B, r32)
        if r / tol > worst[1]:
            worst = (n, r / tol, r)
        assert r < tol, (n, r, r32, tol)
    print(f"[B={B} N={N} k={k} S={S}/{T}] loss {loss.item():.6f} (ref {loss_ref.item():.6f}); worst gradient: {worst[0]} "
          f"rel err {worst[2]:.2e} = {worst[1]:.2f} of its bound")


@pytest.mark.parametrize("B,N,k,target,expect", [
    (320, 64, 2, None, "one"),      # B >= 256, a single tile
    (256, 128, 3, 512, "all"),      # S = T = 2
    (300, 512, 2, None, "mid"),     # 1 < S(=4) < T(=8)
    (24, 750, 3, 72, "mid"),        # forced split S = 3 of T = 12, ragged last tile
    (1024, 200, 2, None, "one"),    # S = 1 with 4 tiles, the last one ragged
])
def test_train_step_split_regimes(B, N, k, target, expect, cuda_device):
    _run_case(B, N, k, cuda_device, target, expect)


def test_train_step_bench_size(cuda_device):
    """The benched configuration itself: B = N = 1024 (BASELINE configs[1]), fp64 oracle on the device."""
    free, _ = torch.cuda.mem_get_info()
    assert free > 150e9, "needs ~100 GB of HBM for the fp64 oracle's activations"
    _run_case(1024, 1024, 2, cuda_device, None, "one")


@pytest.mark.parametrize("which", ["feat", "stn"])
def test_trunk_intermediates_large(which, cuda_device, pass_sequencing):
    """Kernel-level check at S = 1 with many tiles per workgroup (B = 256, N = 1024): every accumulated quantity of
    a trunk's backward vs the fp64 pass-structured prototype fed the same trans and upstream gradient.
    ``feat``: the PointNetfeat trunk (input transform, no ReLU before the max, pointnet.py:140-149); ``stn``: the
    STN3d trunk (no transform, ReLU BEFORE the max — ``dp`` is masked by ``pooled > 0``, pointnet.py:29-33)."""
    from pointnetgpd_amd import train
    from tests.test_gpu_train import _trunk_params
    from tests.train_algo_prototype import trunk_fwd, trunk_bwd
    B, N, k = 256, 1024, 2
    m = build_model(N, k, 96, 4516).train()
    if which == "stn":
        # the max over 1024 points of a normalised channel is ~ +3 sigma, so with the recipe's beta ~ N(0, 0.1) the
        # ReLU behind bn3 never clamps a pooled value; shift every other channel's beta so that the mask really bites
        with torch.no_grad():
            m.feat.stn.bn3.bias[::2] -= 3.0
    x = synth_cloud(B, N, 916, "box") * 4.0
    y = (torch.arange(B) * 7 % k).long()
    P = _trunk_params(m.feat if which == "feat" else m.feat.stn)
    mg = m.to(cuda_device)
    caps = []
    orig = train.TrunkTrainFn.backward

    def wrapped(ctx, dp):
        train.DEBUG_STASH = {}
        out = orig(ctx, dp)
        caps.append(dict(train.DEBUG_STASH))
        train.DEBUG_STASH = None
        return out

    train.TrunkTrainFn.backward = staticmethod(wrapped)
    try:
        logp, trans = mg(x.to(cuda_device))
        F.nll_loss(logp, y.to(cuda_device)).backward()
    finally:
        train.TrunkTrainFn.backward = orig
    feat = caps[0] if which == "feat" else caps[1]   # backward order: the feat trunk first, then the STN trunk
    T = trans.detach().double().cpu()
    dev = cuda_device
    Pd = {n: v.to(dev) for n, v in P.items()}
    _, sv = trunk_fwd(x.double().to(dev), T.to(dev) if which == "feat" else None, Pd, relu_last=which == "stn")
    if which == "stn":
        # the masking itself: entries of dp whose pooled output the ReLU clamped must not reach any gradient
        dead = sv["y"] <= 0
        assert 0.05 < dead.double().mean().item() < 0.95, "the case must exercise the ReLU-before-max mask"
    flips = (feat["idx"].long() != sv["idx"])
    assert flips.double().mean().item() < 1e-3
    # An arg-max that flips at an fp32 near-tie moves one upstream gradient entry to another point with (to round-off)
    # the same z value: both are valid sub-gradients, but they differ by O(1) in that point's rows (one flip in 262,144
    # maxima is already 2.7e-3 of the norm of g2).  The backward under test is therefore compared with the fp64
    # backward evaluated AT THE SAME arg-max points.
    sv["idx"] = feat["idx"].long()
    g = trunk_bwd(feat["dp"], Pd, sv)
    dbg = g["_dbg"]
    for kx in ["dg3", "dbe3", "S2", "sh", "G", "A", "cvec", "a1", "a2", "c1", "c2", "Rb", "g2buf"]:
        tol = 5e-3 if kx in ("a1", "a2", "c1", "c2", "Rb") else 1e-3       # cancelling batch sums: 5e-3
        r = _rel(feat[kx].cpu(), dbg[kx].cpu())
        assert r < tol, (kx, r)
    for kx, ky in [("dW1", "W1"), ("dW2", "W2"), ("dW3", "W3")] + ([("dT", "T")] if which == "feat" else []):
        r = _rel(feat[kx].cpu(), g[ky].cpu())
        print(f"{which} {kx}: rel {r:.3e}")
        assert r < 1e-3, (kx, r)
