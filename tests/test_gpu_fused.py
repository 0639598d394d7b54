"""The fused per-direction training entries (pngpd_trunk_train_fwd/_bwd, pngpd_head_train_fwd/_bwd) against the
pass-by-pass sequencing of the same kernels (bit for bit), and the flat Adam (pngpd_adam_flat / optim.FlatAdam) against
torch.optim.Adam — the reference's step, PointNetGPD/main_1v.py:72-76."""
import copy

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import build_model, synth_cloud

pytestmark = pytest.mark.gpu


def _step(model, x, y, sequencing, precision="fp32", fp32_side=False):
    from pointnetgpd_amd import train
    m = copy.deepcopy(model).train()
    train.set_sequencing(sequencing)
    train.set_train_precision(precision, fp32_side_passes=fp32_side)
    try:
        logp, trans = m(x)
        loss = F.nll_loss(logp, y)
        loss.backward()
    finally:
        train.set_sequencing("fused")
        train.set_train_precision("fp32")
    grads = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    bufs = {n: b.detach().clone() for n, b in m.named_buffers()}
    return loss.detach(), logp.detach(), trans.detach(), grads, bufs


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16", "bf16x3+side-fp32"])
@pytest.mark.parametrize("B,N,k", [(16, 750, 2), (5, 100, 3), (64, 1024, 2), (3, 64, 2)])
def test_fused_equals_passes_bitwise(B, N, k, mode, cuda_device):
    """One foreign call per direction == the kernels called one by one: loss, log-probs, trans, all 44 gradients and
    all 30 BatchNorm buffers are bit-identical (every shape class of the split rule: S = T, 1 < S < T, tails)."""
    prec, side = mode.split("+")[0], mode.endswith("side-fp32")
    m = build_model(N, k, 40 + B, 3300 + B).to(cuda_device)
    x = synth_cloud(B, N, 700 + B, "box").to(cuda_device)
    y = (torch.arange(B) * 7 % k).long().to(cuda_device)
    ref = _step(m, x, y, "passes", prec, side)
    got = _step(m, x, y, "fused", prec, side)
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]) and torch.equal(ref[2], got[2])
    for n in ref[3]:
        assert torch.equal(ref[3][n], got[3][n]), n
    for n in ref[4]:
        assert torch.equal(ref[4][n], got[4][n]), n


def test_fused_forward_without_backward(cuda_device):
    """no_grad / frozen parameters: the forward keeps nothing and still matches."""
    B, N, k = 8, 300, 2
    m = build_model(N, k, 41, 3400).to(cuda_device).train()
    x = synth_cloud(B, N, 701, "box").to(cuda_device)
    m2 = copy.deepcopy(m)
    logp, trans = m(x)
    with torch.no_grad():
        logp2, trans2 = m2(x)
    assert torch.equal(logp.detach(), logp2) and torch.equal(trans.detach(), trans2)
    for (n, a), (_, b) in zip(m.named_buffers(), m2.named_buffers()):
        assert torch.equal(a, b), n


def test_flat_adam_matches_torch_adam(cuda_device):
    """pngpd_adam_flat == torch.optim.Adam's single-tensor update, over several steps on the same gradients."""
    from pointnetgpd_amd.optim import FlatAdam
    torch.manual_seed(5)
    a = [torch.nn.Parameter(torch.randn(s, device=cuda_device)) for s in [(64, 3, 1), (64,), (1024, 128), (9,), (3, 256)]]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa, ob = FlatAdam(a, lr=0.005), torch.optim.Adam(b, lr=0.005, foreach=False, fused=False)
    for it in range(5):
        for p, q in zip(a, b):
            g = torch.randn_like(q) * (10.0 ** (-it))
            q.grad = g
            p.grad.copy_(g)
        oa.step(); ob.step()
        for p, q in zip(a, b):
            # one Adam update is lr-sized (5e-3); agreement to 1e-3 of that per step
            assert (p - q).abs().max().item() < 5e-6 * (it + 1), it
    sd = oa.state_dict()
    assert set(sd["state"][0]) >= {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 5.0
    for i, q in enumerate(b):
        assert torch.allclose(sd["state"][i]["exp_avg"], ob.state[q]["exp_avg"], rtol=1e-5, atol=1e-8)


def test_flat_adam_training_step_overwrites_flat_gradients(cuda_device):
    """With FlatAdam attached the fused backward writes every gradient into its slice of ONE buffer (bit-identical to
    the autograd-returned gradients), zero_grad is a no-op, and an eval forward after step() sees the new weights."""
    from pointnetgpd_amd.optim import FlatAdam
    B, N, k = 16, 750, 2
    m = build_model(N, k, 42, 3500).to(cuda_device)
    x = synth_cloud(B, N, 702, "box").to(cuda_device)
    y = (torch.arange(B) % k).long().to(cuda_device)
    ref = _step(m, x, y, "fused")
    mt = copy.deepcopy(m).train()
    m = m.train()
    opt = FlatAdam(m.parameters(), lr=0.005)
    tor = torch.optim.Adam(mt.parameters(), lr=0.005)
    m.eval()
    with torch.no_grad():
        before = m(x)[0].clone()
    m.train()
    for it in range(2):
        opt.zero_grad(); tor.zero_grad()
        logp, _ = m(x)
        F.nll_loss(logp, y).backward()
        if it == 0:
            for n, p in m.named_parameters():
                assert p.grad.data_ptr() == p._pngpd_grad.data_ptr(), n
                assert torch.equal(p.grad, ref[3][n]), n
        opt.step()
        lt, _ = mt(x)
        F.nll_loss(lt, y).backward()
        tor.step()
        assert torch.equal(logp.detach(), lt.detach()) if it == 0 else (logp - lt).abs().max().item() < 1e-3
        if it == 0:
            # identical gradients in, one Adam update each: agreement to 1e-3 of the lr-sized step.  (Later steps are
            # not comparable parameter by parameter: FC biases ahead of a train-mode BN have rounding-noise gradients,
            # and Adam turns the SIGN of that noise into a full lr-sized step.)
            for (n, p), (_, q) in zip(m.named_parameters(), mt.named_parameters()):
                assert (p - q).abs().max().item() < 5e-6, (n, (p - q).abs().max().item())
    m.eval()
    with torch.no_grad():
        after = m(x)[0]
    assert (after - before).abs().max().item() > 1e-4      # the fold cache saw the in-place update


def test_struct_sizes_match_header(cuda_device):
    import ctypes
    from pointnetgpd_amd import _lib
    lib = _lib.load()
    lib.pngpd_struct_bytes.restype = ctypes.c_size_t
    assert lib.pngpd_struct_bytes(0) == ctypes.sizeof(_lib.TrunkTrainArgs)
    assert lib.pngpd_struct_bytes(1) == ctypes.sizeof(_lib.HeadTrainArgs)
