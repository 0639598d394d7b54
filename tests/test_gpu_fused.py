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
    from pointnetgpd_amd import optim
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
                assert p.grad is optim.grad_view(p), n
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


def test_flat_adam_checkpoint_round_trip_then_foreign_optimizer(cuda_device, tmp_path):
    """the flat gradient views live in the optimizer, not on the Parameters.  A whole-module
    checkpoint of a FlatAdam-trained model is one 6.4 MB storage (no pickled gradient buffer, plain
    ``_rebuild_parameter``), and the loaded model — or the same model after its FlatAdam was replaced by
    torch.optim.Adam — trains: the fused backward returns gradients through autograd, ``p.grad`` is set, parameters move."""
    import gc
    import os
    from pointnetgpd_amd import mains, optim, install_reference_aliases
    from pointnetgpd_amd.optim import FlatAdam
    B, N, k = 8, 128, 2
    m = build_model(N, k, 43, 3501).to(cuda_device).train()
    x = synth_cloud(B, N, 703, "box").to(cuda_device)
    y = (torch.arange(B) % k).long().to(cuda_device)
    opt = FlatAdam(m.parameters(), lr=0.005)
    opt.zero_grad(); F.nll_loss(m(x)[0], y).backward(); opt.step()
    assert not any("_pngpd" in a for p in m.parameters() for a in vars(p))
    path = str(tmp_path / "flat.model")
    mains.save_model(m, path)
    assert os.path.getsize(path) < 7.5e6, os.path.getsize(path)          # parameters + buffers, not twice that
    install_reference_aliases()
    back = torch.load(path, map_location=cuda_device, weights_only=False).train()
    for p in back.parameters():
        assert optim.grad_view(p) is None and p.grad is None
    tor = torch.optim.Adam(back.parameters(), lr=0.005)
    w0 = {n: p.detach().clone() for n, p in back.named_parameters()}
    tor.zero_grad(); F.nll_loss(back(x)[0], y).backward()
    assert all(p.grad is not None for p in back.parameters())
    tor.step()
    assert any((p.detach() - w0[n]).abs().max().item() > 1e-4 for n, p in back.named_parameters())
    # the ORIGINAL model with its FlatAdam dropped and a torch optimizer in its place
    ref_g = {n: p.grad.detach().clone() for n, p in back.named_parameters()}
    del opt
    gc.collect()
    for p in m.parameters():
        assert optim.grad_view(p) is None                 # the registry entry died with the optimizer
    tor2 = torch.optim.Adam(m.parameters(), lr=0.005)
    tor2.zero_grad(set_to_none=True)
    w1 = m.fc3.weight.detach().clone()
    F.nll_loss(m(x)[0], y).backward()
    assert all(p.grad is not None for p in m.parameters())
    tor2.step()
    assert (m.fc3.weight.detach() - w1).abs().max().item() > 1e-5
    assert set(ref_g) == {n for n, _ in m.named_parameters()}


def test_flat_adam_partial_backward_and_double_write(cuda_device):
    """The fused backward overwrites gradient slices.  (i) A slice written twice before step() raises instead of
    silently keeping the last contribution; (ii) a fused piece that took no part in a backward (its slices hold the
    PREVIOUS step's gradient) is left untouched by step() — parameters, moments — exactly as ``torch.optim.Adam`` skips
    a parameter whose ``.grad`` is None; (iii) ``p.grad`` dropped by foreign code (``set_to_none``) falls back to
    autograd and step() picks the gradient up."""
    from pointnetgpd_amd import optim
    from pointnetgpd_amd.optim import FlatAdam
    B, N, k = 8, 128, 2
    m = build_model(N, k, 44, 3502).to(cuda_device).train()
    x = synth_cloud(B, N, 704, "box").to(cuda_device)
    y = (torch.arange(B) % k).long().to(cuda_device)
    opt = FlatAdam(m.parameters(), lr=0.005)
    opt.zero_grad(); F.nll_loss(m(x)[0], y).backward()
    with pytest.raises(RuntimeError, match="written twice"):
        F.nll_loss(m(x)[0], y).backward()
    opt.zero_grad()
    # (ii) only the feature extractor takes part: the head is skipped, the feature extractor moves
    F.nll_loss(m(x)[0], y).backward(); opt.step()
    head_before = {n: p.detach().clone() for n, p in m.named_parameters() if not n.startswith("feat.")}
    mom_before = opt.state[m.fc3.weight]["exp_avg"].clone()
    feat_before = m.feat.conv3.weight.detach().clone()
    assert m.fc3.weight.grad.abs().max().item() > 0
    opt.zero_grad()
    feat, _ = m.feat(x)
    feat.square().mean().backward()
    opt.step()
    for n, p in m.named_parameters():
        if not n.startswith("feat."):
            assert torch.equal(p.detach(), head_before[n]), n              # no momentum-only drift
    assert torch.equal(opt.state[m.fc3.weight]["exp_avg"], mom_before)
    assert (m.feat.conv3.weight.detach() - feat_before).abs().max().item() > 0
    assert (m.feat.stn.fc3.weight.grad.abs().max().item()) > 0
    # (iii) gradients dropped by the module's own zero_grad: autograd path, then adopted by step()
    m.zero_grad(set_to_none=True)
    assert all(optim.grad_view(p) is None for p in m.parameters())
    w0 = m.feat.conv3.weight.detach().clone()
    F.nll_loss(m(x)[0], y).backward()
    g = m.feat.conv3.weight.grad.detach().clone()
    opt.step()
    assert m.feat.conv3.weight.grad is optim.grad_view(m.feat.conv3.weight)
    assert torch.equal(m.feat.conv3.weight.grad, g)
    assert (m.feat.conv3.weight.detach() - w0).abs().max().item() > 1e-5


def test_flat_adam_state_dict_after_graph_replays(cuda_device):
    """a captured graph advances only the device-resident step counter — state_dict() reports it; and
    load_state_dict() FILLS the device lr tensor a captured graph reads instead of replacing it."""
    from pointnetgpd_amd.optim import FlatAdam
    from pointnetgpd_amd.train import GraphedTrainStep
    B, N, k = 8, 128, 2
    m = build_model(N, k, 45, 3503).to(cuda_device).train()
    x = synth_cloud(B, N, 705, "box").to(cuda_device)
    y = (torch.arange(B) % k).long().to(cuda_device)
    step = GraphedTrainStep(m, B, N, lr=0.005)
    for _ in range(4):
        step(x, y)
    sd = step.optimizer.state_dict()
    assert {float(v["step"]) for v in sd["state"].values()} == {4.0}
    lr_t = step.optimizer.param_groups[0]["lr"]
    sd["param_groups"][0]["lr"] = 0.00125
    step.optimizer.load_state_dict(sd)
    assert step.optimizer.param_groups[0]["lr"] is lr_t and abs(lr_t.item() - 0.00125) < 1e-9
    assert step.optimizer._step == 4 and step.optimizer.step_dev.item() == 4.0


def test_struct_sizes_match_header(cuda_device):
    import ctypes
    from pointnetgpd_amd import _lib
    lib = _lib.load()
    lib.pngpd_struct_bytes.restype = ctypes.c_size_t
    assert lib.pngpd_struct_bytes(0) == ctypes.sizeof(_lib.TrunkTrainArgs)
    assert lib.pngpd_struct_bytes(1) == ctypes.sizeof(_lib.HeadTrainArgs)


def test_flat_adam_with_a_plain_autograd_parameter_next_to_fused_pieces(cuda_device):
    """a parameter FlatAdam owns but no fused piece writes — a custom head on a PointNetfeat —
    receives its gradient by autograd accumulation into its flat view.  It must train exactly like under
    torch.optim.Adam: neither wiped before the update nor summed over steps."""
    import copy
    from pointnetgpd_amd.model.pointnet import PointNetfeat
    from pointnetgpd_amd.optim import FlatAdam
    B, N = 8, 128
    torch.manual_seed(3)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.feat = PointNetfeat(num_points=N, input_chann=3, global_feat=True)
            self.out = torch.nn.Linear(1024, 4)

        def forward(self, x):
            g, _ = self.feat(x)
            return self.out(g)

    a = Net().to(cuda_device).train()
    b = copy.deepcopy(a)
    oa, ob = FlatAdam(a.parameters(), lr=0.005), torch.optim.Adam(b.parameters(), lr=0.005, foreach=False, fused=False)
    w0 = a.out.weight.detach().clone()
    for i in range(4):
        x = synth_cloud(B, N, 720 + i, "box").to(cuda_device)
        t = torch.randn(B, 4, generator=torch.Generator().manual_seed(i)).to(cuda_device)
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad()
            (m(x) - t).square().mean().backward()
            if i == 0:
                assert m.out.weight.grad.abs().max().item() > 0
            o.step()
        # same gradient on both sides this step (no stale sum, no wipe): Adam's first steps are sign-like, so compare
        # the parameters tightly
        assert torch.allclose(a.out.weight, b.out.weight, atol=2e-5, rtol=0), i
        assert torch.allclose(a.out.bias, b.out.bias, atol=2e-5, rtol=0), i
    assert (a.out.weight.detach() - w0).abs().max().item() > 1e-3            # it trained
    assert torch.allclose(a.feat.conv1.weight, b.feat.conv1.weight, atol=5e-4, rtol=0)
