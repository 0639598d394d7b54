"""Eval-mode weights are derived from the LIVE parameters on every forward (pngpd_fold_model, one launch per model):
no edit of a parameter or BatchNorm buffer can leave a stale folded copy behind — including in-place edits through
``.data``, which bump no version counter.  Every mutation idiom below must change the next eval
output, and that output must equal the oracle's on the mutated state (reference idiom replaced: whole-module reload,
main_1v.py:148-155; the reference itself reads the live tensors in every forward, pointnet.py:29-31,144-147)."""
import numpy as np
import pytest
import torch
from torch.nn.utils import parameters_to_vector, vector_to_parameters

from oracle import pointnet_oracle as po
from tests.helpers import build_model, state_dict_cpu, synth_cloud

pytestmark = pytest.mark.gpu


def _eval(m, x):
    with torch.no_grad():
        lp, tr = m(x)
    return lp.cpu(), tr.cpu()


def _oracle(m, x):
    with torch.no_grad():
        return po.forward_torch(state_dict_cpu(m), x.cpu())


def _mutations(m, dev):
    g = torch.Generator().manual_seed(5)

    def load_sd():
        other = build_model(m.num_points, 2, 77, 78)
        m.load_state_dict(other.state_dict())

    def adam_step():
        opt = torch.optim.Adam(m.parameters(), lr=0.05)
        for p in m.parameters():
            p.grad = torch.randn(p.shape, generator=g).to(dev)
        opt.step()
        for p in m.parameters():
            p.grad = None

    def flat_adam_step():
        from pointnetgpd_amd.optim import FlatAdam
        opt = FlatAdam(m.parameters(), lr=0.05)
        opt.flat_g.normal_()
        opt.step()

    def data_copy():                              # bumps NO version counter of the parameter
        w = m.feat.conv3.weight
        w.data.copy_(torch.randn(w.shape, generator=g).to(dev) * 0.05)

    def data_normal():                            # the init idiom the verdict names
        m.fc1.weight.data.normal_(0, 0.05)

    def vec_roundtrip():
        v = parameters_to_vector(m.parameters())
        vector_to_parameters(v * 1.25 + 0.001, m.parameters())

    def buffer_data():
        m.feat.stn.bn3.running_mean.data.add_(0.05)
        m.bn2.running_var.data.mul_(1.7)

    def set_():                                   # storage swap: the source ADDRESS changes -> the plan is rebuilt
        w = m.feat.stn.fc3.weight
        w.data = torch.randn(w.shape, generator=g).to(dev) * 0.1

    return [load_sd, adam_step, flat_adam_step, data_copy, data_normal, vec_roundtrip, buffer_data, set_]


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
def test_fold_cache_invalidation(cuda_device, mode):
    m = build_model(200, 2, 3, 4).eval().to(cuda_device)
    m.set_precision(mode)
    x = synth_cloud(6, 200, 9, "box").to(cuda_device)
    prev, _ = _eval(m, x)
    for mut in _mutations(m, cuda_device):
        mut()
        m.eval()
        lp, tr = _eval(m, x)
        ref_lp, ref_tr = _oracle(m, x)
        assert not torch.equal(lp, prev), f"{mut.__name__}: the eval output did not move (stale folded weights)"
        assert (lp - ref_lp).abs().max().item() < 2e-4, mut.__name__
        assert (tr - ref_tr).abs().max().item() < 2e-4, mut.__name__
        prev = lp


def test_fold_model_equals_per_layer_entries(cuda_device):
    """pngpd_fold_model's three layouts against the single-layer entries (pngpd_fold_conv_bn, pngpd_split_pack_bf16):
    bit-identical."""
    from pointnetgpd_amd import ops
    from pointnetgpd_amd.model import pointnet as pn
    m = build_model(64, 3, 11, 12).eval().to(cuda_device)
    for mod in (m.feat, m.feat.stn):
        w1, b1, w2, b2, w3, b3 = pn._trunk_infer_weights(mod, cuda_device)
        x1, _, x2, _, x3, _ = pn._trunk_infer_weights_x3(mod, cuda_device)
        for i, (w, b, xw) in enumerate(((w1, b1, x1), (w2, b2, x2), (w3, b3, x3)), 1):
            conv, bn = getattr(mod, f"conv{i}"), getattr(mod, f"bn{i}")
            lay = ops.LAYOUT_ROWMAJOR if i == 1 else ops.LAYOUT_MFMA_B
            rw, rb = ops.fold_conv_bn(conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                      eps=bn.eps, layout=lay)
            assert torch.equal(rw.reshape(-1), w.reshape(-1)) and torch.equal(rb, b)
            if i > 1:
                row, _ = ops.fold_conv_bn(conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                          eps=bn.eps, layout=ops.LAYOUT_ROWMAJOR)
                assert torch.equal(ops.split_pack_bf16(row).reshape(-1), xw.reshape(-1))


def test_one_fold_launch_per_forward(cuda_device):
    """The whole tree (feat, feat.stn, head) folds in ONE launch per eval forward; nested modules launch nothing."""
    from pointnetgpd_amd.model import pointnet as pn
    m = build_model(128, 2, 1, 2).eval().to(cuda_device)
    x = synth_cloud(4, 128, 3, "box").to(cuda_device)
    calls = []
    orig = pn._FoldPlan.launch
    pn._FoldPlan.launch = lambda self: (calls.append(1), orig(self))[1]
    try:
        _eval(m, x)
        _eval(m, x)
        assert len(calls) == 2
        with torch.no_grad():
            m.feat(x)
        assert len(calls) == 3
    finally:
        pn._FoldPlan.launch = orig
