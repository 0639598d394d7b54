"""GPU: the opt-in bf16x3 inference trunk (3-term split-bf16 products on the bf16 matrix cores) against the
reference's golden vectors and the exact-fp32 path.  Contract: log-probs within 1e-3 of the reference
(north_star); asserted here at 2e-4, argmax-exact where the margin exceeds 1e-3."""
import numpy as np
import pytest
import torch

from oracle import pointnet_oracle as po
from tests.helpers import golden_files, build_model, assert_checksums, state_dict_cpu, synth_cloud

pytestmark = pytest.mark.gpu
EVAL = golden_files("pointnet_eval_")


@pytest.fixture()
def x3():
    from pointnetgpd_amd.model import pointnet as pn
    pn.set_inference_precision("bf16x3")
    yield pn
    pn.set_inference_precision("fp32")


@pytest.mark.parametrize("path", EVAL, ids=lambda p: p.split("pointnet_eval_")[-1][:-4])
def test_golden_eval_x3(path, x3, cuda_device):
    fx = np.load(path)
    m = build_model(fx["num_points"], fx["k"], fx["seed_w"], fx["seed_bn"]).eval()
    assert_checksums(m, fx)
    m = m.to(cuda_device)
    x = torch.from_numpy(fx["x"]).to(cuda_device)
    with torch.no_grad():
        logp, trans = m(x)
        feat_pool, _ = m.feat(x)
    np.testing.assert_allclose(trans.cpu().numpy(), fx["trans"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(feat_pool.cpu().numpy(), fx["feat_pool"], atol=5e-4, rtol=5e-4)
    np.testing.assert_allclose(logp.cpu().numpy(), fx["logp"], atol=2e-4, rtol=0)
    ref = torch.from_numpy(fx["logp"])
    margin = (ref.max(1)[0] - ref.kthvalue(ref.shape[1] - 1, 1)[0]) > 1e-3
    assert (logp.argmax(1).cpu()[margin] == ref.argmax(1)[margin]).all()


@pytest.mark.parametrize("B,N,k", [(64, 750, 2), (5, 1000, 3), (1, 500, 3), (3, 129, 2), (2, 1, 2)])
def test_x3_vs_fp32_and_oracle(B, N, k, x3, cuda_device):
    m = build_model(N, k, 140 + B, 5350 + B).eval()
    sd = state_dict_cpu(m)
    x = synth_cloud(B, N, 1500 + B, "box")
    with torch.no_grad():
        lp_ref, tr_ref = po.forward_torch(sd, x)
    mg = m.to(cuda_device)
    with torch.no_grad():
        lp3, tr3 = mg(x.to(cuda_device))
        x3.set_inference_precision("fp32")
        lp32, tr32 = mg(x.to(cuda_device))
        x3.set_inference_precision("bf16x3")
    np.testing.assert_allclose(lp3.cpu().numpy(), lp_ref.numpy(), atol=2e-4, rtol=0)
    np.testing.assert_allclose(tr3.cpu().numpy(), tr_ref.numpy(), atol=2e-4, rtol=0)
    np.testing.assert_allclose(lp3.cpu().numpy(), lp32.cpu().numpy(), atol=1e-4, rtol=0)
    assert not torch.equal(lp3, lp32) or B * N < 4      # it really is a different arithmetic


def test_x3_full_size_properties(x3, cuda_device):
    B, N = 1024, 1024
    m = build_model(N, 2, 150, 5360).eval().to(cuda_device)
    xg = synth_cloud(B, N, 1777, "box").to(cuda_device)
    with torch.no_grad():
        lp, tr = m(xg)
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(1)).to(cuda_device)
        lp_p, tr_p = m(xg[:, :, perm].contiguous())
        x3.set_inference_precision("fp32")
        lp32, _ = m(xg)
        x3.set_inference_precision("bf16x3")
    assert torch.equal(lp, lp_p) and torch.equal(tr, tr_p)        # point-order invariance stays bitwise
    assert (lp - lp32).abs().max().item() < 1e-4
    assert (lp.argmax(1) == lp32.argmax(1)).float().mean().item() > 0.999
