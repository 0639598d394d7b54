"""GPU parity: the HIP inference path (through the C ABI) vs the golden vectors recorded from
the reference and vs the CPU oracle on seeded inputs.  Tolerance: log-probs within 1e-3
(BASELINE north_star), argmax-exact; we assert a tighter 2e-4 on log-probs/trans here."""
import numpy as np
import pytest
import torch

from oracle import pointnet_oracle as po
from tests.helpers import golden_files, build_model, assert_checksums, state_dict_cpu, synth_cloud

pytestmark = pytest.mark.gpu

ATOL_LOGP = 2e-4   # north_star bound is 1e-3
EVAL = golden_files("pointnet_eval_")


def _loaded_native():
    maps = open("/proc/self/maps").read()
    assert "libpngpd.so" in maps, "native library not loaded — GPU tests must run the HIP path"


@pytest.mark.parametrize("path", EVAL, ids=lambda p: p.split("pointnet_eval_")[-1][:-4])
def test_golden_eval(path, cuda_device):
    fx = np.load(path)
    m = build_model(fx["num_points"], fx["k"], fx["seed_w"], fx["seed_bn"]).eval()
    assert_checksums(m, fx)
    m = m.to(cuda_device)
    x = torch.from_numpy(fx["x"]).to(cuda_device)
    with torch.no_grad():
        logp, trans = m(x)
        stn_pool = m.feat.stn._forward_hip_infer  # noqa: F841 (exercised below through feat)
        feat_pool, trans2 = m.feat(x)
    _loaded_native()
    np.testing.assert_allclose(trans.cpu().numpy(), fx["trans"], atol=ATOL_LOGP, rtol=0)
    np.testing.assert_allclose(trans2.cpu().numpy(), fx["trans"], atol=ATOL_LOGP, rtol=0)
    np.testing.assert_allclose(feat_pool.cpu().numpy(), fx["feat_pool"], atol=2e-4, rtol=2e-4)
    np.testing.assert_allclose(logp.cpu().numpy(), fx["logp"], atol=ATOL_LOGP, rtol=0)
    assert (logp.argmax(1).cpu().numpy() == fx["logp"].argmax(1)).all()


def test_trunk_kernel_vs_oracle_pooled(cuda_device):
    """Kernel-level: pngpd_trunk_fwd_infer against the numpy fp64 trunk (both trunks, ragged N)."""
    from pointnetgpd_amd import ops
    from pointnetgpd_amd.model import pointnet as pn
    for n in (1, 63, 64, 65, 200, 750):
        m = build_model(n, 2, 31, 4340).eval()
        sd = state_dict_cpu(m)
        x = synth_cloud(3, n, 300 + n, "gauss")
        _, _, inter = po.forward_numpy(sd, x.numpy(), return_intermediates=True)
        m = m.to(cuda_device)
        xg = x.to(cuda_device)
        with torch.no_grad():
            stn = ops.trunk_fwd_infer(xg, None, *pn._trunk_infer_weights(m.feat.stn, cuda_device), relu_last=True)
            tr = torch.from_numpy(inter["trans"]).float().to(cuda_device)
            feat = ops.trunk_fwd_infer(xg, tr, *pn._trunk_infer_weights(m.feat, cuda_device), relu_last=False)
        np.testing.assert_allclose(stn.cpu().numpy(), inter["stn_pool"], atol=1e-4, rtol=1e-4)
        np.testing.assert_allclose(feat.cpu().numpy(), inter["feat_pool"], atol=1e-4, rtol=1e-4)


def test_fc_kernel_vs_torch(cuda_device):
    """pngpd_fc_fwd epilogues vs a plain fp32 torch composite (ragged B and Nout)."""
    from pointnetgpd_amd import ops
    g = torch.Generator().manual_seed(5)
    for (B, K, Nout) in [(1, 256, 9), (5, 256, 2), (33, 1024, 512), (64, 512, 256), (70, 256, 3), (2, 256, 32),
                         (37, 500, 2), (64, 12, 130), (3, 4, 5), (40, 504, 40)]:     # K = 8 m + 4: the half-block tail
        a = torch.randn(B, K, generator=g); W = torch.randn(Nout, K, generator=g) / K ** 0.5
        bias = torch.randn(Nout, generator=g)
        ref = a.double() @ W.double().T + bias.double()
        ag, Wg, bg = a.to(cuda_device), W.to(cuda_device), bias.to(cuda_device)
        out = ops.fc_fwd(ag, Wg, bg, ops.EPI_NONE).cpu()
        np.testing.assert_allclose(out.numpy(), ref.numpy(), atol=1e-5, rtol=1e-5)
        out = ops.fc_fwd(ag, Wg, bg, ops.EPI_RELU).cpu()
        np.testing.assert_allclose(out.numpy(), ref.clamp(min=0).numpy(), atol=1e-5, rtol=1e-5)
        if Nout <= 32:
            out = ops.fc_fwd(ag, Wg, bg, ops.EPI_LOG_SOFTMAX).cpu()
            np.testing.assert_allclose(out.numpy(), torch.log_softmax(ref, -1).numpy(), atol=1e-5, rtol=1e-5)
        if Nout == 9:
            out = ops.fc_fwd(ag, Wg, bg, ops.EPI_ADD_IDEN3).cpu()
            np.testing.assert_allclose(out.numpy(), (ref + torch.eye(3).double().view(1, 9)).numpy(), atol=1e-5, rtol=1e-5)


def test_fold_kernel(cuda_device):
    from pointnetgpd_amd import ops
    g = torch.Generator().manual_seed(6)
    C, K = 128, 64
    W = torch.randn(C, K, generator=g); b = torch.randn(C, generator=g)
    gam = torch.randn(C, generator=g); bet = torch.randn(C, generator=g)
    mu = torch.randn(C, generator=g); var = torch.rand(C, generator=g) + 0.5
    s = gam.double() / torch.sqrt(var.double() + 1e-5)
    wf, bf = ops.fold_conv_bn(*[t.to(cuda_device) for t in (W, b, gam, bet, mu, var)])
    np.testing.assert_allclose(wf.cpu().numpy(), (W.double() * s[:, None]).float().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(bf.cpu().numpy(), ((b.double() - mu.double()) * s + bet.double()).float().numpy(), rtol=1e-6, atol=1e-6)
    wp, _ = ops.fold_conv_bn(*[t.to(cuda_device) for t in (W, b, gam, bet, mu, var)], layout=ops.LAYOUT_MFMA_B)
    # unpack: [cb][kb][h][j][t] -> W[cb*32+j][kb*8+h*4+t]
    un = wp.cpu().view(C // 32, K // 8, 2, 32, 4).permute(0, 3, 1, 2, 4).reshape(C, K)
    np.testing.assert_array_equal(un.numpy(), wf.cpu().numpy())


@pytest.mark.parametrize("B,N,k", [(64, 750, 2), (7, 1000, 3), (1, 500, 3), (130, 64, 2)])
def test_model_vs_oracle(B, N, k, cuda_device):
    """Whole forward vs the oracle (same ATen ops as the reference) on seeded inputs."""
    m = build_model(N, k, 40 + B, 4350 + B).eval()
    sd = state_dict_cpu(m)
    x = synth_cloud(B, N, 500 + B, "box")
    with torch.no_grad():
        lp_ref, tr_ref = po.forward_torch(sd, x)
    m = m.to(cuda_device)
    with torch.no_grad():
        lp, tr = m(x.to(cuda_device))
    np.testing.assert_allclose(tr.cpu().numpy(), tr_ref.numpy(), atol=ATOL_LOGP, rtol=0)
    np.testing.assert_allclose(lp.cpu().numpy(), lp_ref.numpy(), atol=ATOL_LOGP, rtol=0)
    margin = (lp_ref.max(1)[0] - lp_ref.kthvalue(lp_ref.shape[1] - 1, 1)[0]) > 1e-3
    assert (lp.argmax(1).cpu()[margin] == lp_ref.argmax(1)[margin]).all()


def test_full_size_properties(cuda_device):
    """BASELINE config 2 size (B=1024,N=1024): size-independent exact properties + the fp64 oracle
    over all 1024 clouds (run on the device)."""
    B, N = 1024, 1024
    m = build_model(N, 2, 50, 4360).eval()
    sd = state_dict_cpu(m)
    x = synth_cloud(B, N, 777, "box")
    mg = m.to(cuda_device)
    xg = x.to(cuda_device)
    with torch.no_grad():
        lp, tr = mg(xg)
        # (1) permutation invariance over points: bitwise (each point's features do not depend
        #     on its position; max is order-independent)
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(1)).to(cuda_device)
        lp_p, tr_p = mg(xg[:, :, perm].contiguous())
        assert torch.equal(lp, lp_p) and torch.equal(tr, tr_p)
        # (2) batch equivariance: bitwise
        bperm = torch.randperm(B, generator=torch.Generator().manual_seed(2)).to(cuda_device)
        lp_b, tr_b = mg(xg[bperm].contiguous())
        assert torch.equal(lp[bperm], lp_b) and torch.equal(tr[bperm], tr_b)
        # (3) log-probs normalise
        assert torch.allclose(lp.exp().sum(1), torch.ones(B, device=cuda_device), atol=1e-5)
        # (4) determinism
        lp2, _ = mg(xg)
        assert torch.equal(lp, lp2)
    # (5) the oracle over the WHOLE batch (fp64, run on the device): every one of the B clouds, not a slice
    from tests.helpers import oracle_forward_on_device
    lp_ref, tr_ref = oracle_forward_on_device(sd, x, cuda_device)
    np.testing.assert_allclose(lp.cpu().numpy(), lp_ref.numpy(), atol=ATOL_LOGP, rtol=0)
    np.testing.assert_allclose(tr.cpu().numpy(), tr_ref.numpy(), atol=ATOL_LOGP, rtol=0)
    assert (lp.argmax(1).cpu() == lp_ref.argmax(1)).all()


def test_duplicate_points_do_not_change_pool(cuda_device):
    """Sampling with replacement (dataset.py:443) duplicates points; the pooled feature of a cloud
    equals the pooled feature of the cloud with points repeated."""
    m = build_model(128, 2, 60, 4370).eval()
    m2 = build_model(256, 2, 60, 4370).eval()
    x = synth_cloud(4, 128, 901, "box")
    with torch.no_grad():
        f1, t1 = m.to(cuda_device).feat(x.to(cuda_device))
        f2, t2 = m2.to(cuda_device).feat(torch.cat([x, x], dim=2).to(cuda_device))
    assert torch.equal(f1, f2) and torch.equal(t1, t2)


def test_wrong_num_points_raises(cuda_device):
    m = build_model(64, 2, 1, -1).eval().to(cuda_device)
    with pytest.raises(RuntimeError, match="num_points"):
        m(torch.zeros(2, 3, 65, device=cuda_device))
    with pytest.raises(RuntimeError, match="Float"):
        m(torch.zeros(2, 3, 64, device=cuda_device, dtype=torch.float64))


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16"])
@pytest.mark.parametrize("B,N", [(6, 200), (40, 1024)])       # one workgroup per cloud / several workgroups per cloud
def test_non_finite_coordinates_propagate(B, N, precision, cuda_device):
    """A NaN coordinate poisons exactly its own cloud's outputs, as through the reference's max_pool1d / torch.max
    (NaN-propagating) — although the kernels reduce with fmaxf (NaN-dropping) and ReLU maps NaN to 0.  The other
    clouds of the batch are bit-identical to a clean run (eval-mode BatchNorm couples nothing)."""
    from pointnetgpd_amd.model import pointnet as pn
    m = build_model(N, 3, 41, 4141).eval()
    x = synth_cloud(B, N, 808, "box")
    xb = x.clone()
    xb[1, 2, N // 2] = float("nan")
    xb[B - 1, 0, N - 1] = float("inf")
    with torch.no_grad():
        ref_lp, ref_tr = m(xb)                            # CPU: the ATen composite the reference runs
    mg = m.to(cuda_device)
    pn.set_inference_precision(precision)
    try:
        with torch.no_grad():
            lp_clean, _ = mg(x.to(cuda_device))
            lp, tr = mg(xb.to(cuda_device))
    finally:
        pn.set_inference_precision("fp32")
    lp, tr, lp_clean = lp.cpu(), tr.cpu(), lp_clean.cpu()
    assert torch.isnan(ref_lp[1]).all() and torch.isnan(lp[1]).all() and torch.isnan(tr[1]).all()
    assert not torch.isfinite(lp[B - 1]).any() and not torch.isfinite(ref_lp[B - 1]).any()
    clean = [b for b in range(B) if b not in (1, B - 1)]
    assert torch.equal(lp[clean], lp_clean[clean]) and torch.isfinite(lp[clean]).all()
