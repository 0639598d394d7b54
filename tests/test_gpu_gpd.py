"""GPU parity of the GPD-baseline row against the records of the executed reference
(tests/golden/gpd_*.npz) and the oracle: projection images (bit-exact, floats included), depth registration and
back-projection (bit-exact), GPDClassifier forward (1e-5), and the drop-in ``cloudgen`` functions."""
import os

import numpy as np
import pytest
import torch

from oracle import gpd_oracle as go
from tests import synth_gpd
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_projection_images_bit_exact(cuda_device):
    from pointnetgpd_amd import gpd_ops
    fx = np.load(os.path.join(GOLDEN, "gpd_projection.npz"))
    tags = [str(t) for t in fx["tags"]]
    pts, nrm, off, widths = [], [], [0], []
    for tag in tags:                                       # all cases as ONE batch of grasps
        ind = fx[f"{tag}/in_ind"]
        pts.append(fx[f"{tag}/pc"][ind]); nrm.append(fx[f"{tag}/normals"][ind])
        off.append(off[-1] + len(ind)); widths.append(float(fx[f"{tag}/width"]))
    pts, nrm = np.concatenate(pts), np.concatenate(nrm)
    for chann in (3, 12):
        out = gpd_ops.project_grasps(pts, nrm, np.array(off, dtype=np.int32), np.array(widths), chann, cuda_device)
        assert "libpngpd.so" in open("/proc/self/maps").read()
        assert out.shape == (len(tags), 60, 60, chann) and out.dtype == torch.float64
        for g, tag in enumerate(tags):
            np.testing.assert_array_equal(out[g].cpu().numpy(), fx[f"{tag}/out{chann}"], err_msg=f"{tag} chann {chann}")


def test_projection_large_random_vs_oracle(cuda_device):
    """Many points per grasp (several 1,024-point chunks, voxels far above the 50-point cap) and ragged sizes."""
    from pointnetgpd_amd import gpd_ops
    rng = np.random.default_rng(5)
    sizes = [3000, 1, 0, 1025, 257]
    pts, nrm, off, widths = [], [], [0], []
    for m in sizes:
        w = rng.uniform(0.03, 0.085)
        p = rng.uniform(-1, 1, size=(m, 3)) * np.array([w / 4, w / 2, w / 4])
        if m > 100:
            p[: m // 3] = p[0] + rng.normal(size=(m // 3, 3)) * 1e-4          # one voxel with hundreds of points
        n = rng.normal(size=(m, 3))
        if m > 10:
            n[rng.choice(m, 5, replace=False), 1] = np.nan
        pts.append(p); nrm.append(n); off.append(off[-1] + m); widths.append(w)
    out = gpd_ops.project_grasps(np.concatenate(pts), np.concatenate(nrm), np.array(off, dtype=np.int32),
                                 np.array(widths), 12, cuda_device).cpu().numpy()
    for g, m in enumerate(sizes):
        if m == 0:
            assert not out[g].any()                      # the reference raises on an empty hand; we return zeros
            continue
        np.testing.assert_array_equal(out[g], go.project_pc(pts[g], nrm[g], widths[g], 12), err_msg=str(m))


@pytest.mark.parametrize("tag", ["small", "vga"])
def test_depth_registration_and_cloud(tag, cuda_device):
    from pointnetgpd_amd import gpd_ops, cloudgen
    fx = np.load(os.path.join(GOLDEN, "gpd_cloudgen.npz"))
    sc = synth_gpd.cloudgen_scene(tag)
    reg = gpd_ops.register_depth_map(sc["depth"], sc["rgb"].shape, sc["depthK"], sc["rgbK"], sc["H"], cuda_device)
    regn = reg.cpu().numpy()
    assert int((regn > 0).sum()) == int(fx[f"{tag}/reg_nonzero"])
    np.testing.assert_array_equal(regn.reshape(-1)[fx[f"{tag}/reg_pix"]], fx[f"{tag}/reg_val"])
    np.testing.assert_array_equal(regn, go.register_depth_map(sc["depth"], sc["rgb"].shape, sc["depthK"], sc["rgbK"], sc["H"]))
    regm = regn.copy(); regm[sc["mask"]] = 0
    xyz, col = gpd_ops.depth_map_to_cloud(regm, sc["rgbK"], sc["refFromRGB"], sc["objFromref"], rgb=sc["rgb"],
                                          device=cuda_device)
    assert xyz.shape[0] == int(fx[f"{tag}/cloud_len"])
    rows = fx[f"{tag}/cloud_rows"]
    np.testing.assert_array_equal(xyz.cpu().numpy()[rows], fx[f"{tag}/cloud_val"][:, :3])
    np.testing.assert_array_equal(col.cpu().numpy()[rows].astype(np.float64), fx[f"{tag}/cloud_val"][:, 3:])
    np.testing.assert_array_equal(xyz.cpu().numpy(), go.depth_map_to_cloud(regm, sc["rgbK"], sc["refFromRGB"], sc["objFromref"]))
    if tag == "small":                                    # the drop-in functions, numpy in / numpy out
        reg2 = cloudgen.registerDepthMap(sc["depth"], sc["rgb"], sc["depthK"], sc["rgbK"], sc["H"])
        np.testing.assert_array_equal(reg2, fx["small/registered"])
        reg2[sc["mask"]] = 0
        cloud = cloudgen.registeredDepthMapToPointCloud(reg2, sc["rgb"], sc["rgbK"], sc["refFromRGB"], sc["objFromref"])
        np.testing.assert_array_equal(cloud[0], fx["small/cloud"])
        org = cloudgen.registeredDepthMapToPointCloud(reg2, sc["rgb"], sc["rgbK"], sc["refFromRGB"], sc["objFromref"],
                                                      organized=True)
        np.testing.assert_array_equal(org, fx["small/cloud_organized"])      # executed reference, NaN where no depth


@pytest.mark.parametrize("chann", [3, 12])
def test_gpd_classifier_forward(chann, cuda_device):
    from pointnetgpd_amd.model.gpd import GPDClassifier
    fx = np.load(os.path.join(GOLDEN, "gpd_classifier.npz"))
    torch.manual_seed(100 + chann)
    m = GPDClassifier(chann).eval()
    x = torch.from_numpy(fx[f"x{chann}"])
    with torch.no_grad():
        logp = m.to(cuda_device)(x.to(cuda_device))
    assert "libpngpd.so" in open("/proc/self/maps").read()
    np.testing.assert_allclose(logp.cpu().numpy(), fx[f"logp{chann}"], atol=1e-5, rtol=0)
    # a larger batch against the oracle's functional restatement
    g = torch.Generator().manual_seed(9)
    xb = torch.rand(70, chann, 60, 60, generator=g)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = go.gpd_forward_torch(sd, xb)
        got = m(xb.to(cuda_device)).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-5, rtol=0)
    assert (got.argmax(1) == ref.argmax(1)).all()


@pytest.mark.parametrize("chann,B", [(3, 1), (3, 37), (12, 16), (12, 200)])
def test_gpd_classifier_training_on_hip(chann, B, cuda_device):
    """train() mode on a CUDA tensor runs libpngpd forward AND backward (gpd_ops.GPDNetFn; main_1v_gpd.py:97-106's
    ``loss = F.nll_loss(model(data), target); loss.backward()``): every parameter gradient against the same module in
    fp64 on the CPU (ATen), on projection-like images — mostly empty, so whole pooling windows tie at the bias and the
    first-position rule of ATen's max_pool2d decides where the gradient goes."""
    import torch.nn.functional as F
    from pointnetgpd_amd.model.gpd import GPDClassifier
    torch.manual_seed(40 + chann + B)
    m = GPDClassifier(chann)
    ref = GPDClassifier(chann).double()
    ref.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, chann, 60, 60, generator=g)
    x = x * (torch.rand(B, 1, 60, 60, generator=g) < 0.3)             # 70 % empty pixels, as the projection images
    x[:, :, :20, :] = 0                                               # and an empty band: exact ties over whole windows
    target = torch.randint(0, 2, (B,), generator=g)
    ref.train()
    m = m.to(cuda_device).train()
    if B <= 64:
        lref = F.nll_loss(ref(x.double()), target)
    else:
        # a big batch holds a few windows whose two largest values differ by less than fp32 resolves: fp32 and fp64 then
        # pool different positions and the gradient of conv1 moves by ~1e-3 of its maximum (ATen's own fp32 path differs
        # from fp64 by as much).  The derivative of what the kernels COMPUTED pools at the positions they recorded:
        from pointnetgpd_amd import gpd_ops
        with torch.no_grad():
            xd = x.to(cuda_device)
            p1, a1 = gpd_ops.conv5_pool2_arg(xd, m.conv1.weight.detach(), m.conv1.bias.detach())
            _, a2 = gpd_ops.conv5_pool2_arg(p1, m.conv2.weight.detach(), m.conv2.bias.detach())

        def pool_at(c, a):                                            # (B,C,2H,2W) fp64, recorded positions (B,C,H,W) u8
            Bc, C, H2, W2 = c.shape
            win = c.view(Bc, C, H2 // 2, 2, W2 // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(Bc, C, H2 // 2, W2 // 2, 4)
            return win.gather(4, a.long().unsqueeze(-1)).squeeze(-1)
        h = pool_at(ref.conv1(x.double()), a1.cpu())
        assert (h - F.max_pool2d(ref.conv1(x.double()), 2, 2)).abs().max() < 1e-5     # near ties only
        h2 = pool_at(ref.conv2(h), a2.cpu())
        assert (h2 - F.max_pool2d(ref.conv2(h), 2, 2)).abs().max() < 1e-5
        h = ref.fc2(ref.relu(ref.fc1(h2.reshape(-1, 7200))))
        lref = F.nll_loss(F.log_softmax(h, dim=-1), target)
    lref.backward()
    out = m(x.to(cuda_device))
    assert "libpngpd.so" in open("/proc/self/maps").read() and out.grad_fn.name().startswith("GPDNetFn")
    loss = F.nll_loss(out, target.to(cuda_device))
    loss.backward()
    assert abs(loss.item() - lref.item()) < 1e-5
    with torch.no_grad():
        assert torch.equal(m.eval()(x.to(cuda_device)), out.detach())  # the stages of train() are the stages of eval()
    for (name, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        got, want = p.grad.double().cpu(), q.grad
        err = (got - want).abs().max().item()
        assert err <= 2e-5 * want.abs().max().item() + 1e-9, (name, err, want.abs().max().item())
    # deterministic: a second backward reproduces the gradients bit for bit
    first = [p.grad.clone() for p in m.parameters()]
    m.zero_grad(set_to_none=True)
    m.train()
    F.nll_loss(m(x.to(cuda_device)), target.to(cuda_device)).backward()
    assert all(torch.equal(a, p.grad) for a, p in zip(first, m.parameters()))


def test_conv5_pool2_arg_records_first_maximum(cuda_device):
    """The recorded window position reproduces the pooled value, and on exact ties it is the first in row-major order
    (ATen max_pool2d: strict >)."""
    from pointnetgpd_amd import gpd_ops
    g = torch.Generator().manual_seed(1)
    x = torch.rand(3, 4, 28, 28, generator=g)
    x[:, :, 10:, :] = 0.0
    w = torch.randn(7, 4, 5, 5, generator=g) * 0.1
    b = torch.randn(7, generator=g)
    out, arg = gpd_ops.conv5_pool2_arg(x.to(cuda_device), w.to(cuda_device), b.to(cuda_device))
    assert torch.equal(out, gpd_ops.conv5_pool2(x.to(cuda_device), w.to(cuda_device), b.to(cuda_device)))
    conv = torch.nn.functional.conv2d(x.double(), w.double(), b.double())
    pooled, idx = torch.nn.functional.max_pool2d(conv, 2, 2, return_indices=True)
    yy, xx = idx // 24, idx % 24
    code = ((yy % 2) * 2 + (xx % 2)).to(torch.uint8)
    a = arg.cpu()
    assert (a[:, :, 6:, :] == 0).all()                                 # the empty band: ties -> position 0
    # elsewhere the fp32 choice equals the fp64 one except at near-ties, where the chosen VALUE is still the maximum
    chosen = conv.view(3, 7, 12, 2, 12, 2).permute(0, 1, 2, 4, 3, 5).reshape(3, 7, 12, 12, 4).gather(
        4, a.long().unsqueeze(-1)).squeeze(-1)
    assert (chosen - pooled).abs().max() < 1e-5 and (a == code).float().mean() > 0.999


def test_gpd_classifier_cuda_dropout_is_explicit(cuda_device):
    """dropout=True in train() mode has no libpngpd kernel: a CUDA tensor raises like every other CUDA path without one,
    instead of silently dispatching to ATen / MIOpen; the opt-in runs the reference's own composite (gpd.py:5-31)."""
    from pointnetgpd_amd.model.gpd import GPDClassifier
    torch.manual_seed(3)
    m = GPDClassifier(3, dropout=True).to(cuda_device).train()
    x = torch.rand(4, 3, 60, 60, device=cuda_device)
    with pytest.raises(RuntimeError, match="allow_aten_training"):
        m(x)
    try:
        GPDClassifier.allow_aten_training = True
        out = m(x)
        out.sum().backward()
        assert out.shape == (4, 2) and m.conv1.weight.grad is not None
    finally:
        GPDClassifier.allow_aten_training = False
    x.requires_grad_(True)
    with pytest.raises(RuntimeError, match="input images"):
        GPDClassifier(3).to(cuda_device).train()(x).sum().backward()


def test_fc_fwd_splitk_vs_torch(cuda_device):
    """pngpd_fc_fwd_splitk (the classifier's fc1: few output tiles, K = 7200) vs an fp64 composite; ragged B / Nout,
    slices that do not divide K evenly, a contraction too short to split; deterministic."""
    from pointnetgpd_amd import gpd_ops
    g = torch.Generator().manual_seed(8)
    for (B, K, Nout, relu) in [(1, 7200, 500, True), (37, 7200, 500, True), (64, 7200, 500, False), (200, 1000, 33, True),
                               (5, 64, 7, False), (33, 8, 40, True)]:
        a = torch.randn(B, K, generator=g); W = torch.randn(Nout, K, generator=g) / K ** 0.5
        bias = torch.randn(Nout, generator=g)
        ref = a.double() @ W.double().T + bias.double()
        if relu:
            ref = ref.clamp(min=0)
        ag, Wg, bg = a.to(cuda_device), W.to(cuda_device), bias.to(cuda_device)
        out = gpd_ops.fc_fwd_splitk(ag, Wg, bg, relu)
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)
        assert torch.equal(out, gpd_ops.fc_fwd_splitk(ag, Wg, bg, relu))
