"""GPU: the torch-free C++ consumer of the C ABI (examples/cabi_consumer.cpp, built by `make example` / build())
against the same calls made through the ctypes binding AND a plain fp64 numpy evaluation of the same layers."""
import os
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "cabi_consumer")


class Lcg:
    """The generator of examples/cabi_consumer.cpp."""

    def __init__(self, s):
        self.s = s

    def fill(self, n, scale, shift=0.0):
        out = np.empty(n, dtype=np.float32)
        s = self.s
        for i in range(n):
            s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
            out[i] = s >> 8
        self.s = s
        u = out * np.float32(1.0 / 16777216.0) - np.float32(0.5)
        return u * np.float32(scale) + np.float32(shift)


@pytest.mark.parametrize("B,N", [(5, 200), (3, 1000)])
def test_cabi_consumer_matches_binding(B, N, cuda_device):
    from pointnetgpd_amd import ops
    if not os.path.exists(EXE):          # normally built by __graft_entry__.build(); same toolchain on the GPU box
        subprocess.run(["make", "-C", os.path.join(ROOT, "pointnetgpd_amd", "csrc"), "example"], check=True)
    res = subprocess.run([EXE, str(B), str(N)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.strip().splitlines()
    pool_cs = np.array([[float(v) for v in ln.split()[2:]] for ln in lines if ln.startswith("pool")])
    fc = np.array([[float(v) for v in ln.split()[2:]] for ln in lines if ln.startswith("fc")])
    assert pool_cs.shape == (B, 2) and fc.shape == (B, 9)

    g = Lcg(12345)
    x = g.fill(B * 3 * N, 0.1).reshape(B, 3, N)
    T = g.fill(B * 9, 0.2).reshape(B, 3, 3)
    T[:, np.arange(3), np.arange(3)] += np.float32(1.0)
    C = [3, 64, 128, 1024]
    layers = []
    for l in range(3):
        W = g.fill(C[l + 1] * C[l], 2.0 / C[l]).reshape(C[l + 1], C[l])
        layers.append(dict(W=W, b=g.fill(C[l + 1], 0.1), g=g.fill(C[l + 1], 1.0, 1.0), be=g.fill(C[l + 1], 0.2),
                           mu=g.fill(C[l + 1], 0.2), var=g.fill(C[l + 1], 1.0, 1.0)))
    layers[2]["g"][::7] *= -1
    Wfc = g.fill(9 * 1024, 0.05).reshape(9, 1024)
    bfc = g.fill(9, 0.1)

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda_device)
    folded = []
    for l, L in enumerate(layers):
        folded += list(ops.fold_conv_bn(t(L["W"]), t(L["b"]), t(L["g"]), t(L["be"]), t(L["mu"]), t(L["var"]), 1e-5,
                                        ops.LAYOUT_ROWMAJOR if l == 0 else ops.LAYOUT_MFMA_B))
    pool = ops.trunk_fwd_infer(t(x), t(T), *folded, False)
    out = ops.fc_fwd(pool, t(Wfc), t(bfc), ops.EPI_ADD_IDEN3)
    pool_np, out_np = pool.cpu().numpy().astype(np.float64), out.cpu().numpy()
    # same library, same inputs -> same bits; the printed 10 significant digits bound the comparison
    np.testing.assert_allclose(fc, out_np, rtol=2e-9, atol=0)
    np.testing.assert_allclose(pool_cs[:, 0], pool_np.sum(1), rtol=1e-9)
    np.testing.assert_allclose(pool_cs[:, 1], np.abs(pool_np).sum(1), rtol=1e-9)

    # and both agree with a straight fp64 evaluation of conv+BN(eval)+ReLU, max over points, FC + identity
    h = np.einsum("bji,bjn->bin", T.astype(np.float64), x.astype(np.float64))           # x' = x^T T  (pointnet.py:140-143)
    for l, L in enumerate(layers):
        z = np.einsum("ck,bkn->bcn", L["W"].astype(np.float64), h) + L["b"].astype(np.float64)[None, :, None]
        s = L["g"].astype(np.float64) / np.sqrt(L["var"].astype(np.float64) + 1e-5)
        z = (z - L["mu"].astype(np.float64)[None, :, None]) * s[None, :, None] + L["be"].astype(np.float64)[None, :, None]
        h = np.maximum(z, 0) if l < 2 else z
    ref_pool = h.max(2)
    ref_fc = ref_pool @ Wfc.astype(np.float64).T + bfc + np.eye(3).reshape(1, 9)
    np.testing.assert_allclose(pool_np, ref_pool, rtol=0, atol=2e-5)
    np.testing.assert_allclose(fc, ref_fc, rtol=0, atol=2e-4)


@pytest.mark.parametrize("B,N", [(5, 200), (16, 750)])
def test_cabi_consumer_train_step_matches_binding(B, N, cuda_device):
    """The torch-free consumer drives ONE training step of a trunk through the fused per-direction entries
    (pngpd_trunk_train_fwd / _bwd, caller-provided save / scratch) and pngpd_adam_flat; the same calls through the
    ctypes binding on the same inputs give the same numbers, and a too-small scratch buffer is refused."""
    import ctypes
    from pointnetgpd_amd import _lib, ops
    subprocess.run(["make", "-C", os.path.join(ROOT, "pointnetgpd_amd", "csrc"), "example"], check=True,
                   capture_output=True)
    res = subprocess.run([EXE, str(B), str(N), "train"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    got = {ln.split()[1]: (float(ln.split()[2]), float(ln.split()[3]))
           for ln in res.stdout.splitlines() if ln.startswith("train ")}
    assert len(got) == 16, got.keys()

    g = Lcg(12345)
    x = g.fill(B * 3 * N, 0.1).reshape(B, 3, N)
    T = g.fill(B * 9, 0.2).reshape(B, 3, 3)
    T[:, np.arange(3), np.arange(3)] += np.float32(1.0)
    C = [3, 64, 128, 1024]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda_device)
    L = []
    for l in range(3):
        L.append(dict(W=t(g.fill(C[l + 1] * C[l], 2.0 / C[l]).reshape(C[l + 1], C[l])), b=t(g.fill(C[l + 1], 0.1)),
                      g=g.fill(C[l + 1], 1.0, 1.0), be=t(g.fill(C[l + 1], 0.2)), mu=t(g.fill(C[l + 1], 0.2)),
                      var=t(g.fill(C[l + 1], 1.0, 1.0))))
    L[2]["g"][::7] *= -1
    for l in range(3):
        L[l]["g"] = t(L[l]["g"])
    g.fill(9 * 1024, 0.05); g.fill(9, 0.1)                    # the consumer's FC layer draws
    dp = t(g.fill(B * 1024, 2.0).reshape(B, 1024))
    xd, Td = t(x), t(T)
    lib = _lib.load()
    a = _lib.TrunkTrainArgs()
    a.x, a.trans, a.B, a.N, a.S = xd.data_ptr(), Td.data_ptr(), B, N, ops.train_splits(B, N)
    a.need_bwd, a.eps, a.momentum = 1, 1e-5, 0.1
    for i, l in enumerate(L, 1):
        setattr(a, f"w{i}", l["W"].data_ptr()); setattr(a, f"b{i}", l["b"].data_ptr())
        setattr(a, f"g{i}", l["g"].data_ptr()); setattr(a, f"be{i}", l["be"].data_ptr())
        setattr(a, f"rm{i}", l["mu"].data_ptr()); setattr(a, f"rv{i}", l["var"].data_ptr())
    sb, wb = lib.pngpd_trunk_train_save_bytes(ctypes.addressof(a)), lib.pngpd_trunk_train_scratch_bytes(ctypes.addressof(a))
    save = torch.empty(sb, dtype=torch.uint8, device=cuda_device)
    ws = torch.empty(wb, dtype=torch.uint8, device=cuda_device)
    a.save, a.save_bytes, a.scratch, a.scratch_bytes = save.data_ptr(), sb, ws.data_ptr(), wb
    pooled = torch.empty(B, 1024, device=cuda_device); zhat = torch.empty_like(pooled)
    idx = torch.empty(B, 1024, dtype=torch.int32, device=cuda_device)
    a.pooled, a.idx, a.zhat = pooled.data_ptr(), idx.data_ptr(), zhat.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.pngpd_trunk_train_fwd(ctypes.addressof(a), st), "fwd")
    sizes = [192, 64, 64, 64, 8192, 128, 128, 128, 131072, 1024, 1024, 1024]
    names = ["dW1", "db1", "dg1", "dbe1", "dW2", "db2", "dg2", "dbe2", "dW3", "db3", "dg3", "dbe3"]
    grad = torch.empty(sum(sizes), device=cuda_device)
    views = grad.split(sizes)
    dT = torch.empty(B, 3, 3, device=cuda_device)
    a.dp, a.dT = dp.data_ptr(), dT.data_ptr()
    for n, v in zip(names, views):
        setattr(a, n, v.data_ptr())
    _lib.check(lib.pngpd_trunk_train_bwd(ctypes.addressof(a), st), "bwd")
    m, v2 = torch.zeros(1024 * 128, device=cuda_device), torch.zeros(1024 * 128, device=cuda_device)
    _lib.check(lib.pngpd_adam_flat(L[2]["W"].data_ptr(), views[8].data_ptr(), m.data_ptr(), v2.data_ptr(), 1024 * 128,
                                   0.005, None, 0.9, 0.999, 1e-8, 1.0, None, 1.0, None, st), "adam")
    torch.cuda.synchronize()
    mine = dict(pooled=pooled, dT=dT, W3_after_adam=L[2]["W"], running_mean3=L[2]["mu"])
    mine.update(dict(zip(names, views)))
    for n, (s, ab) in got.items():
        h = mine[n].double().cpu()
        assert abs(h.sum().item() - s) <= 1e-9 * max(abs(s), ab * 1e-3, 1e-30) + 1e-12, (n, h.sum().item(), s)
        assert abs(h.abs().sum().item() - ab) <= 1e-9 * max(ab, 1e-30) + 1e-12, (n, h.abs().sum().item(), ab)
    assert got["db1"] == (0.0, 0.0) and got["db3"] == (0.0, 0.0)      # conv biases ahead of a train-mode BN
    assert got["dW3"][1] > 0 and got["dT"][1] > 0


@pytest.mark.gpu
def test_probe_mfma_rate_runs_and_counts_flops(cuda_device):
    """The matrix-rate probe of bench.py's roofline block: launches, reports the FLOPs it executes, and lands in a sane
    range (a bare fp32 MFMA stream cannot beat the nominal peak and should not be far below it)."""
    import ctypes
    import torch
    from pointnetgpd_amd import _lib
    lib = _lib.load()
    cus = torch.cuda.get_device_properties(cuda_device).multi_processor_count
    sink = torch.empty(512 * 512, device=cuda_device)
    stream = torch.cuda.current_stream(cuda_device).cuda_stream
    for dtype, per in ((0, 2 * 32 * 32 * 2), (1, 2 * 32 * 32 * 16)):
        for wps in (1, 2):
            flops = ctypes.c_longlong(0)
            iters = 2000
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for rep in range(2):
                e0.record()
                _lib.check(lib.pngpd_probe_mfma_rate(dtype, wps, iters, sink.data_ptr(), ctypes.addressof(flops), stream), "probe")
                e1.record()
                torch.cuda.synchronize()
            assert flops.value == cus * 4 * wps * iters * 8 * per
            tf = flops.value / (e0.elapsed_time(e1) * 1e-3) / 1e12
            peak = 157.3 if dtype == 0 else 2500.0
            assert 0.5 * peak < tf < 1.02 * peak, (dtype, wps, tf)


def test_cabi_index_consumer_cross_checks(cuda_device):
    """examples/cabi_index_consumer.cpp — the torch-free consumer of the index-heavy entries (crop incl. the indexed and
    one-launch forms, the HBM-resident training batch, the GPG sampler's per-pose AND per-unit kernels, GPD) on edge-case
    shapes.  It cross-checks inside: indexed crop == plain crop, device-side collate == the keep rule, indexed sweep ==
    brute force, fused sweep / push-in == the per-pose path (three margins, packed result included), indexed moments ==
    whole-cloud moments — and exits non-zero on any mismatch."""
    import subprocess
    exe = os.path.join(ROOT, "examples", "cabi_index_consumer")
    assert os.path.exists(exe), "run `make -C pointnetgpd_amd/csrc example` (or __graft_entry__.build())"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    for tag in ("crop_indexed_f32", "crop_indexed_f64", "train_batch one-view", "train_batch full-view", "gpg_chain", "gpg_fused",
                "gpd_projection"):
        assert tag in out.stdout, (tag, out.stdout[-1500:])
    line = [l for l in out.stdout.splitlines() if l.startswith("gpg_chain")][0]
    assert "indexed_vs_bruteforce_mismatches 0" in line and int(line.split("potential")[1].split()[0]) > 0
    # ABI v7: pngpd_gpg_frames (np.linalg.eig as LAPACK evaluates it, on the device) == the host build of the same header
    assert "gpg_frames == host build of pngpd_gpg_eig3.h: yes" in out.stdout, out.stdout[-1500:]
