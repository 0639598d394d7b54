"""GPU: the plain-bf16 mode (BASELINE.json configs[2]: "3-class ... bf16") — bf16 cloud storage (6 B/point), single-
product v_mfma_f32_32x32x16_bf16 with fp32 accumulation for the two GEMM layers of the trunk, fp32 BatchNorm
statistics.  SURVEY.md §7: bf16 operands do NOT meet the 1e-3 log-prob contract in general, so this file MEASURES and
BOUNDS the error against the fp32/fp64 oracle (printed with -s, summarised in DESIGN.md §2.1c) instead of asserting
1e-3: max |d log-prob|, arg-max agreement (overall and where the oracle's margin exceeds the bf16 error), and for the
training step the loss and per-tensor gradient errors."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pointnet_oracle as po
from tests.helpers import build_model, state_dict_cpu, synth_cloud, oracle_train_step_on_device

pytestmark = pytest.mark.gpu


@pytest.fixture()
def bf16_modes():
    from pointnetgpd_amd.model import pointnet as pn
    from pointnetgpd_amd import train
    yield pn, train
    pn.set_inference_precision("fp32")
    train.set_train_precision("fp32")


@pytest.mark.parametrize("B,N,k,kind", [(512, 1024, 3, "box"), (64, 750, 2, "box"), (7, 129, 3, "gauss")])
def test_bf16_inference_error(B, N, k, kind, bf16_modes, cuda_device):
    pn, _ = bf16_modes
    m = build_model(N, k, 610 + B, 6100 + B).eval()
    sd = state_dict_cpu(m)
    x = synth_cloud(B, N, 2500 + B, kind)
    with torch.no_grad():
        lp_ref, tr_ref = po.forward_torch(sd, x)
    mg = m.to(cuda_device)
    xg = x.to(cuda_device)
    with torch.no_grad():
        lp32, _ = mg(xg)
        pn.set_inference_precision("bf16")
        lp_b, tr_b = mg(xg)                                   # bf16 arithmetic, fp32 cloud storage
        xb = xg.to(torch.bfloat16)
        lp_bs, tr_bs = mg(xb)                                 # bf16 arithmetic AND bf16 cloud storage
        pn.set_inference_precision("fp32")
        lp_st, _ = mg(xb)                                     # fp32 arithmetic on the bf16-stored cloud
    assert "libpngpd.so" in open("/proc/self/maps").read()
    ref = lp_ref
    margin = ref.max(1)[0] - ref.kthvalue(ref.shape[1] - 1, 1)[0]
    rows = {}
    for name, lp in (("fp32", lp32), ("bf16 arithmetic", lp_b), ("bf16 arithmetic + storage", lp_bs),
                     ("fp32 arithmetic, bf16 storage", lp_st)):
        d = (lp.cpu() - ref).abs().max().item()
        agree = (lp.argmax(1).cpu() == ref.argmax(1)).float().mean().item()
        rows[name] = (d, agree)
        print(f"[B={B} N={N} k={k} {kind}] {name:32s} max|dlogp| {d:.3e}  argmax agreement {agree:.4f}")
    assert rows["fp32"][0] < 1e-3
    # measured round 2: bf16 arithmetic 2e-3 .. 3e-2, storage alone up to ~5e-2 on these clouds
    assert rows["bf16 arithmetic"][0] < 0.15 and rows["bf16 arithmetic"][1] >= 0.95
    assert rows["bf16 arithmetic + storage"][0] < 0.3 and rows["bf16 arithmetic + storage"][1] >= 0.9
    # where the oracle's decision margin exceeds twice the measured error, the decision is identical
    d = rows["bf16 arithmetic"][0]
    clear = margin > 2 * d
    assert (lp_b.argmax(1).cpu()[clear] == ref.argmax(1)[clear]).all()
    assert torch.isfinite(lp_bs).all() and (tr_bs.cpu() - tr_ref).abs().max().item() < 0.3


@pytest.mark.parametrize("kind", ["diverse", "box"])
def test_bf16_train_step_error(kind, bf16_modes, cuda_device):
    """BASELINE configs[2] per-GPU share: k = 3, B = 512, N = 1024; pass C in bf16, bf16 cloud storage.
    "box" = the bench's iid box clouds: the pooled features are nearly identical across the batch, so the FC
    BatchNorms (batch statistics) divide by a vanishing variance and amplify any arithmetic noise — reported, only
    finiteness asserted.  "diverse" = clouds that differ from each other, as crops of real scenes do: bounded."""
    pn, train = bf16_modes
    B, N, k = 512, 1024, 3
    m = build_model(N, k, 77, 6177).train()
    sd = state_dict_cpu(m)
    x = synth_cloud(B, N, 2677, kind)
    y = (torch.arange(B) * 7 % k).long()
    loss_ref, logp_ref, trans_ref, g64, stats_ref = oracle_train_step_on_device(sd, x, y, torch.float64, cuda_device)
    _, logp32, _, g32, _ = oracle_train_step_on_device(sd, x, y, torch.float32, cuda_device)
    res = {}
    names = [n for n in g64 if g64[n].double().norm().item() >= 1e-9]

    def summarise(tag, loss, logp, grads):
        rel = {n: (grads[n].double() - g64[n].double()).norm().item() / g64[n].double().norm().item() for n in names}
        worst = max(rel, key=rel.get)
        big = {n: r for n, r in rel.items() if n.endswith("conv3.weight") or n.endswith("fc1.weight")}
        res[tag] = (abs(loss - loss_ref.item()), (logp - logp_ref).abs().max().item(), rel[worst],
                    float(np.median(list(rel.values()))),
                    (logp.argmax(1) == logp_ref.argmax(1)).float().mean().item())
        print(f"[train {kind} B={B} N={N} k={k}] {tag:22s} |dloss| {res[tag][0]:.2e}  max|dlogp| {res[tag][1]:.2e}  argmax "
              f"{res[tag][4]:.4f}  grad rel err: median {res[tag][3]:.2e}, worst {worst} {rel[worst]:.2e}, "
              + ", ".join(f"{n} {r:.2e}" for n, r in sorted(big.items())))

    summarise("ATen fp32 (yardstick)", F.nll_loss(logp32, y).item(), logp32, g32)
    for mode, xin in (("fp32", x), ("bf16x3", x), ("bf16x3 side-fp32", x), ("bf16", x), ("bf16 side-fp32", x),
                      ("bf16+storage", x.to(torch.bfloat16))):
        mm = build_model(N, k, 77, 6177).train().to(cuda_device)
        train.set_train_precision(mode.split(" ")[0].split("+")[0], fp32_side_passes=mode.endswith("side-fp32"))
        try:
            logp, _ = mm(xin.to(cuda_device))
            loss = F.nll_loss(logp, y.to(cuda_device))
            loss.backward()
        finally:
            train.set_train_precision("fp32")
        assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in mm.parameters())
        summarise(mode, loss.item(), logp.detach().cpu(), {n: p.grad.cpu() for n, p in mm.named_parameters()})
    assert res["fp32"][0] < 1e-3 and res["fp32"][1] < 1e-3            # the exact mode: 1e-3 on any input
    if kind == "diverse":
        assert res["bf16x3"][0] < 1e-3 and res["bf16x3"][1] < 1e-3    # the exact-enough mode: 1e-3
        # plain bf16: bounded, NOT 1e-3 (bounds = measured round-2 values x ~3)
        assert res["bf16"][0] < 0.02 and res["bf16"][1] < 0.1 and res["bf16"][4] >= 0.97
        assert res["bf16"][3] < 0.4 and res["bf16+storage"][3] < 0.4 and res["bf16 side-fp32"][3] < 0.35
        # the side passes on bf16x3 operands cost nothing measurable: same gradient error as with fp32 side passes
        assert res["bf16x3"][3] < max(2 * res["bf16x3 side-fp32"][3], 2e-2)


@pytest.mark.parametrize("nt,tol", [(3, 1e-4), (1, 2e-2)])
def test_bf_side_passes_match_fp32_passes(nt, tol, bf16_modes, cuda_device, monkeypatch, pass_sequencing):
    """Passes B / gather / D / E with their contractions on bf16 (nt = 1) / bf16x3 (nt = 3) operands — the NT variants
    of the fp32 kernels — against the fp32 passes on IDENTICAL inputs: the arguments of every side-pass launch of a
    real fp32 training step are recorded and replayed through the _bf entry points, output buffer by output buffer
    (a fragment-layout mistake would show as an O(1) difference; the measured differences are printed)."""
    from pointnetgpd_amd import ops
    pn, train = bf16_modes
    B, N, k = 48, 200, 2          # 4 tiles per cloud, the last one ragged (200 = 3*64 + 8)
    x = synth_cloud(B, N, 4242, "diverse").to(cuda_device)
    y = (torch.arange(B) * 5 % k).long().to(cuda_device)
    mm = build_model(N, k, 31, 7001).train().to(cuda_device)
    rec = {}
    for name in ("trunk_bn2_stats", "trunk_bwd_gather", "trunk_bwd_d", "trunk_bwd_e"):
        def wrap(*a, _o=getattr(ops, name), _n=name, **kw):
            out = _o(*a, **kw)
            rec.setdefault(_n, []).append((a, kw, out))
            return out
        monkeypatch.setattr(ops, name, wrap)
    logp, _ = mm(x)
    F.nll_loss(logp, y).backward()
    monkeypatch.undo()
    assert all(len(rec[n]) == 2 for n in rec)            # STN trunk and feat trunk
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    worst = ("", 0.0)

    def check(tag, got, ref):
        nonlocal worst
        for i, (g, r) in enumerate(zip(got, ref)):
            if g.dtype == torch.int16:          # plain bf16: the z2 / g2 tiles are stored as bf16 (half the bytes)
                assert g.numel() * 2 == r.numel() * 4 // 2
                g = ops.tiles_bf16_to_f32(g)
            e = rel(g, r)
            if e > worst[1]:
                worst = (f"{tag}[{i}]", e)
            assert e < tol, (tag, i, e)

    # forward order: STN trunk first; backward order: feat trunk first
    for trunk, fi, bi in (("stn", 0, 1), ("feat", 1, 0)):
        conv2 = mm.feat.stn.conv2 if trunk == "stn" else mm.feat.conv2
        w2 = conv2.weight.detach().reshape(128, 64).contiguous()
        w2x, w2tx = ops.split_pack_bf16(w2), ops.split_pack_bf16(w2.t().contiguous())
        (a, kw, out) = rec["trunk_bn2_stats"][fi]
        xx, T, w1, b1c, s1c, t1c, w2p, S = a
        part, z2t = ops.trunk_bn2_stats_bf(xx, T, w1, b1c, s1c, t1c, w2x, S, nt, store_z2=True)
        check(f"{trunk}.B(part,z2t)", (part, z2t), out)
        (a, kw, out) = rec["trunk_bwd_gather"][bi]
        xx, T, w1, b1c, s1c, t1c, w2p, s2c, t2c, idx, coef = a
        check(f"{trunk}.gather(Gp)", (ops.trunk_bwd_gather_bf(xx, T, w1, b1c, s1c, t1c, w2x, s2c, t2c, idx, coef, nt),),
              (out,))
        (a, kw, out) = rec["trunk_bwd_d"][bi]
        xx, T, w1, b1c, s1c, t1c, w2p, s2c, t2c, is2, nm2, Ap, cvec, w3, idx, coef, S, z2 = a
        Ax = ops.split_pack_bf16(ops.unpack_mfma_b_128(Ap).contiguous())
        # identical inputs on both sides.  bf16x3: the recorded fp32 tiles go to both.  Plain bf16 stores its tiles as
        # bf16, so there the bf16 pass B's tiles are fed to the bf16 passes and, widened back to fp32 tiles, to the
        # fp32 passes (otherwise a ReLU mask derived from a rounded z2 near zero would differ, not the arithmetic)
        if nt == 1:
            z_bf, z_ref = z2t, ops.tiles_bf16_to_f32(z2t)
            out = ops.trunk_bwd_d(xx, T, w1, b1c, s1c, t1c, w2p, s2c, t2c, is2, nm2, Ap, cvec, w3, idx, coef, S, z_ref)
        else:
            z_bf = z_ref = z2
        d_out = ops.trunk_bwd_d_bf(xx, s2c, t2c, is2, nm2, Ax, cvec, w3, idx, coef, S, z_bf, nt)
        check(f"{trunk}.D(g2t,pa,ps2)", d_out, out)
        (a, kw, out) = rec["trunk_bwd_e"][bi]
        xx, T, w1, b1c, s1c, t1c, w2p, is1, nm1, is2, nm2, a1m, a2m, dsc2, w2tp, g2t, S, z2 = a
        if nt == 1:
            g_bf, g_ref = d_out[0], ops.tiles_bf16_to_f32(d_out[0])
            out = ops.trunk_bwd_e(xx, T, w1, b1c, s1c, t1c, w2p, is1, nm1, is2, nm2, a1m, a2m, dsc2, w2tp, g_ref, S, z_ref)
        else:
            g_bf = g2t
        check(f"{trunk}.E(pc,pR,pW2)", ops.trunk_bwd_e_bf(xx, T, w1, b1c, s1c, t1c, is1, nm1, is2, nm2, a1m, a2m, dsc2,
                                                           w2tx, g_bf, S, z_bf, nt), out)
    print(f"[nterms={nt}] side passes on bf16 operands vs the fp32 passes on the same inputs: worst output "
          f"{worst[0]} rel {worst[1]:.2e} (bound {tol})")


@pytest.mark.parametrize("nt", [3, 1])
@pytest.mark.parametrize("B,N", [(6, 150), (4, 750), (3, 64)])
def test_bf_pass_c_reads_z2_back(nt, B, N, bf16_modes, cuda_device, monkeypatch, pass_sequencing):
    """Pass C on bf16 operands with z2 read back from pass B's store (LOADZ) against the same kernel recomputing
    layers 1-2: same maxima, arg-maxima and sums.  N = 150: an odd number of 64-point tiles (the second half of the
    last 128-point tile does not exist in z2t); N = 64: a single half tile."""
    from pointnetgpd_amd import ops
    pn, train = bf16_modes
    x = synth_cloud(B, N, 515 + N, "diverse").to(cuda_device)
    y = (torch.arange(B) % 2).long().to(cuda_device)
    mm = build_model(N, 2, 33, 7003).train().to(cuda_device)
    rec = []

    def wrap(*a, _o=ops.trunk_fwd_train_bf, **kw):
        out = _o(*a, **kw)
        rec.append((a, kw, out))
        return out

    monkeypatch.setattr(ops, "trunk_fwd_train_bf", wrap)
    train.set_train_precision("bf16x3" if nt == 3 else "bf16")
    try:
        logp, _ = mm(x)
        F.nll_loss(logp, y).backward()          # a backward follows: pass B stores z2 in both modes
    finally:
        train.set_train_precision("fp32")
    monkeypatch.undo()
    assert len(rec) == 2
    for a, kw, out in rec:
        xx, T, w1, b1c, s1c, t1c, w2x, s2c, t2c, w3sx, S = a
        part, z2t = ops.trunk_bn2_stats_bf(xx, T, w1, b1c, s1c, t1c, w2x, ops.train_splits(B, N), nt, store_z2=True)
        got = ops.trunk_fwd_train_bf(xx, T, w1, b1c, s1c, t1c, w2x, s2c, t2c, w3sx, S, nterms=nt, z2t=z2t)
        ref = ops.trunk_fwd_train_bf(xx, T, w1, b1c, s1c, t1c, w2x, s2c, t2c, w3sx, S, nterms=nt, z2t=None)
        # bf16x3: pass B and pass C accumulate layer 2 in the same order (~bit-equal); plain bf16: the stored z2 is
        # rounded to bf16 (2^-9), which moves near-tied arg-maxima
        tol, flips, stol = (1e-5, 2e-3, 1e-4) if nt == 3 else (2e-2, 0.15, 2e-2)
        assert (got[0] - ref[0]).abs().max().item() <= tol * ref[0].abs().max().item()
        assert (got[1] != ref[1]).float().mean().item() < flips
        for g, r in zip(got[2:4], ref[2:4]):
            assert ((g - r).norm() / r.norm()).item() < stol
