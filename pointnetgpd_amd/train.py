"""Train-mode forward/backward of the hot path as ``torch.autograd.Function``s over libpngpd
(include/pngpd.h "Training path").

Two ways of sequencing the same kernels, bit-identical in their results:

* ``"fused"`` (default): one C-ABI call per direction of each of the four pieces of the graph — the two trunks
  (``pngpd_trunk_train_fwd/_bwd``) and the two FC stacks (``pngpd_head_train_fwd/_bwd``); a training step is eight
  foreign calls, so the reference's own recipe (batch 64, main_1v.py:18-33) is no longer bound by host sequencing.
* ``"passes"``: every pass / finalize kernel called one by one from Python (``TrunkTrainFn`` / ``LinearBnReluFn`` /
  ``LinearEpiFn`` below) — the executable specification of the fused entries, and the form the kernel-level tests
  hook into (``DEBUG_STASH``, recorded launches).

Everything batch-sized AND the parameter-sized fp64 algebra between the passes (BatchNorm statistics ->
affine forms, running-stat updates, reduction of per-workgroup partials, closed-form weight gradients) runs
in HIP kernels reached through the C ABI; this module only sequences them on torch's current stream and owns
the buffers (``torch.empty``).  The remaining torch ops are layout plumbing (transposes / zero padding for the
FC backward GEMMs, ``.view``).

The algebra is derived in DESIGN.md ("Training passes") and verified against autograd in fp64 by
``tests/train_algo_prototype.py``.  Semantics match the reference's train-mode graph
(pointnet.py:27-45,137-154,189-194 under main_1v.py:72-76): batch-statistics BatchNorm with eps 1e-5,
running statistics updated with momentum 0.1 and the unbiased variance, num_batches_tracked += 1.
"""
import ctypes

import torch

from . import _lib, arith, ops
from .ops import _call

F64 = torch.float64
# Arithmetic of the trunk's matrix contractions (per model, arith.py): "fp32" exact (default); "bf16x3" opt-in 3-term
# split-bf16 products (same parity bars as fp32); "bf16" opt-in plain bf16 operands (BASELINE configs[2]; errors measured
# in tests/test_gpu_bf16.py).  In the two opt-in modes EVERY pass contracts on the bf16 matrix cores — pass C
# (trunk_fwd_train_x3_kernel) and the side passes B / gather / D / E.  ``fp32_side_passes=True`` keeps B / gather / D / E
# on the exact fp32 kernels (round 2's first bf16 mode).  BatchNorm statistics, masks, every accumulator and the
# parameter-sized algebra between the passes stay fp32 / fp64 in all modes.
# ``refine_pool`` (reduced-precision modes only): re-evaluate the pooled maxima in exact fp32 at the arg-max points the
# bf16 / bf16x3 pass C chose (pngpd_trunk_pool_refine).  0 off (round 3's behaviour), 1 on the fp32 matrix pipe, 2 on the
# VALU.  The pooled values are single numbers that carried the full bf16 product error into the FC stacks'
# batch-statistics BatchNorms; after the refinement the matrix pass contributes only the CHOICE of the point.
_NTERMS = {"bf16x3": 3, "bf16": 1}
_REFINE_VALU_VARIANT = 1      # == PNGPD_REFINE_VALU_VARIANT (pngpd_internal.h)


def set_train_precision(mode, fp32_side_passes=False, refine_pool=None):
    """Deprecated shim (rounds 1-5): edits the PROCESS DEFAULT; ``model.set_precision(...)`` is per model."""
    arith.set_default(train=mode, fp32_side_passes=bool(fp32_side_passes), refine_pool=refine_pool)


def set_sequencing(mode):
    """Deprecated shim: process default of "fused" (one C-ABI call per trunk / head direction) | "passes" (pass by pass
    from Python); per model: ``model.set_precision(sequencing=...)``."""
    arith.set_default(sequencing=mode)


DEBUG_STASH = None   # set to a dict to capture backward intermediates (tests/test_gpu_train.py::test_trunk_backward_intermediates)


_PM_ONE = {}


def _pm_one(dev):
    t = _PM_ONE.get(dev)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            # never cache tensors that would live in a capturing graph's private pool
            return torch.tensor(1.0, device=dev), torch.tensor(-1.0, device=dev)
        t = _PM_ONE[dev] = (torch.tensor(1.0, device=dev), torch.tensor(-1.0, device=dev))
    return t


def _e(dev, *shape, dtype=torch.float32):
    return torch.empty(*shape, device=dev, dtype=dtype)


def _bufs3(bufs):
    return bufs if bufs is not None else (None, None, None)


def _bump(bufs):
    """The finalize kernels update running_mean / running_var / num_batches_tracked in place behind
    torch's back: bump their version counters so version-keyed caches (the eval-mode fold cache) see it."""
    if bufs is not None:
        for t in bufs:
            if t is not None:
                torch.autograd.graph.increment_version(t)


def _reduce(t, outer, R, n):
    """(outer,R,n) fp32 partials -> (outer,n) fp64, deterministic."""
    out = _e(t.device, outer, n, dtype=F64)
    _call("pngpd_reduce_partials", t, t, int(outer), int(R), int(n), out)
    return out


class TrunkTrainFn(torch.autograd.Function):
    """x (B,3,N) [, trans (B,3,3)] -> pooled (B,1024) through conv1/bn1/relu, conv2/bn2/relu,
    conv3/bn3[/relu], max over N — batch-statistics BatchNorm.  Gradients for the 12 parameters
    and for ``trans``."""

    @staticmethod
    def forward(ctx, x, trans, W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, g3, be3, relu_last, eps,
                momentum, bufs1, bufs2, bufs3, cfg):
        B, _, N = x.shape
        precision, fp32_side, refine_pool = cfg.train, cfg.fp32_side_passes, cfg.refine_pool
        dev = x.device
        x = x.contiguous()
        T = trans.detach().contiguous() if trans is not None else None
        c = lambda t: t.detach().contiguous()
        w1, w2, w3 = c(W1).reshape(64, 3), c(W2).reshape(128, 64), c(W3).reshape(1024, 128)
        b1c, g1c, be1c = c(b1), c(g1), c(be1)
        b2c, g2c, be2c = c(b2), c(g2), c(be2)
        b3c, g3c, be3c = c(b3), c(g3), c(be3)
        S = ops.train_splits(B, N)
        blk = B * S
        # ---- pass A + BN1 (closed form from per-cloud moments)
        mom = ops.cloud_moments(x)
        chan1, stats1 = _e(dev, 4, 64), _e(dev, 140, dtype=F64)
        rm, rv, nbt = _bufs3(bufs1)
        _call("pngpd_bn1_finalize", x, mom, T, B, N, w1, b1c, g1c, be1c, float(eps), float(momentum), rm, rv, nbt,
              chan1, stats1)
        _bump(bufs1)
        s1c, t1c, is1, nm1 = chan1[0], chan1[1], chan1[2], chan1[3]
        # ---- pass B + BN2
        w2p = ops.pack_mfma_b(w2)
        # z2 = W2 h1 is computed once, here, and handed to passes C / D / E (512 B per point; 256 B in plain-bf16 mode)
        nt = 0 if precision == "fp32" else _NTERMS[precision]
        nt_side = 0 if fp32_side else nt      # arithmetic of passes B / gather / D / E
        w2x = ops.split_pack_bf16(w2) if nt else None
        if nt_side:
            # z2 is computed once, here, and read back by passes C / D / E: fp32 tiles in bf16x3 mode, bf16 tiles (half
            # the bytes) in plain-bf16 mode.  Measured at B=N=1024: reading back instead of recomputing layers 1-2 in
            # pass C is worth 4.80 -> 4.77 ms (bf16x3) and 3.12 -> 3.06 ms (bf16, bf16 tiles; with fp32 tiles it LOST,
            # 3.60 -> 3.69 ms)
            part, z2t = ops.trunk_bn2_stats_bf(x, T, w1, b1c, s1c, t1c, w2x, S, nt_side, store_z2=True)
        else:
            part, z2t = ops.trunk_bn2_stats(x, T, w1, b1c, s1c, t1c, w2p, S,
                                            store_z2=nt == 0 or any(ctx.needs_input_grad))   # fp32 pass C always reads z2 back
        chan2, stats2 = _e(dev, 4, 128), _e(dev, 256, dtype=F64)
        rm, rv, nbt = _bufs3(bufs2)
        tot2 = _reduce(part, 1, blk, 256)
        _call("pngpd_bn2_finalize", x, tot2, B, N, b2c, g2c, be2c, float(eps), float(momentum), rm, rv, nbt,
              chan2, stats2)
        _bump(bufs2)
        s2c, t2c, is2, nm2 = chan2[0], chan2[1], chan2[2], chan2[3]
        # ---- pass C + BN3 + pool
        sgn = torch.where(g3c >= 0, *_pm_one(dev))        # cached 0-dim +1 / -1: no per-step scalar fills
        Sc = S
        if precision != "fp32":
            w3s = (w3 * sgn[:, None]).contiguous()
            pmax, parg, psum, psh, Sc = ops.trunk_fwd_train_bf(x, T, w1, b1c, s1c, t1c, w2x, s2c,
                                                               t2c, ops.split_pack_bf16(w3s), S, nterms=nt,
                                                               z2t=z2t if nt_side == nt else None)
        else:
            w3sp = ops.pack_mfma_b(w3, scale=sgn)
            pmax, parg, psum, psh = ops.trunk_fwd_train(x, T, w1, b1c, s1c, t1c, w2p, s2c, t2c, w3sp, S, z2t)
        stats3 = _e(dev, 2048, dtype=F64)
        rm, rv, nbt = _bufs3(bufs3)
        # the pass's two partial buffers in one launch: sum / sum of squares of z3s, and the column sums of h2
        tot3, sh = ops.reduce4((psum, 1, B * Sc, 2048), (psh, 1, psh.shape[0], 128))
        _call("pngpd_bn3_finalize", x, tot3, B, N, b3c, g3c, float(momentum), rm, rv, nbt, stats3)
        _bump(bufs3)
        pooled, idx, zhat = _e(dev, B, 1024), _e(dev, B, 1024, dtype=torch.int32), _e(dev, B, 1024)
        _call("pngpd_pool_finalize", x, pmax, parg, B, Sc, stats3, g3c, be3c, float(eps), int(relu_last), pooled,
              idx, zhat)
        if nt and refine_pool:
            # the reduced-precision pass chose the points; their values are re-evaluated in exact fp32
            if refine_pool == 1:
                zex = ops.trunk_pool_refine(x, T, w1, b1c, s1c, t1c, w2p, s2c, t2c, idx,
                                            w3sp=ops.pack_mfma_b(w3, scale=sgn), variant=0)
            else:
                zex = ops.trunk_pool_refine(x, T, w1, b1c, s1c, t1c, w2p, s2c, t2c, idx, w3=w3, g3=g3c,
                                            w3sp=ops.pack_mfma_b(w3, scale=sgn), variant=_REFINE_VALU_VARIANT)
            _call("pngpd_pool_finalize", x, zex, idx, B, 1, stats3, g3c, be3c, float(eps), int(relu_last), pooled,
                  idx, zhat)
        ctx.relu_last, ctx.eps, ctx.has_t, ctx.S = relu_last, eps, T is not None, S
        ctx.z2t = z2t          # a plain workspace buffer, not part of the autograd graph
        ctx.nt_side, ctx.w2x = nt_side, w2x
        ctx.save_for_backward(x, T if T is not None else x.new_empty(0), w1, b1c, g1c, w2, g2c, w3, g3c, mom,
                              chan1, stats1, chan2, stats2, stats3, pooled, idx, zhat, w2p, sh)
        return pooled

    @staticmethod
    def backward(ctx, dp):
        (x, T, w1, b1c, g1c, w2, g2c, w3, g3c, mom, chan1, stats1, chan2, stats2, stats3, pooled, idx, zhat,
         w2p, sh) = ctx.saved_tensors
        T = T if ctx.has_t else None
        eps, S = float(ctx.eps), ctx.S
        B, _, N = x.shape
        dev = x.device
        blk = B * S
        s1c, t1c, is1, nm1 = chan1[0], chan1[1], chan1[2], chan1[3]
        s2c, t2c, is2, nm2 = chan2[0], chan2[1], chan2[2], chan2[3]
        # ---- BN3 affine grads, sparse-term weights, dense-correction scalars, pass-D operands
        dp = dp.contiguous()
        coef, dg3, dbe3, m12 = _e(dev, B, 1024), _e(dev, 1024), _e(dev, 1024), _e(dev, 2048, dtype=F64)
        _call("pngpd_bn3_bwd_prep", x, dp, pooled, zhat, B, N, g3c, stats3, eps, int(ctx.relu_last), coef, dg3,
              dbe3, m12)
        Ap, cvec = _e(dev, 128 * 128), _e(dev, 128)
        _call("pngpd_a_cvec_finalize", x, sh, B, N, w3, g3c, stats3, m12, eps, Ap, cvec)
        # ---- arg-extremum gather (sparse term of dW3) and pass D (g2, its BN2 sums, and the Gram of h2)
        z2t, nt = ctx.z2t, ctx.nt_side
        if nt:
            Gp = ops.trunk_bwd_gather_bf(x, T, w1, b1c, s1c, t1c, ctx.w2x, s2c, t2c, idx, coef, nt)
            Ax = ops.split_pack_bf16(ops.unpack_mfma_b_128(Ap).contiguous())
            g2t, pa, ps2 = ops.trunk_bwd_d_bf(x, s2c, t2c, is2, nm2, Ax, cvec, w3, idx, coef, S, z2t, nt)
        else:
            Gp = ops.trunk_bwd_gather(x, T, w1, b1c, s1c, t1c, w2p, s2c, t2c, idx, coef)
            g2t, pa, ps2 = ops.trunk_bwd_d(x, T, w1, b1c, s1c, t1c, w2p, s2c, t2c, is2, nm2, Ap, cvec, w3, idx, coef,
                                           S, z2t)
        G, a12, S2c = ops.reduce4((Gp, 1, Gp.shape[0], 1024 * 128), (pa, 1, blk, 256), (ps2, 1, blk, 12 * 1024))
        dW3 = _e(dev, 1024, 128)
        _call("pngpd_dw3_finalize", x, G, S2c, sh, B, N, w3, g3c, stats3, m12, eps, dW3)
        dg2, dbe2, evec = _e(dev, 128), _e(dev, 128), _e(dev, 3, 128)
        _call("pngpd_bwd_e_prep", x, a12, B, N, g2c, stats2, eps, dg2, dbe2, evec)
        # ---- pass E (also contracts dW2 = sum_points dz2 h1^T on the MFMA)
        if nt:
            w2tx = ops.split_pack_bf16(w2.t().contiguous())
            pc, pR, pW2 = ops.trunk_bwd_e_bf(x, T, w1, b1c, s1c, t1c, is1, nm1, is2, nm2, evec[0], evec[1], evec[2],
                                             w2tx, g2t, S, z2t, nt)
        else:
            w2tp = ops.pack_mfma_b(w2.t().contiguous())
            pc, pR, pW2 = ops.trunk_bwd_e(x, T, w1, b1c, s1c, t1c, w2p, is1, nm1, is2, nm2, evec[0], evec[1], evec[2],
                                          w2tp, g2t, S, z2t)
        dW2_64, c12, Rb = ops.reduce4((pW2, 1, blk, 128 * 64), (pc, 1, blk, 128), (pR, B, S, 192))
        dW2 = dW2_64[0].to(torch.float32)
        dW1, dg1, dbe1 = _e(dev, 64, 3), _e(dev, 64), _e(dev, 64)
        dT = _e(dev, B, 3, 3) if (T is not None and ctx.needs_input_grad[1]) else None
        _call("pngpd_dw1_finalize", x, Rb, T, mom, B, N, c12, stats1, w1, b1c, g1c, eps, dW1, dg1, dbe1, dT)
        if DEBUG_STASH is not None:
            from .ops import g2t_to_rows
            S2full = _s2_full(S2c[0], dev)
            a12v, c12v = a12[0], c12[0]
            DEBUG_STASH.update(dict(dp=dp.to(F64), dg3=dg3, dbe3=dbe3, S2=S2full, sh=sh[0],
                                    G=G[0].view(1024, 128), cvec=cvec, a1=a12v.view(128, 2)[:, 0],
                                    a2=a12v.view(128, 2)[:, 1], c1=c12v.view(64, 2)[:, 0],
                                    c2=c12v.view(64, 2)[:, 1], Rb=Rb.view(B, 64, 3), dW1=dW1, dW2=dW2, dW3=dW3, dT=dT,
                                    g2buf=g2t_to_rows(ops.tiles_bf16_to_f32(g2t) if g2t.dtype == torch.int16 else g2t,
                                                      B, N), idx=idx, coef=coef,
                                    A=Ap.view(4, 16, 2, 32, 4).permute(0, 3, 1, 2, 4).reshape(128, 128)))
        z = lambda n: torch.zeros(n, device=dev, dtype=torch.float32)   # conv bias ahead of train-mode BN: exactly 0
        return (None, dT,
                dW1.view(64, 3, 1), z(64), dg1, dbe1,
                dW2.view(128, 64, 1), z(128), dg2, dbe2,
                dW3.view(1024, 128, 1), z(1024), dg3, dbe3,
                None, None, None, None, None, None, None)


def _s2_full(S2c, dev):
    """The 12 accumulator blocks pass D keeps -> the full symmetric 128x128 second-moment matrix (debug/tests)."""
    blocks = S2c.view(4, 3, 16, 2, 32)                          # [wave][q][r][h][j]
    r = torch.arange(16, device=dev)
    out = torch.zeros(128, 128, device=dev, dtype=S2c.dtype)
    for w in range(4):
        for q in range(3 if w < 2 else 2):
            a, b = w, (w + q) % 4
            blk = torch.empty(32, 32, device=dev, dtype=S2c.dtype)
            for h in range(2):
                rows = (r & 3) + 8 * (r >> 2) + 4 * h
                blk[rows] = blocks[w, q, :, h, :]
                if q == 2:      # block (w, w+2): wave w+2 holds the partial over the other half of the points
                    blk[rows] += blocks[w + 2, 2, :, h, :]
            out[a * 32:(a + 1) * 32, b * 32:(b + 1) * 32] = blk
            out[b * 32:(b + 1) * 32, a * 32:(a + 1) * 32] = blk.t()
    return out


def _linear_bwd(g, inp, W):
    """g (B,Nout) -> (dinp (B,K), dW (Nout,K), db (Nout)): pngpd_fc_bwd, operands read in place."""
    return ops.fc_bwd(g.contiguous(), inp, W)


class LinearBnReluFn(torch.autograd.Function):
    """relu(BatchNorm1d_train(inp @ W^T + b))  (pointnet.py:35-36,191-192)."""

    @staticmethod
    def forward(ctx, inp, W, b, gamma, beta, eps, momentum, bufs):
        inp = inp.contiguous()
        Wd, bd = W.detach().contiguous(), b.detach().contiguous()
        z = ops.fc_fwd(inp, Wd, bd, ops.EPI_NONE)
        y, mean, var = ops.bn1d_fwd_train(z, gamma.detach().contiguous(), beta.detach().contiguous(), eps, 1,
                                          momentum, bufs)
        _bump(bufs)
        ctx.eps = eps
        ctx.save_for_backward(inp, Wd, gamma.detach().contiguous(), z, y, mean, var)
        return y

    @staticmethod
    def backward(ctx, dy):
        inp, W, gamma, z, y, mean, var = ctx.saved_tensors
        dz, dgamma, dbeta = ops.bn1d_bwd(dy.contiguous(), z, y, gamma, mean, var, ctx.eps, 1)
        dinp, dW, db = _linear_bwd(dz, inp, W)
        # a bias ahead of a train-mode BatchNorm: sum_b dz is exactly zero in exact arithmetic — the fused entry writes
        # the exact value instead of its rounding residue (as the trunks do for the conv biases)
        return dinp, dW, torch.zeros_like(db), dgamma, dbeta, None, None, None


class LinearEpiFn(torch.autograd.Function):
    """fc3 with a fused tail: ``+ eye(3)`` (pointnet.py:37-43) or ``log_softmax`` (:193-194)."""

    @staticmethod
    def forward(ctx, inp, W, b, epilogue):
        inp = inp.contiguous()
        Wd = W.detach().contiguous()
        out = ops.fc_fwd(inp, Wd, b.detach().contiguous(), epilogue)
        ctx.epilogue = epilogue
        ctx.save_for_backward(inp, Wd, out)
        return out

    @staticmethod
    def backward(ctx, g):
        inp, W, out = ctx.saved_tensors
        g = g.contiguous()
        if ctx.epilogue == ops.EPI_LOG_SOFTMAX:
            g = ops.log_softmax_bwd(g, out)
        dinp, dW, db = _linear_bwd(g, inp, W)
        return dinp, dW, db, None



# ---------------------------------------------------------------------------------------------------------------
# fused sequencing: one foreign call per direction (pngpd_train_step.hip)
# ---------------------------------------------------------------------------------------------------------------
_PREC_CODE = {"fp32": 0, "bf16x3": 3, "bf16": 1}
_TRUNK_GRAD_LAYOUT = (("dW1", 192), ("db1", 64), ("dg1", 64), ("dbe1", 64),
                      ("dW2", 8192), ("db2", 128), ("dg2", 128), ("dbe2", 128),
                      ("dW3", 131072), ("db3", 1024), ("dg3", 1024), ("dbe3", 1024))
_SIZE_CACHE = {}


def _stream_ptr(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _ccall(fn_name, args, dev):
    fn = _FNS.get(fn_name)
    if fn is None:
        fn = _FNS[fn_name] = getattr(_lib.load(), fn_name)
    if torch.cuda.current_device() == dev.index:
        code = fn(ctypes.addressof(args), _stream_ptr(dev))
    else:
        with torch.cuda.device(dev):
            code = fn(ctypes.addressof(args), _stream_ptr(dev))
    if code != 0:
        _lib.check(code, fn_name)


_FNS = {}


def _dp(t):
    return t.data_ptr() if t is not None else None


def _grad_out(param):
    """The persistent gradient view a LIVE flat optimizer (optim.FlatAdam) holds for ``param`` — only while
    ``param.grad`` still is that view — or None (the backward then returns gradients through autograd)."""
    from .optim import grad_view
    return grad_view(param)


class FusedTrunkFn(torch.autograd.Function):
    """The trunk of STN3d / PointNetfeat in train mode through pngpd_trunk_train_fwd / _bwd: x (B,3,N) [, trans
    (B,3,3)] -> pooled (B,1024).  ``grad_outs``: None, or the 12 persistent gradient tensors of the parameters (a flat
    optimizer's views) — the backward then writes them in place and returns no parameter gradients."""

    @staticmethod
    def forward(ctx, x, trans, W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, g3, be3, relu_last, eps, momentum,
                bufs1, bufs2, bufs3, grad_outs, cfg):
        B, _, N = x.shape
        dev = x.device
        x = x.contiguous()
        T = trans.detach().contiguous() if trans is not None else None
        P = [t.detach().contiguous() for t in (W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, g3, be3)]
        for t in [x] + P + ([T] if T is not None else []):
            if not t.is_cuda or t.dtype != torch.float32:
                raise RuntimeError("trunk_train: expected float32 CUDA tensors")
        a = _lib.TrunkTrainArgs()
        a.x, a.trans, a.B, a.N = x.data_ptr(), _dp(T), B, N
        a.S = ops.train_splits(B, N)
        a.relu_last = int(bool(relu_last))
        a.precision = _PREC_CODE[cfg.train]
        a.fp32_side = int(cfg.fp32_side_passes)
        a.refine = cfg.refine_pool if a.precision else 0
        need_bwd = any(ctx.needs_input_grad) or grad_outs is not None
        a.need_bwd = int(need_bwd)
        a.eps, a.momentum = float(eps), float(momentum)
        (a.w1, a.b1, a.g1, a.be1, a.w2, a.b2, a.g2, a.be2, a.w3, a.b3, a.g3, a.be3) = [t.data_ptr() for t in P]
        for i, bufs in ((1, bufs1), (2, bufs2), (3, bufs3)):
            if bufs is not None:
                setattr(a, f"rm{i}", bufs[0].data_ptr()); setattr(a, f"rv{i}", bufs[1].data_ptr())
                setattr(a, f"nbt{i}", _dp(bufs[2]))
        key = (B, N, a.S, a.precision, a.fp32_side, a.refine)
        sizes = _SIZE_CACHE.get(key)
        if sizes is None:
            lib = _lib.load()
            sizes = _SIZE_CACHE[key] = (lib.pngpd_trunk_train_save_bytes(ctypes.addressof(a)),
                                        lib.pngpd_trunk_train_scratch_bytes(ctypes.addressof(a)))
        # a forward without a backward keeps nothing: its "save" is scratch too
        save = torch.empty(sizes[0], device=dev, dtype=torch.uint8) if need_bwd else None
        ws = ops._workspace(dev, sizes[1] + (0 if need_bwd else sizes[0] + 256))
        a.scratch, a.scratch_bytes = ws.data_ptr(), sizes[1]
        if need_bwd:
            a.save = save.data_ptr()
        else:
            a.save = ws.data_ptr() + ((sizes[1] + 255) & ~255)
        a.save_bytes = sizes[0]
        pooled = torch.empty(B, 1024, device=dev, dtype=torch.float32)
        idx = torch.empty(B, 1024, device=dev, dtype=torch.int32)
        zhat = torch.empty(B, 1024, device=dev, dtype=torch.float32)
        a.pooled, a.idx, a.zhat = pooled.data_ptr(), idx.data_ptr(), zhat.data_ptr()
        _ccall("pngpd_trunk_train_fwd", a, dev)
        _bump(bufs1); _bump(bufs2); _bump(bufs3)
        ctx.args, ctx.save_buf, ctx.grad_outs, ctx.sizes = a, save, grad_outs, sizes
        ctx.has_t = T is not None
        ctx.save_for_backward(x, T if T is not None else x.new_empty(0), pooled, idx, zhat, *P)
        return pooled

    @staticmethod
    def backward(ctx, dp):
        saved = ctx.saved_tensors          # raises if a parameter was modified in place since the forward
        x = saved[0]
        dev = x.device
        a = ctx.args
        dp = dp.contiguous()
        a.dp = dp.data_ptr()
        if ctx.grad_outs is not None:
            outs = ctx.grad_outs.views
            views = None
        else:
            buf = torch.empty(sum(n for _, n in _TRUNK_GRAD_LAYOUT), device=dev, dtype=torch.float32)
            views = buf.split([n for _, n in _TRUNK_GRAD_LAYOUT])
            outs = views
        for (name, _), t in zip(_TRUNK_GRAD_LAYOUT, outs):
            setattr(a, name, t.data_ptr())
        dT = None
        if ctx.has_t and ctx.needs_input_grad[1]:
            dT = torch.empty(a.B, 3, 3, device=dev, dtype=torch.float32)
        a.dT = _dp(dT)
        ws = ops._workspace(dev, ctx.sizes[1])
        a.scratch, a.scratch_bytes = ws.data_ptr(), ctx.sizes[1]
        _ccall("pngpd_trunk_train_bwd", a, dev)
        if views is None:
            ctx.grad_outs.mark_written()
        if views is None:
            grads = (None,) * 12
        else:
            v = views
            grads = (v[0].view(64, 3, 1), v[1], v[2], v[3], v[4].view(128, 64, 1), v[5], v[6], v[7],
                     v[8].view(1024, 128, 1), v[9], v[10], v[11])
        return (None, dT) + grads + (None,) * 8


class FusedHeadFn(torch.autograd.Function):
    """fc1/bn/relu -> fc2/bn/relu -> fc3 + tail (pointnet.py:35-43 / :191-194) through pngpd_head_train_fwd / _bwd."""

    @staticmethod
    def forward(ctx, inp, W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, epilogue, eps, momentum, bufs1, bufs2,
                grad_outs, target=None, loss_mean=True):
        ctx.set_materialize_grads(False)      # an output nobody differentiates arrives as None, not as a zero fill
        inp = inp.contiguous()
        dev = inp.device
        P = [t.detach().contiguous() for t in (W1, b1, g1, be1, W2, b2, g2, be2, W3, b3)]
        a = _lib.HeadTrainArgs()
        a.inp, a.B, a.K0 = inp.data_ptr(), inp.shape[0], inp.shape[1]
        a.H1, a.H2, a.k = P[0].shape[0], P[4].shape[0], P[8].shape[0]
        a.epilogue, a.eps, a.momentum = int(epilogue), float(eps), float(momentum)
        (a.W1, a.b1, a.g1, a.be1, a.W2, a.b2, a.g2, a.be2, a.W3, a.b3) = [t.data_ptr() for t in P]
        for i, bufs in ((1, bufs1), (2, bufs2)):
            if bufs is not None:
                setattr(a, f"rm{i}", bufs[0].data_ptr()); setattr(a, f"rv{i}", bufs[1].data_ptr())
                setattr(a, f"nbt{i}", _dp(bufs[2]))
        key = ("head", a.B, a.K0, a.H1, a.H2, a.k)
        sizes = _SIZE_CACHE.get(key)
        if sizes is None:
            lib = _lib.load()
            sizes = _SIZE_CACHE[key] = (lib.pngpd_head_train_save_bytes(ctypes.addressof(a)),
                                        lib.pngpd_head_train_scratch_bytes(ctypes.addressof(a)))
        save = torch.empty(sizes[0], device=dev, dtype=torch.uint8)
        a.save, a.save_bytes = save.data_ptr(), sizes[0]
        out = torch.empty(a.B, a.k, device=dev, dtype=torch.float32)
        a.out = out.data_ptr()
        loss = None
        if target is not None:
            if int(epilogue) != ops.EPI_LOG_SOFTMAX:
                raise RuntimeError("head_train: a target needs the log_softmax tail")
            if not target.is_cuda or target.dtype != torch.int64 or tuple(target.shape) != (a.B,):
                raise RuntimeError("head_train: target must be a CUDA int64 tensor of shape (B,)")
            target = target.contiguous()
            loss = torch.empty((), device=dev, dtype=torch.float32)
            a.target, a.loss, a.loss_mean = target.data_ptr(), loss.data_ptr(), int(bool(loss_mean))
        _ccall("pngpd_head_train_fwd", a, dev)
        _bump(bufs1); _bump(bufs2)
        ctx.args, ctx.save_buf, ctx.grad_outs, ctx.sizes = a, save, grad_outs, sizes
        ctx.target = target
        ctx.save_for_backward(inp, out, *P)
        if loss is None:
            return out
        return out, loss

    @staticmethod
    def backward(ctx, g, g_loss=None):
        saved = ctx.saved_tensors
        inp, P = saved[0], saved[2:]
        dev = inp.device
        a = ctx.args
        if g is None and g_loss is None:
            raise RuntimeError("head_train backward: neither the log-probabilities nor the loss carry a gradient")
        g = g.contiguous() if g is not None else None
        a.gout = _dp(g)
        if g_loss is not None:
            g_loss = g_loss.to(torch.float32).contiguous()
            a.gloss = g_loss.data_ptr()
        else:
            a.gloss = None
            if ctx.target is not None:
                a.target = None               # the loss took no part in this backward: plain log_softmax upstream
        shapes = [tuple(t.shape) for t in P]
        if ctx.grad_outs is not None:
            outs, views = ctx.grad_outs.views, None
        else:
            sizes = [((t.numel() + 63) // 64) * 64 for t in P]        # 256-byte aligned segments
            buf = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
            views = [v[:t.numel()].view(sh) for v, t, sh in zip(buf.split(sizes), P, shapes)]
            outs = views
        for name, t in zip(("dW1", "db1", "dg1", "dbe1", "dW2", "db2", "dg2", "dbe2", "dW3", "db3"), outs):
            setattr(a, name, t.data_ptr())
        dinp = torch.empty_like(inp) if ctx.needs_input_grad[0] else None
        a.dinp = _dp(dinp)
        ws = ops._workspace(dev, ctx.sizes[1])
        a.scratch, a.scratch_bytes = ws.data_ptr(), ctx.sizes[1]
        _ccall("pngpd_head_train_bwd", a, dev)
        if views is None:
            ctx.grad_outs.mark_written()
        grads = (None,) * 10 if views is None else tuple(views)
        if ctx.target is not None:
            a.target = ctx.target.data_ptr()
        return (dinp,) + grads + (None,) * 8


def _bufs(bn):
    return (bn.running_mean, bn.running_var, bn.num_batches_tracked) if bn.track_running_stats else None


def _use_fused(cfg=None):
    return (cfg.sequencing if cfg is not None else arith.default("sequencing")) == "fused" and DEBUG_STASH is None


def _grad_outs(params):
    """The optim.GradGroup of a fused piece (its ``views`` are the in-place gradient targets), or None."""
    from .optim import grad_group
    return grad_group(params)


def _ungroup(params):
    """Non-fused call path: the parameters' gradients arrive by autograd accumulation (optim.ungroup)."""
    from .optim import ungroup
    ungroup(params)


def trunk_train(mod, x, trans, relu_last):
    """Train-mode trunk of a module holding conv1..3 / bn1..3."""
    mom = mod.bn1.momentum if mod.bn1.momentum is not None else 0.1
    cfg = arith.resolve(mod)
    if _use_fused(cfg):
        params = (mod.conv1.weight, mod.conv1.bias, mod.bn1.weight, mod.bn1.bias,
                  mod.conv2.weight, mod.conv2.bias, mod.bn2.weight, mod.bn2.bias,
                  mod.conv3.weight, mod.conv3.bias, mod.bn3.weight, mod.bn3.bias)
        return FusedTrunkFn.apply(x, trans, *params, bool(relu_last), float(mod.bn1.eps), float(mom),
                                  _bufs(mod.bn1), _bufs(mod.bn2), _bufs(mod.bn3), _grad_outs(params), cfg)
    _ungroup((mod.conv1.weight, mod.conv1.bias, mod.bn1.weight, mod.bn1.bias, mod.conv2.weight, mod.conv2.bias,
              mod.bn2.weight, mod.bn2.bias, mod.conv3.weight, mod.conv3.bias, mod.bn3.weight, mod.bn3.bias))
    return TrunkTrainFn.apply(x, trans,
                              mod.conv1.weight, mod.conv1.bias, mod.bn1.weight, mod.bn1.bias,
                              mod.conv2.weight, mod.conv2.bias, mod.bn2.weight, mod.bn2.bias,
                              mod.conv3.weight, mod.conv3.bias, mod.bn3.weight, mod.bn3.bias,
                              bool(relu_last), float(mod.bn1.eps), float(mom),
                              _bufs(mod.bn1), _bufs(mod.bn2), _bufs(mod.bn3), cfg)


def fc_bn_relu_train(lin, bn, inp):
    mom = bn.momentum if bn.momentum is not None else 0.1
    return LinearBnReluFn.apply(inp, lin.weight, lin.bias, bn.weight, bn.bias, float(bn.eps), float(mom), _bufs(bn))


def fc_epilogue_train(lin, inp, epilogue):
    return LinearEpiFn.apply(inp, lin.weight, lin.bias, epilogue)


def head_train(fc1, bn1, fc2, bn2, fc3, inp, epilogue, target=None, reduction="mean", cfg=None):
    """relu(bn1(fc1(inp))) -> relu(bn2(fc2(.))) -> fc3 + tail, train mode (pointnet.py:35-43, :191-194).
    ``target`` (B,) int64: also returns ``F.nll_loss(out, target, reduction=reduction)`` (main_1v.py:74) -> (out, loss);
    in fused sequencing the loss and its backward run inside the head's own two foreign calls."""
    if reduction not in ("mean", "sum"):
        raise ValueError("reduction must be 'mean' or 'sum'")
    if _use_fused(cfg):
        mom = bn1.momentum if bn1.momentum is not None else 0.1
        params = (fc1.weight, fc1.bias, bn1.weight, bn1.bias, fc2.weight, fc2.bias, bn2.weight, bn2.bias,
                  fc3.weight, fc3.bias)
        return FusedHeadFn.apply(inp, *params, int(epilogue), float(bn1.eps), float(mom), _bufs(bn1), _bufs(bn2),
                                 _grad_outs(params), target, reduction == "mean")
    _ungroup((fc1.weight, fc1.bias, bn1.weight, bn1.bias, fc2.weight, fc2.bias, bn2.weight, bn2.bias,
              fc3.weight, fc3.bias))
    g = fc_bn_relu_train(fc1, bn1, inp)
    g = fc_bn_relu_train(fc2, bn2, g)
    out = fc_epilogue_train(fc3, g, epilogue)
    if target is None:
        return out
    return out, torch.nn.functional.nll_loss(out, target, reduction=reduction)


def loss_backward(loss):
    """``loss.backward()`` (main_1v.py:75) with a cached unit gradient: autograd's own root gradient is a
    ``ones_like`` = one fill launch per step."""
    if loss.is_cuda and loss.dim() == 0 and loss.dtype == torch.float32:
        loss.backward(gradient=_pm_one(loss.device)[0])
    else:
        loss.backward()


class GraphedTrainStep:
    """One training step (``main_1v.py:72-76``: zero_grad, forward, ``nll_loss``, backward, Adam step) of a fixed
    (batch, num_points) shape captured ONCE as a HIP graph and replayed.  A step is ≈190 kernel launches whose
    host cost (Python autograd Functions + ctypes + the foreach optimizer) exceeds the device time at the
    reference's own batch sizes (B = 64, N = 750: 2.8 ms eager for ≈0.8 ms of kernels); the replay has no host
    work beyond two input copies.

        step = GraphedTrainStep(model, batch=64, num_points=750, lr=0.005)
        loss, logp = step(x, y)          # x (64,3,750) fp32 CUDA, y (64,) int64 CUDA; static output tensors
        step.optimizer                   # torch.optim.Adam(capturable=True, lr as a device tensor -> StepLR works)

    ``optimizer``: a FlatAdam(capturable=True) (default) or a torch optimizer built with capturable=True.
    The three warm-up steps the capture needs run on zero data and are UNDONE (parameters, BatchNorm buffers and
    Adam state restored in place), so the first replay is the first real step.  Single process only: gradient
    all-reduce is not captured (use the eager loop + ``ddp.GradAverager`` under torchrun)."""

    def __init__(self, model, batch, num_points, lr=0.005, optimizer=None, warmup=3):
        import torch.nn.functional as F
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise RuntimeError("GraphedTrainStep captures no collective: use the eager loop under torchrun")
        self.model = model.train()
        p0 = next(model.parameters())
        dev = p0.device
        if dev.type != "cuda":
            raise RuntimeError("GraphedTrainStep needs the model on a CUDA device")
        k = model.fc3.out_features
        self.x = torch.zeros(batch, 3, num_points, device=dev)
        self.y = (torch.arange(batch, device=dev) % k).long()
        from .optim import FlatAdam
        if optimizer is None:
            optimizer = FlatAdam(model.parameters(), lr=torch.tensor(float(lr), device=dev), capturable=True)
        flat = isinstance(optimizer, FlatAdam)
        if flat and not optimizer.capturable:
            raise RuntimeError("the FlatAdam must be constructed with capturable=True (device-resident step / lr)")
        if not flat and not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise RuntimeError("the optimizer must be constructed with capturable=True")
        self.optimizer = optimizer
        had_state = len(optimizer.state) > 0
        saved_model = {n: t.detach().clone() for n, t in model.state_dict().items()}
        saved_opt = [(st, {n: v.detach().clone() for n, v in st.items() if torch.is_tensor(v)})
                     for st in optimizer.state.values()] if had_state else None
        saved_step = (optimizer._step, optimizer.step_dev.clone()) if flat else None

        def one_step():
            optimizer.zero_grad(set_to_none=True)
            if hasattr(self.model, "forward_loss"):
                loss, logp, _ = self.model.forward_loss(self.x, self.y)
                loss_backward(loss)
            else:
                logp, _ = self.model(self.x)
                loss = F.nll_loss(logp, self.y)
                loss.backward()
            optimizer.step()
            return loss, logp

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self.x.normal_(std=0.02)                 # any finite cloud; the effect of these steps is undone below
            for _ in range(max(1, warmup)):
                loss, logp = one_step()
            del loss, logp
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: the capture must not be invalidated by another thread's event query — torch.distributed's RCCL
        # watchdog polls the completed collectives of earlier steps (bench_strong aborted on exactly that, once in three runs)
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss, self.logp = one_step()
        # undo warm-up + capture-time side effects, in place (the graph holds these addresses)
        with torch.no_grad():
            for n, t in model.state_dict().items():
                t.copy_(saved_model[n])
            if had_state:
                for st, sv in saved_opt:
                    for n, v in sv.items():
                        if st[n].device == v.device and st[n].shape == v.shape:
                            st[n].copy_(v)
            else:
                for st in optimizer.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
            if flat:
                optimizer._step = saved_step[0]
                optimizer.step_dev.copy_(saved_step[1])
        self.x.zero_()
        self._state = [t for t in list(model.parameters()) + list(model.buffers())]

    def __call__(self, x, y):
        if tuple(x.shape) != tuple(self.x.shape) or tuple(y.shape) != tuple(self.y.shape):
            raise RuntimeError(f"GraphedTrainStep was captured for x {tuple(self.x.shape)}, y {tuple(self.y.shape)}; "
                               f"got {tuple(x.shape)}, {tuple(y.shape)}")
        self.x.copy_(x)
        self.y.copy_(y)
        self.graph.replay()
        # a replay rewrites parameters and BatchNorm buffers without any Python-side in-place op: tell the
        # version-keyed caches (eval-mode fold cache) that they changed
        for t in self._state:
            torch.autograd.graph.increment_version(t)
        return self.loss, self.logp
