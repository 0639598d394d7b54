"""Train-mode forward/backward of the hot path as ``torch.autograd.Function``s over the libpngpd
training passes (include/pngpd.h "Training path").

What runs where
---------------
* every per-point / per-sample computation (the per-point MLP recompute passes, the max-pool,
  second moments, the FC GEMMs, BatchNorm1d over the batch, log-softmax backward) is a HIP kernel
  reached through the C ABI;
* the *parameter-sized* closed-form algebra that turns the passes' accumulated sums into BatchNorm
  statistics and weight gradients (64x3 … 1024x128 matrices, fp64) and the reduction of
  per-workgroup partial buffers is done here with torch ops on the same device/stream.

The algebra is derived in DESIGN.md ("Training passes") and verified against autograd in fp64 by
``tests/train_algo_prototype.py``.  Semantics match the reference's train-mode graph
(pointnet.py:27-45,137-154,189-194 under main_1v.py:72-76): batch-statistics BatchNorm with
eps 1e-5, running statistics updated with momentum 0.1 and the unbiased variance.
"""
import torch

from . import ops

F64 = torch.float64
DEBUG_STASH = None   # set to a dict to capture backward intermediates (tools/diag_train.py)


def _update_running(buffers, mean, var_biased, count, momentum):
    """nn.BatchNorm1d running-stat update (momentum form, unbiased variance)."""
    if buffers is None:
        return
    rm, rv, nbt = buffers
    with torch.no_grad():
        unbiased = var_biased * (count / max(count - 1, 1))
        rm.mul_(1 - momentum).add_(mean.to(rm.dtype), alpha=momentum)
        rv.mul_(1 - momentum).add_(unbiased.to(rv.dtype), alpha=momentum)
        if nbt is not None:
            nbt.add_(1)


def _sym3(m6):
    """(B,6) = xx,xy,xz,yy,yz,zz -> (B,3,3)."""
    xx, xy, xz, yy, yz, zz = m6.unbind(1)
    return torch.stack([torch.stack([xx, xy, xz], 1), torch.stack([xy, yy, yz], 1),
                        torch.stack([xz, yz, zz], 1)], 1)


class TrunkTrainFn(torch.autograd.Function):
    """x (B,3,N) [, trans (B,3,3)] -> pooled (B,1024) through conv1/bn1/relu, conv2/bn2/relu,
    conv3/bn3[/relu], max over N — batch-statistics BatchNorm.  Gradients for the 12 parameters
    and for ``trans``."""

    @staticmethod
    def forward(ctx, x, trans, W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, g3, be3, relu_last, eps,
                momentum, bufs1, bufs2, bufs3):
        B, _, N = x.shape
        M = B * N
        dev = x.device
        x = x.contiguous()
        T = trans.contiguous() if trans is not None else None
        w1 = W1.detach().reshape(64, 3).contiguous()
        w2 = W2.detach().reshape(128, 64).contiguous()
        w3 = W3.detach().reshape(1024, 128).contiguous()
        f32 = lambda t: t.to(torch.float32).contiguous()
        # ---- pass A: BN1 statistics in closed form from per-cloud moments (fp64)
        mom = ops.cloud_moments(x)
        m_b, S_b = mom[:, :3], _sym3(mom[:, 3:])
        if T is not None:
            T64 = T.detach().to(F64)
            mp_b = torch.einsum("bi,bij->bj", m_b, T64)
            Sp_b = torch.einsum("bki,bkl,blj->bij", T64, S_b, T64)
        else:
            T64, mp_b, Sp_b = None, m_b, S_b
        mx = mp_b.sum(0) / M
        Cx = Sp_b.sum(0) / M - torch.outer(mx, mx)
        W1d = w1.to(F64)
        mu1 = W1d @ mx + b1.detach().to(F64)
        var1 = torch.einsum("ci,ij,cj->c", W1d, Cx, W1d).clamp_min(0)
        is1 = torch.rsqrt(var1 + eps)
        s1c = g1.detach().to(F64) * is1
        t1c = be1.detach().to(F64) - mu1 * s1c
        nm1 = -mu1 * is1
        b1f, s1cf, t1cf = f32(b1.detach()), f32(s1c), f32(t1c)
        # ---- pass B: BN2 statistics
        w2p = ops.pack_mfma_b(w2)
        part = ops.trunk_bn2_stats(x, T, w1, b1f, s1cf, t1cf, w2p)
        tot = part.sum(0, dtype=F64)
        mu2r = tot[:, 0] / M
        var2 = (tot[:, 1] / M - mu2r * mu2r).clamp_min(0)
        is2 = torch.rsqrt(var2 + eps)
        s2c = g2.detach().to(F64) * is2
        t2c = be2.detach().to(F64) - mu2r * s2c
        nm2 = -mu2r * is2
        s2cf, t2cf = f32(s2c), f32(t2c)
        # ---- pass C: layer 3 (sign-folded), statistics + max/argmax
        sgn = torch.where(g3.detach() >= 0, torch.ones_like(g3), -torch.ones_like(g3)).to(torch.float32)
        w3sp = ops.pack_mfma_b(w3, scale=sgn.contiguous())
        pmax, parg, psum = ops.trunk_fwd_train(x, T, w1, b1f, s1cf, t1cf, w2p, s2cf, t2cf, w3sp)
        tot3 = psum.sum(0, dtype=F64)
        mu3s = tot3[0] / M
        var3 = (tot3[1] / M - mu3s * mu3s).clamp_min(0)
        zmax, si = pmax.max(1)
        idx = parg.gather(1, si.unsqueeze(1)).squeeze(1).contiguous()
        sig3 = torch.sqrt(var3 + eps)
        sgn64 = sgn.to(F64)
        zhat_ext = sgn64 * (zmax.to(F64) - mu3s) / sig3
        y = g3.detach().to(F64) * zhat_ext + be3.detach().to(F64)
        pooled = (torch.relu(y) if relu_last else y).to(torch.float32)
        # ---- running statistics of the reference's pre-BN activations
        _update_running(bufs1, mu1, var1, M, momentum)
        _update_running(bufs2, mu2r + b2.detach().to(F64), var2, M, momentum)
        _update_running(bufs3, sgn64 * mu3s + b3.detach().to(F64), var3, M, momentum)
        ctx.relu_last, ctx.eps, ctx.M, ctx.has_t = relu_last, eps, M, T is not None
        ctx.save_for_backward(x, T if T is not None else x.new_empty(0), w1, b1f, g1.detach(), w2, g2.detach(),
                              w3, g3.detach(), m_b, S_b, mx, Cx, mu1, var1, var2, var3, sig3, zhat_ext, y, idx,
                              s1cf, t1cf, f32(is1), f32(nm1), w2p, s2cf, t2cf, f32(is2), f32(nm2))
        return pooled

    @staticmethod
    def backward(ctx, dp):
        (x, T, w1, b1f, g1, w2, g2, w3, g3, m_b, S_b, mx, Cx, mu1, var1, var2, var3, sig3, zhat_ext, y, idx,
         s1cf, t1cf, is1f, nm1f, w2p, s2cf, t2cf, is2f, nm2f) = ctx.saved_tensors
        T = T if ctx.has_t else None
        M, eps = ctx.M, ctx.eps
        B, _, N = x.shape
        f32 = lambda t: t.to(torch.float32).contiguous()
        dp = dp.to(F64)
        if ctx.relu_last:
            dp = dp * (y > 0).to(F64)
        g1d, g2d, g3d = g1.to(F64), g2.to(F64), g3.to(F64)
        W1d, W2d, W3d = w1.to(F64), w2.to(F64), w3.to(F64)
        # ---- BN3 affine grads + dense-correction scalars
        dg3 = (dp * zhat_ext).sum(0)
        dbe3 = dp.sum(0)
        m1, m2 = dbe3 / M, dg3 / M
        s3 = g3d / sig3
        coef = f32(dp * s3[None, :])
        # ---- hidden-activation moments, sparse gather
        ps2, ps1, psh = ops.trunk_h_moments(x, T, w1, b1f, s1cf, t1cf, w2p, s2cf, t2cf)
        S2 = ps2.sum(0, dtype=F64); S1 = ps1.sum(0, dtype=F64)
        shs = psh.sum(0, dtype=F64); sh, sh1 = shs[:128], shs[128:]
        mh, mh1 = sh / M, sh1 / M
        Sc = S2 - M * torch.outer(mh, mh)
        Sc1 = S1 - M * torch.outer(mh1, mh1)
        Gp = ops.trunk_bwd_gather(x, T, w1, b1f, s1cf, t1cf, w2p, s2cf, t2cf, idx, coef)
        G = Gp.sum(0, dtype=F64)
        dW3 = G - s3[:, None] * (m1[:, None] * sh[None, :] + (m2 / sig3)[:, None] * (W3d @ Sc))
        # ---- pass D
        Dv = g3d * m2 / (sig3 * sig3)
        A = W3d.T @ (Dv[:, None] * W3d)
        A = 0.5 * (A + A.T)
        u = W3d.T @ (s3 * m1)
        cvec = f32(A @ mh - u)
        Ap = ops.pack_mfma_b(f32(A))
        g2buf, pa, pP = ops.trunk_bwd_d(x, T, w1, b1f, s1cf, t1cf, w2p, s2cf, t2cf, is2f, nm2f, Ap, cvec, w3,
                                        idx, coef)
        pas = pa.sum(0, dtype=F64)
        a1, a2 = pas[:, 0], pas[:, 1]
        Pm = pP.sum(0, dtype=F64)
        sig2 = torch.sqrt(var2 + eps)
        s2 = g2d / sig2
        dW2 = s2[:, None] * (Pm - (a1 / M)[:, None] * sh1[None, :] - (a2 / (M * sig2))[:, None] * (W2d @ Sc1))
        # ---- pass E
        w2tp = ops.pack_mfma_b(w2.t().contiguous())
        pc, pR = ops.trunk_bwd_e(x, T, w1, b1f, s1cf, t1cf, w2p, is1f, nm1f, is2f, nm2f, f32(a1 / M), f32(a2 / M),
                                 f32(s2), w2tp, g2buf)
        pcs = pc.sum(0, dtype=F64)
        c1, c2 = pcs[:, 0], pcs[:, 1]
        Rb = pR.sum(1, dtype=F64)                      # (B,64,3)
        sig1 = torch.sqrt(var1 + eps)
        s1 = g1d / sig1
        if T is not None:
            T64 = T.to(F64)
            Rp = torch.einsum("bci,bij->cj", Rb, T64)
        else:
            Rp = Rb.sum(0)
        dW1 = s1[:, None] * (Rp - (c1 / M)[:, None] * (mx * M)[None, :]
                             - (c2 / (M * sig1))[:, None] * (W1d @ (Cx * M)))
        dT = None
        if T is not None and ctx.needs_input_grad[1]:
            Sx_xp = torch.einsum("bik,bkj->bij", S_b, T64)
            b1d = b1f.to(F64)
            term3 = (torch.einsum("bij,cj->bic", Sx_xp, W1d) + m_b[:, :, None] * (b1d - mu1)[None, None, :]) \
                / sig1[None, None, :]
            Y = s1[None, None, :] * (Rb.transpose(1, 2) - m_b[:, :, None] * (c1 / M)[None, None, :]
                                     - term3 * (c2 / M)[None, None, :])
            dT = torch.einsum("bic,cj->bij", Y, W1d).to(torch.float32)
        if DEBUG_STASH is not None:
            DEBUG_STASH.update(dict(dp=dp, dg3=dg3, dbe3=dbe3, S2=S2, S1=S1, sh=sh, sh1=sh1, G=G, A=A, cvec=cvec,
                                    a1=a1, a2=a2, Pm=Pm, c1=c1, c2=c2, Rb=Rb, dW1=dW1, dW2=dW2, dW3=dW3, dT=dT,
                                    g2buf=g2buf, idx=idx, coef=coef))
        z = lambda t: torch.zeros_like(t, dtype=torch.float32)
        o = lambda t, ref: t.to(torch.float32).reshape(ref)
        return (None, dT,
                o(dW1, (64, 3, 1)), z(b1f), o(c2, (64,)), o(c1, (64,)),
                o(dW2, (128, 64, 1)), z(g2), o(a2, (128,)), o(a1, (128,)),
                o(dW3, (1024, 128, 1)), z(g3), o(dg3, (1024,)), o(dbe3, (1024,)),
                None, None, None, None, None, None)


def _pad_cols(t, mult=8):
    """Zero-pad the last dim to a multiple of ``mult`` (fc kernel contracts K % 8 == 0)."""
    k = t.shape[-1]
    pad = (-k) % mult
    if pad == 0:
        return t.contiguous()
    return torch.nn.functional.pad(t, (0, pad)).contiguous()


def _linear_bwd(g, inp, W):
    """g (B,Nout) -> (dinp (B,K), dW (Nout,K), db (Nout)) with the MFMA FC kernel:
    dW = g^T inp (contraction over the batch), dinp = g W (contraction over Nout)."""
    Nout, K = W.shape
    zero_k = torch.zeros(K, device=g.device, dtype=torch.float32)
    gt = _pad_cols(g.t())                      # (Nout, B')
    it = _pad_cols(inp.t())                    # (K, B')
    dW = ops.fc_fwd(gt, it, zero_k, ops.EPI_NONE)          # (Nout, K)
    gp = _pad_cols(g)                          # (B, Nout')
    wt = _pad_cols(W.t())                      # (K, Nout')
    dinp = ops.fc_fwd(gp, wt, zero_k, ops.EPI_NONE)        # (B, K)
    db = g.sum(0)
    return dinp, dW, db


class LinearBnReluFn(torch.autograd.Function):
    """relu(BatchNorm1d_train(inp @ W^T + b))  (pointnet.py:35-36,191-192)."""

    @staticmethod
    def forward(ctx, inp, W, b, gamma, beta, eps, momentum, bufs):
        inp = inp.contiguous()
        Wd, bd = W.detach().contiguous(), b.detach().contiguous()
        z = ops.fc_fwd(inp, Wd, bd, ops.EPI_NONE)
        y, mean, var = ops.bn1d_fwd_train(z, gamma.detach().contiguous(), beta.detach().contiguous(), eps, 1)
        _update_running(bufs, mean, var, inp.shape[0], momentum)
        ctx.eps = eps
        ctx.save_for_backward(inp, Wd, gamma.detach().contiguous(), z, y, mean, var)
        return y

    @staticmethod
    def backward(ctx, dy):
        inp, W, gamma, z, y, mean, var = ctx.saved_tensors
        dz, dgamma, dbeta = ops.bn1d_bwd(dy.contiguous(), z, y, gamma, mean, var, ctx.eps, 1)
        dinp, dW, db = _linear_bwd(dz, inp, W)
        return dinp, dW, db, dgamma, dbeta, None, None, None


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


def _bufs(bn):
    return (bn.running_mean, bn.running_var, bn.num_batches_tracked) if bn.track_running_stats else None


def trunk_train(mod, x, trans, relu_last):
    """Train-mode trunk of a module holding conv1..3 / bn1..3."""
    mom = mod.bn1.momentum if mod.bn1.momentum is not None else 0.1
    return TrunkTrainFn.apply(x, trans,
                              mod.conv1.weight, mod.conv1.bias, mod.bn1.weight, mod.bn1.bias,
                              mod.conv2.weight, mod.conv2.bias, mod.bn2.weight, mod.bn2.bias,
                              mod.conv3.weight, mod.conv3.bias, mod.bn3.weight, mod.bn3.bias,
                              bool(relu_last), float(mod.bn1.eps), float(mom),
                              _bufs(mod.bn1), _bufs(mod.bn2), _bufs(mod.bn3))


def fc_bn_relu_train(lin, bn, inp):
    mom = bn.momentum if bn.momentum is not None else 0.1
    return LinearBnReluFn.apply(inp, lin.weight, lin.bias, bn.weight, bn.bias, float(bn.eps), float(mom), _bufs(bn))


def fc_epilogue_train(lin, inp, epilogue):
    return LinearEpiFn.apply(inp, lin.weight, lin.bias, epilogue)
