"""Pass-structured prototype of the train-mode trunk algorithm (test infrastructure).

Mirrors, pass by pass, what the HIP training kernels compute, in plain torch (any dtype), so
(1) the algebra (closed-form backward through conv1x1 -> BatchNorm(batch stats) -> [ReLU] ->
max-pool, SURVEY.md Appendix B extended through all three layers) is verified against autograd
in fp64, and (2) each HIP pass can be compared against the matching intermediate here.

Trunk:  x (B,3,N) [optionally x' = T_b^T x]  -> conv1/bn1/relu -> conv2/bn2/relu -> conv3/bn3
        [-> relu] -> max over N -> pooled (B,1024)
"""
import torch

EPS = 1e-5


def trunk_fwd(x, T, P, relu_last):
    """P: dict W1 (64,3) b1 g1 be1 W2 (128,64) b2 g2 be2 W3 (1024,128) b3 g3 be3.
    Returns pooled (B,1024) and a dict of saved quantities (what the kernels keep)."""
    B, _, N = x.shape
    M = B * N
    xp = torch.einsum("bin,bij->bjn", x, T) if T is not None else x
    # ---- pass A: BN1 statistics in closed form from per-sample input moments
    m_b = x.sum(2)                                   # (B,3)    sum_n x
    S_b = torch.einsum("bin,bjn->bij", x, x)         # (B,3,3)  sum_n x x^T
    if T is not None:
        mp_b = torch.einsum("bi,bij->bj", m_b, T)    # sum_n x'
        Sp_b = torch.einsum("bki,bkl,blj->bij", T, S_b, T)  # sum_n x' x'^T = T^T S T
    else:
        mp_b, Sp_b = m_b, S_b
    mx = mp_b.sum(0) / M
    Cx = Sp_b.sum(0) / M - torch.outer(mx, mx)       # covariance of x'
    mu1 = P["W1"] @ mx + P["b1"]
    var1 = torch.einsum("ci,ij,cj->c", P["W1"], Cx, P["W1"])
    s1 = P["g1"] / torch.sqrt(var1 + EPS)
    W1f = P["W1"] * s1[:, None]; b1f = (P["b1"] - mu1) * s1 + P["be1"]
    # ---- pass B: BN2 statistics (recompute h1)
    h1 = torch.relu(torch.einsum("ci,bin->bcn", W1f, xp) + b1f[None, :, None])
    z2 = torch.einsum("oc,bcn->bon", P["W2"], h1)    # bias omitted: cancels in BN
    mu2r = z2.mean((0, 2)); var2 = z2.var((0, 2), unbiased=False)
    s2 = P["g2"] / torch.sqrt(var2 + EPS)
    W2f = P["W2"] * s2[:, None]; b2f = -mu2r * s2 + P["be2"]
    h2 = torch.relu(torch.einsum("oc,bcn->bon", W2f, h1) + b2f[None, :, None])
    # ---- pass C: layer 3 with sign-folded weights, stats + max/argmax
    sgn = torch.where(P["g3"] >= 0, torch.ones_like(P["g3"]), -torch.ones_like(P["g3"]))
    z3s = torch.einsum("oc,bcn->bon", P["W3"] * sgn[:, None], h2)   # sgn * (W3 h2), no bias
    mu3s = z3s.mean((0, 2)); var3 = z3s.var((0, 2), unbiased=False)
    zmax, idx = z3s.max(2)                           # (B,1024)
    sig3 = torch.sqrt(var3 + EPS)
    zhat_ext = sgn * (zmax - mu3s) / sig3            # zhat3 at the arg-extremum
    y = P["g3"] * zhat_ext + P["be3"]
    pooled = torch.relu(y) if relu_last else y
    saved = dict(x=x, T=T, xp=xp, m_b=m_b, S_b=S_b, mp_b=mp_b, Sp_b=Sp_b, mx=mx, Cx=Cx,
                 mu1=mu1, var1=var1, s1=s1, W1f=W1f, b1f=b1f,
                 mu2r=mu2r, var2=var2, s2=s2, W2f=W2f, b2f=b2f,
                 sgn=sgn, mu3s=mu3s, var3=var3, sig3=sig3, idx=idx, zhat_ext=zhat_ext, y=y,
                 relu_last=relu_last, M=M)
    # batch statistics of the *reference's* pre-BN activations (for the running-stat update)
    saved["bn_mean"] = (mu1, mu2r + P["b2"], sgn * mu3s + P["b3"])
    saved["bn_var"] = (var1, var2, var3)
    return pooled, saved


def trunk_bwd(dp, P, sv):
    """dp (B,1024): gradient wrt pooled.  Returns dict of parameter grads (+ dT if T given)."""
    x, T, xp, M = sv["x"], sv["T"], sv["xp"], sv["M"]
    B, _, N = x.shape
    if sv["relu_last"]:
        dp = dp * (sv["y"] > 0).to(dp.dtype)
    # ---- BN3 affine grads and the two dense-correction scalars per channel
    dg3 = (dp * sv["zhat_ext"]).sum(0)
    dbe3 = dp.sum(0)
    m1 = dbe3 / M; m2 = dg3 / M
    s3 = P["g3"] / sv["sig3"]                        # gamma/sigma
    coef = dp * s3[None, :]                          # (B,1024) weight of the sparse term
    # ---- recompute h1, h2 (passes D/E do this per tile)
    h1 = torch.relu(torch.einsum("ci,bin->bcn", sv["W1f"], xp) + sv["b1f"][None, :, None])
    z2 = torch.einsum("oc,bcn->bon", P["W2"], h1)
    zhat2 = (z2 - sv["mu2r"][None, :, None]) / torch.sqrt(sv["var2"] + EPS)[None, :, None]
    h2 = torch.relu(P["g2"][None, :, None] * zhat2 + P["be2"][None, :, None])
    # ---- gather pass: G_c = sum_b coef[b,c] * h2[b,:,idx[b,c]]
    h2_at = torch.gather(h2.transpose(1, 2), 1, sv["idx"][:, :, None].expand(-1, -1, 128))  # (B,1024,128)
    G = torch.einsum("bc,bck->ck", coef, h2_at)
    # ---- h2 moments (accumulated in pass D)
    sh = h2.sum((0, 2)); mh = sh / M
    Sc = torch.einsum("bkn,bln->kl", h2, h2) - M * torch.outer(mh, mh)
    # ---- dW3 (closed form)
    # (G already carries the gamma/sigma factor through coef)
    dW3 = G - s3[:, None] * (m1[:, None] * sh[None, :] + (m2 / sv["sig3"])[:, None] * (P["W3"] @ Sc))
    db3 = torch.zeros_like(P["b3"])
    # ---- dh2 = sparse - u - A (h2 - mean_h)
    D = P["g3"] * m2 / (sv["sig3"] ** 2)
    A = P["W3"].T @ (D[:, None] * P["W3"])           # (128,128)
    u = P["W3"].T @ (s3 * m1)                        # (128,)
    dh2 = -(u[None, :, None] + torch.einsum("kl,bln->bkn", A, h2 - mh[None, :, None]))
    sparse = torch.zeros_like(dh2)                   # scatter-add coef[b,c] * W3_c at point idx[b,c]
    contrib = coef[:, :, None] * P["W3"][None, :, :]                       # (B,1024,128)
    sparse.transpose(1, 2).scatter_add_(1, sv["idx"][:, :, None].expand(-1, -1, 128), contrib)
    dh2 = dh2 + sparse
    g2 = dh2 * (h2 > 0).to(dh2.dtype)                # grad wrt bn2 output
    # ---- pass D accumulations
    a1 = g2.sum((0, 2)); a2 = (g2 * zhat2).sum((0, 2))
    dg2, dbe2 = a2, a1
    sig2 = torch.sqrt(sv["var2"] + EPS)
    Pm = torch.einsum("bon,bcn->oc", g2, h1)         # (128,64)
    sh1 = h1.sum((0, 2)); mh1 = sh1 / M
    Sc1 = torch.einsum("bkn,bln->kl", h1, h1) - M * torch.outer(mh1, mh1)
    s2 = sv["s2"]
    dW2 = s2[:, None] * (Pm - (a1 / M)[:, None] * sh1[None, :] - (a2 / (M * sig2))[:, None] * (P["W2"] @ Sc1))
    db2 = torch.zeros_like(P["b2"])
    # ---- pass E: dz2 -> dh1 -> g1, accumulations (per sample, original-x coordinates)
    dz2 = s2[None, :, None] * (g2 - (a1 / M)[None, :, None] - zhat2 * (a2 / M)[None, :, None])
    dh1 = torch.einsum("oc,bon->bcn", P["W2"], dz2)
    g1 = dh1 * (h1 > 0).to(dh1.dtype)
    sig1 = torch.sqrt(sv["var1"] + EPS)
    z1 = torch.einsum("ci,bin->bcn", P["W1"], xp) + P["b1"][None, :, None]
    zhat1 = (z1 - sv["mu1"][None, :, None]) / sig1[None, :, None]
    c1 = g1.sum((0, 2)); c2 = (g1 * zhat1).sum((0, 2))     # c2 needs zhat1: accumulated in pass E
    dg1, dbe1 = c2, c1
    Rb = torch.einsum("bcn,bin->bci", g1, x)         # (B,64,3)  sum_n g1 x^T   (original coords)
    s1 = sv["s1"]
    # sum_m g1 x'^T = sum_b Rb T_b ; sum_m zhat1 x'^T = (1/sig1) W1 Cx M
    Rp = torch.einsum("bci,bij->cj", Rb, T) if T is not None else Rb.sum(0)
    dW1 = s1[:, None] * (Rp - (c1 / M)[:, None] * (sv["mx"] * M)[None, :]
                         - (c2 / (M * sig1))[:, None] * (P["W1"] @ (sv["Cx"] * M)))
    db1 = torch.zeros_like(P["b1"])
    grads = dict(W1=dW1, b1=db1, g1=dg1, be1=dbe1, W2=dW2, b2=db2, g2=dg2, be2=dbe2,
                 W3=dW3, b3=db3, g3=dg3, be3=dbe3)
    grads["_dbg"] = dict(dg3=dg3, dbe3=dbe3, S2=torch.einsum("bkn,bln->kl", h2, h2),
                         S1=torch.einsum("bkn,bln->kl", h1, h1), sh=sh, sh1=sh1, G=G, A=A,
                         cvec=A @ mh - u, a1=a1, a2=a2, Pm=Pm, c1=c1, c2=c2, Rb=Rb,
                         g2buf=g2.transpose(1, 2), idx=sv["idx"], coef=coef)
    if T is not None:
        # dT_b = Y_b W1,  Y_b[i,c] = sum_n x_n[i] dz1_n[c]
        # dz1 = s1 (g1 - c1/M - zhat1 c2/M),  zhat1_n = (W1 x'_n + b1 - mu1)/sig1
        Sx_xp = torch.einsum("bik,bkj->bij", sv["S_b"], T)          # sum_n x x'^T  (B,3,3)
        term3 = (torch.einsum("bij,cj->bic", Sx_xp, P["W1"]) +
                 sv["m_b"][:, :, None] * (P["b1"] - sv["mu1"])[None, None, :]) / sig1[None, None, :]
        Y = s1[None, None, :] * (Rb.transpose(1, 2) - sv["m_b"][:, :, None] * (c1 / M)[None, None, :]
                                 - term3 * (c2 / M)[None, None, :])
        grads["T"] = torch.einsum("bic,cj->bij", Y, P["W1"])
    return grads
