"""Tensor-level wrappers over the libpngpd C ABI (include/pngpd.h).

PyTorch owns the memory and the stream; every function here only validates shapes,
allocates outputs from the caching allocator and forwards raw pointers.  CUDA tensors
only — there is no CPU implementation behind these calls.
"""
import ctypes

import torch

from . import _lib

LAYOUT_ROWMAJOR = 0
LAYOUT_MFMA_B = 1
EPI_NONE, EPI_RELU, EPI_ADD_IDEN3, EPI_LOG_SOFTMAX = 0, 1, 2, 3
BN_EPS = 1e-5


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _req(t, name, shape=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA tensor (libpngpd has no CPU path)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected scalar type Float but found {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def set_option(name, value):
    _lib.check(_lib.load().pngpd_set_option(name.encode(), int(value)), "set_option")


def fold_conv_bn(weight, bias, bn_weight=None, bn_bias=None, running_mean=None, running_var=None,
                 eps=BN_EPS, layout=LAYOUT_ROWMAJOR):
    """(C,K[,1]) weight + eval-mode BatchNorm1d -> folded (Wf, bf).  gamma=None: no BN."""
    lib = _lib.load()
    w = weight.detach()
    if w.dim() == 3:
        w = w[:, :, 0]
    w = _req(w.contiguous(), "weight")
    C, K = w.shape
    b = _req(bias.detach().contiguous(), "bias", (C,))
    args = [bn_weight, bn_bias, running_mean, running_var]
    if bn_weight is not None:
        args = [_req(a.detach().contiguous(), "bn", (C,)) for a in args]
    wf = torch.empty(C * K, device=w.device, dtype=torch.float32)
    bf = torch.empty(C, device=w.device, dtype=torch.float32)
    with _lib.device_guard(w.device):
        _lib.check(lib.pngpd_fold_conv_bn(_ptr(w), _ptr(b), _ptr(args[0]), _ptr(args[1]), _ptr(args[2]),
                                          _ptr(args[3]), float(eps), C, K, layout, _ptr(wf), _ptr(bf),
                                          _stream(w)), "fold_conv_bn")
    return (wf.view(C, K) if layout == LAYOUT_ROWMAJOR else wf), bf


def trunk_fwd_infer(x, trans, w1, b1, w2p, b2, w3p, b3, relu_last):
    """x (B,3,N) -> pooled (B,1024).  See pngpd_trunk_fwd_infer in include/pngpd.h."""
    lib = _lib.load()
    _req(x, "x")
    if x.dim() != 3 or x.shape[1] != 3:
        raise RuntimeError(f"x: expected (B,3,N), got {tuple(x.shape)}")
    B, _, N = x.shape
    if trans is not None:
        _req(trans, "trans", (B, 3, 3))
    _req(w1, "w1", (64, 3)); _req(b1, "b1", (64,))
    _req(w2p, "w2p", (128 * 64,)); _req(b2, "b2", (128,))
    _req(w3p, "w3p", (1024 * 128,)); _req(b3, "b3", (1024,))
    out = torch.empty(B, 1024, device=x.device, dtype=torch.float32)
    nbytes = lib.pngpd_trunk_workspace_bytes(B, N)
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    with _lib.device_guard(x.device):
        _lib.check(lib.pngpd_trunk_fwd_infer(_ptr(x), B, N, _ptr(trans), _ptr(w1), _ptr(b1), _ptr(w2p), _ptr(b2),
                                             _ptr(w3p), _ptr(b3), int(bool(relu_last)), _ptr(out), _ptr(ws),
                                             nbytes, _stream(x)), "trunk_fwd_infer")
    return out


def split_pack_bf16(Wf):
    """(C,K) BN-folded fp32 row-major -> hi/lo bf16 fragments for the bf16x3 trunk (int16 tensor, 2*C*K)."""
    lib = _lib.load()
    Wf = _req(Wf.contiguous(), "Wf")
    C, K = Wf.shape
    out = torch.empty(2 * C * K, device=Wf.device, dtype=torch.int16)
    with _lib.device_guard(Wf.device):
        _lib.check(lib.pngpd_split_pack_bf16(_ptr(Wf), C, K, _ptr(out), _stream(Wf)), "split_pack_bf16")
    return out


def trunk_fwd_infer_x3(x, trans, w1, b1, w2x, b2, w3x, b3, relu_last):
    """bf16x3 variant of trunk_fwd_infer (see include/pngpd.h)."""
    lib = _lib.load()
    _req(x, "x")
    if x.dim() != 3 or x.shape[1] != 3:
        raise RuntimeError(f"x: expected (B,3,N), got {tuple(x.shape)}")
    B, _, N = x.shape
    if trans is not None:
        _req(trans, "trans", (B, 3, 3))
    _req(w1, "w1", (64, 3)); _req(b1, "b1", (64,)); _req(b2, "b2", (128,)); _req(b3, "b3", (1024,))
    if w2x.dtype != torch.int16 or w2x.numel() != 2 * 128 * 64 or w3x.dtype != torch.int16 or w3x.numel() != 2 * 1024 * 128:
        raise RuntimeError("w2x/w3x: expected split_pack_bf16 outputs")
    out = torch.empty(B, 1024, device=x.device, dtype=torch.float32)
    nbytes = lib.pngpd_trunk_workspace_bytes(B, N)
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    with _lib.device_guard(x.device):
        _lib.check(lib.pngpd_trunk_fwd_infer_x3(_ptr(x), B, N, _ptr(trans), _ptr(w1), _ptr(b1), _ptr(w2x), _ptr(b2),
                                                _ptr(w3x), _ptr(b3), int(bool(relu_last)), _ptr(out), _ptr(ws),
                                                nbytes, _stream(x)), "trunk_fwd_infer_x3")
    return out


def fc_fwd(inp, W, bias, epilogue):
    """out = epilogue(inp @ W^T + bias).  inp (B,K), W (Nout,K)."""
    lib = _lib.load()
    _req(inp, "in")
    B, K = inp.shape
    Nout = W.shape[0]
    _req(W, "W", (Nout, K)); _req(bias, "bias", (Nout,))
    out = torch.empty(B, Nout, device=inp.device, dtype=torch.float32)
    with _lib.device_guard(inp.device):
        _lib.check(lib.pngpd_fc_fwd(_ptr(inp), B, K, _ptr(W), _ptr(bias), Nout, int(epilogue), _ptr(out),
                                    _stream(inp)), "fc_fwd")
    return out


# ---------------------------------------------------------------------------------------------
# training passes (include/pngpd.h "Training path")
# ---------------------------------------------------------------------------------------------
_FN = {}


def _call(name, ref, *args):
    """Generic C-ABI call: tensors -> device pointers, None -> NULL, scalars as is; the stream of
    ``ref``'s device is appended.  Every tensor must be CUDA + contiguous (dtype is the callee's
    contract and is checked by the typed wrappers below).  Kept lean: a training step makes ~150 of these."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(_lib.load(), name)
    conv = []
    for a in args:
        if isinstance(a, torch.Tensor):
            if not a.is_cuda or not a.is_contiguous():
                raise RuntimeError(f"{name}: expected contiguous CUDA tensors")
            conv.append(a.data_ptr())           # argtypes are c_void_p: plain ints / None convert directly
        else:
            conv.append(a)
    dev = ref.device
    conv.append(torch.cuda.current_stream(dev).cuda_stream)
    if torch.cuda.current_device() == dev.index:
        code = fn(*conv)
    else:
        with torch.cuda.device(dev):
            code = fn(*conv)
    if code != 0:
        _lib.check(code, name)


def _f32(t, name, shape=None):
    return _req(t, name, shape)


def train_splits(B, N):
    return _lib.load().pngpd_trunk_train_splits(int(B), int(N))


def set_train_target_blocks(v):
    _lib.check(_lib.load().pngpd_train_set_target_blocks(int(v)), "train_set_target_blocks")


def pack_mfma_b(W, scale=None):
    """(C,K) fp32 -> MFMA_B packed flat tensor (optionally rows scaled by ``scale``)."""
    C, K = W.shape
    zb = torch.zeros(C, device=W.device, dtype=torch.float32)
    wp, _ = fold_conv_bn(W, zb, bn_weight=scale, layout=LAYOUT_MFMA_B) if scale is None else \
        _fold_scale(W, zb, scale)
    return wp


def _fold_scale(W, b, scale):
    lib = _lib.load()
    W = _req(W.detach().contiguous(), "W"); C, K = W.shape
    scale = _req(scale.detach().contiguous(), "scale", (C,))
    wf = torch.empty(C * K, device=W.device, dtype=torch.float32)
    bf = torch.empty(C, device=W.device, dtype=torch.float32)
    with _lib.device_guard(W.device):
        _lib.check(lib.pngpd_fold_conv_bn(_ptr(W), _ptr(b), _ptr(scale), None, None, None, 0.0, C, K,
                                          LAYOUT_MFMA_B, _ptr(wf), _ptr(bf), _stream(W)), "fold(scale)")
    return wf, bf


def cloud_moments(x):
    B, _, N = x.shape
    mom = torch.empty(B, 9, device=x.device, dtype=torch.float64)
    _call("pngpd_cloud_moments", x, _f32(x, "x"), B, N, mom)
    return mom


def trunk_bn2_stats(x, trans, w1, b1, s1c, t1c, w2p):
    B, _, N = x.shape
    blk = B * train_splits(B, N)
    part = torch.empty(blk, 128, 2, device=x.device, dtype=torch.float32)
    _call("pngpd_trunk_bn2_stats", x, x, B, N, trans, w1, b1, s1c, t1c, w2p, part)
    return part


def trunk_fwd_train(x, trans, w1, b1, s1c, t1c, w2p, s2c, t2c, w3sp):
    B, _, N = x.shape
    S = train_splits(B, N)
    pmax = torch.empty(B, S, 1024, device=x.device, dtype=torch.float32)
    parg = torch.empty(B, S, 1024, device=x.device, dtype=torch.int32)
    psum = torch.empty(B * S, 2, 1024, device=x.device, dtype=torch.float32)
    _call("pngpd_trunk_fwd_train", x, x, B, N, trans, w1, b1, s1c, t1c, w2p, s2c, t2c, w3sp, pmax, parg, psum)
    return pmax, parg, psum


def trunk_fwd_train_x3(x, trans, w1, b1, s1c, t1c, w2x, s2c, t2c, w3sx):
    """bf16x3 variant of trunk_fwd_train; returns (pmax (B,S,1024), parg, psum (B*S,2,1024), S)."""
    B, _, N = x.shape
    S = max(1, min(train_splits(B, N), (N + 127) // 128))
    pmax = torch.empty(B, S, 1024, device=x.device, dtype=torch.float32)
    parg = torch.empty(B, S, 1024, device=x.device, dtype=torch.int32)
    psum = torch.empty(B * S, 2, 1024, device=x.device, dtype=torch.float32)
    _call("pngpd_trunk_fwd_train_x3", x, x, B, N, trans, w1, b1, s1c, t1c, w2x, s2c, t2c, w3sx, int(S), pmax, parg,
          psum)
    return pmax, parg, psum, S


def h_moments_splits(B, N):
    """Workgroups per cloud for the h-moments pass: 1 once B alone fills the 256 CUs, else enough to reach them."""
    T = (N + 63) // 64
    return 1 if B >= 256 else max(1, min(T, (256 + B - 1) // B))


def trunk_h_moments(x, trans, w1, b1, s1c, t1c, w2p, s2c, t2c):
    """-> per-workgroup partials ps2 (B*S,128,128), ps1 (B*S,64,64), psh (B*S,192); reduce over dim 0."""
    B, _, N = x.shape
    S = h_moments_splits(B, N)
    ps2 = torch.empty(B * S, 128, 128, device=x.device, dtype=torch.float32)
    ps1 = torch.empty(B * S, 64, 64, device=x.device, dtype=torch.float32)
    psh = torch.empty(B * S, 192, device=x.device, dtype=torch.float32)
    _call("pngpd_trunk_h_moments", x, x, B, N, trans, w1, b1, s1c, t1c, w2p, s2c, t2c, int(S), ps2, ps1, psh)
    return ps2, ps1, psh


def trunk_bwd_gather(x, trans, w1, b1, s1c, t1c, w2p, s2c, t2c, idx, coef, clouds_per_range=None):
    B, _, N = x.shape
    if clouds_per_range is None:      # 16 workgroups per range: ~256 workgroups at small B, 16 clouds per range at large B
        clouds_per_range = max(1, min(16, B // 16))
    R = (B + clouds_per_range - 1) // clouds_per_range
    Gp = torch.empty(R, 1024, 128, device=x.device, dtype=torch.float32)
    _call("pngpd_trunk_bwd_gather", x, x, B, N, trans, w1, b1, s1c, t1c, w2p, s2c, t2c, idx, coef,
          int(clouds_per_range), Gp)
    return Gp


def trunk_bwd_d(x, trans, w1, b1, s1c, t1c, w2p, s2c, t2c, is2, nm2, Ap, cvec, w3, idx, coef):
    B, _, N = x.shape
    blk = B * train_splits(B, N)
    g2buf = torch.empty(B, N, 128, device=x.device, dtype=torch.float32)
    pa = torch.empty(blk, 128, 2, device=x.device, dtype=torch.float32)
    _call("pngpd_trunk_bwd_d", x, x, B, N, trans, w1, b1, s1c, t1c, w2p, s2c, t2c, is2, nm2, Ap, cvec, w3,
          idx, coef, g2buf, pa)
    return g2buf, pa


def trunk_bwd_e(x, trans, w1, b1, s1c, t1c, w2p, is1, nm1, is2, nm2, a1m, a2m, dsc2, w2tp, g2buf):
    B, _, N = x.shape
    S = train_splits(B, N)
    pc = torch.empty(B * S, 64, 2, device=x.device, dtype=torch.float32)
    pR = torch.empty(B, S, 64, 3, device=x.device, dtype=torch.float32)
    pW2 = torch.empty(B * S, 128, 64, device=x.device, dtype=torch.float32)
    _call("pngpd_trunk_bwd_e", x, x, B, N, trans, w1, b1, s1c, t1c, w2p, is1, nm1, is2, nm2, a1m, a2m, dsc2,
          w2tp, g2buf, pc, pR, pW2)
    return pc, pR, pW2


def bn1d_fwd_train(z, gamma, beta, eps, relu):
    B, C = z.shape
    y = torch.empty_like(z)
    mean = torch.empty(C, device=z.device, dtype=torch.float32)
    var = torch.empty(C, device=z.device, dtype=torch.float32)
    _call("pngpd_bn1d_fwd_train", z, _f32(z, "z"), B, C, _f32(gamma, "gamma", (C,)), _f32(beta, "beta", (C,)),
          float(eps), int(relu), y, mean, var)
    return y, mean, var


def bn1d_bwd(dy, z, y, gamma, mean, var, eps, relu):
    B, C = z.shape
    dz = torch.empty_like(z)
    dgamma = torch.empty(C, device=z.device, dtype=torch.float32)
    dbeta = torch.empty(C, device=z.device, dtype=torch.float32)
    _call("pngpd_bn1d_bwd", z, _f32(dy, "dy", (B, C)), z, y, B, C, gamma, mean, var, float(eps), int(relu),
          dz, dgamma, dbeta)
    return dz, dgamma, dbeta


def log_softmax_bwd(g, logp):
    B, K = logp.shape
    out = torch.empty_like(logp)
    _call("pngpd_log_softmax_bwd", logp, _f32(g, "g", (B, K)), logp, B, K, out)
    return out
