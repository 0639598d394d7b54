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


_WS_CACHE = {}


def _workspace(dev, nbytes):
    """Scratch for the per-workgroup partial maxima, cached per (device, stream) and grown on demand — the launches
    that use it are ordered on that stream, so one buffer per stream is enough (no per-call allocation)."""
    if nbytes == 0:
        return None
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _WS_CACHE.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = _WS_CACHE[key] = torch.empty((nbytes + 3) // 4, device=dev, dtype=torch.float32)
    return ws


def trunk_fwd_infer(x, trans, w1, b1, w2p, b2, w3p, b3, relu_last, splits=0):
    """x (B,3,N) -> pooled (B,1024).  See pngpd_trunk_fwd_infer in include/pngpd.h.  ``splits``: workgroups per
    cloud (0 = the library's default rule); the workspace is sized for exactly the value passed to the launch."""
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
    S = int(splits) if splits and splits > 0 else lib.pngpd_trunk_infer_splits(B, N, 0)
    nbytes = lib.pngpd_trunk_workspace_bytes(B, N, S)
    ws = _workspace(x.device, nbytes)
    with _lib.device_guard(x.device):
        _lib.check(lib.pngpd_trunk_fwd_infer(_ptr(x), B, N, _ptr(trans), _ptr(w1), _ptr(b1), _ptr(w2p), _ptr(b2),
                                             _ptr(w3p), _ptr(b3), int(bool(relu_last)), S, _ptr(out), _ptr(ws),
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


def trunk_fwd_infer_bf(x, trans, w1, b1, w2x, b2, w3x, b3, relu_last, nterms=3, splits=0, want_arg=False):
    """Reduced-precision trunk on the bf16 matrix cores (pngpd_trunk_fwd_infer_bf): nterms = 3 "bf16x3" split
    products, nterms = 1 plain bf16.  x (B,3,N) float32 or bfloat16 (bf16 cloud storage).  ``want_arg``: also return the
    (B,1024) int32 arg-max points (pngpd_trunk_fwd_infer_bf_arg) — the input of the eval-mode pool refinement."""
    lib = _lib.load()
    if not isinstance(x, torch.Tensor) or not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16) or \
            not x.is_contiguous():
        raise RuntimeError("x: expected a contiguous CUDA float32 or bfloat16 tensor")
    if x.dim() != 3 or x.shape[1] != 3:
        raise RuntimeError(f"x: expected (B,3,N), got {tuple(x.shape)}")
    B, _, N = x.shape
    if trans is not None:
        _req(trans, "trans", (B, 3, 3))
    _req(w1, "w1", (64, 3)); _req(b1, "b1", (64,)); _req(b2, "b2", (128,)); _req(b3, "b3", (1024,))
    if w2x.dtype != torch.int16 or w2x.numel() != 2 * 128 * 64 or w3x.dtype != torch.int16 or w3x.numel() != 2 * 1024 * 128:
        raise RuntimeError("w2x/w3x: expected split_pack_bf16 outputs")
    out = torch.empty(B, 1024, device=x.device, dtype=torch.float32)
    S = int(splits) if splits and splits > 0 else lib.pngpd_trunk_infer_bf_splits(B, N, 0)
    nbytes = B * S * 1024 * (8 if want_arg else 4) if S > 1 else 0
    ws = _workspace(x.device, nbytes)
    if want_arg:
        arg = torch.empty(B, 1024, device=x.device, dtype=torch.int32)
        with _lib.device_guard(x.device):
            _lib.check(lib.pngpd_trunk_fwd_infer_bf_arg(_ptr(x), int(x.dtype == torch.bfloat16), B, N, _ptr(trans),
                                                        _ptr(w1), _ptr(b1), _ptr(w2x), _ptr(b2), _ptr(w3x), _ptr(b3),
                                                        int(bool(relu_last)), int(nterms), S, _ptr(out), _ptr(arg),
                                                        _ptr(ws), nbytes, _stream(x)), "trunk_fwd_infer_bf_arg")
        return out, arg
    with _lib.device_guard(x.device):
        _lib.check(lib.pngpd_trunk_fwd_infer_bf(_ptr(x), int(x.dtype == torch.bfloat16), B, N, _ptr(trans), _ptr(w1),
                                                _ptr(b1), _ptr(w2x), _ptr(b2), _ptr(w3x), _ptr(b3),
                                                int(bool(relu_last)), int(nterms), S, _ptr(out), _ptr(ws), nbytes,
                                                _stream(x)), "trunk_fwd_infer_bf")
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


TRAIN_TARGET_BLOCKS = 0        # 0: the library's cost model (pngpd_trunk_splits); > 0: aim at that many workgroups


def train_splits(B, N, target_blocks=None):
    """Workgroups per cloud for the training passes.  A pure function of its arguments: the caller passes the
    result to every pass AND sizes the partial buffers with it (the library keeps no tuning state)."""
    return _lib.load().pngpd_trunk_splits(int(B), int(N), int(target_blocks or TRAIN_TARGET_BLOCKS or 0))


def pack_mfma_b(W, scale=None):
    """(C,K) fp32 -> MFMA_B packed flat tensor (optionally rows scaled by ``scale``).  No bias operand: the fold
    kernel reads a NULL bias as zeros."""
    lib = _lib.load()
    W = _req(W.detach().contiguous(), "W")
    C, K = W.shape
    if scale is not None:
        scale = _req(scale.detach().contiguous(), "scale", (C,))
    wf = torch.empty(C * K, device=W.device, dtype=torch.float32)
    bf = torch.empty(C, device=W.device, dtype=torch.float32)
    with _lib.device_guard(W.device):
        _lib.check(lib.pngpd_fold_conv_bn(_ptr(W), None, _ptr(scale), None, None, None, 0.0, C, K, LAYOUT_MFMA_B,
                                          _ptr(wf), _ptr(bf), _stream(W)), "pack_mfma_b")
    return wf


def cloud_moments(x):
    B, _, N = x.shape
    mom = torch.empty(B, 9, device=x.device, dtype=torch.float64)
    _call("pngpd_cloud_moments", x, _f32(x, "x"), B, N, mom)
    return mom


def trunk_bn2_stats(x, trans, w1, b1, s1c, t1c, w2p, S, store_z2=True):
    """-> part (B*S,128,2) partial sums of z2 = W2 h1, z2t (z2 itself in the lane-major tile layout the later passes
    read back instead of recomputing layers 1-2, or None)."""
    B, _, N = x.shape
    part = torch.empty(B * S, 128, 2, device=x.device, dtype=torch.float32)
    z2t = torch.empty(_lib.load().pngpd_trunk_g2t_bytes(B, N) // 4, device=x.device, dtype=torch.float32) \
        if store_z2 else None
    _call("pngpd_trunk_bn2_stats", x, x, B, N, trans, w1, b1, s1c, t1c, w2p, int(S), part, z2t)
    return part, z2t


def _tile_buffer(x, nterms):
    """z2 / g2 tile storage of the bf16 passes: fp32 tiles (bf16x3) or bf16 tiles of half the size (plain bf16)."""
    B, _, N = x.shape
    nbytes = _lib.load().pngpd_trunk_g2t_bytes(B, N)
    if int(nterms) == 1:
        return torch.empty(nbytes // 4, device=x.device, dtype=torch.int16)
    return torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)


def tiles_bf16_to_f32(t16):
    """bf16 tiles [(b*T+tile)][4][256][8] -> the fp32 tile layout [(b*T+tile)][8][256][4] (tests / debugging)."""
    v = (t16.view(-1, 4, 256, 8).to(torch.int32) << 16).view(torch.float32)
    return v.view(-1, 4, 256, 2, 4).permute(0, 1, 3, 2, 4).reshape(-1)


def trunk_bn2_stats_bf(x, trans, w1, b1, s1c, t1c, w2x, S, nterms, store_z2=True):
    """Pass B with layer 2 on bf16 (nterms 1) / bf16x3 (3) operands; w2x = split_pack_bf16(W2).  The z2 tiles are
    fp32 for nterms 3 and bf16 (int16 tensor, half the bytes) for nterms 1."""
    B, _, N = x.shape
    part = torch.empty(B * S, 128, 2, device=x.device, dtype=torch.float32)
    z2t = _tile_buffer(x, nterms) if store_z2 else None
    _call("pngpd_trunk_bn2_stats_bf", x, x, B, N, trans, w1, b1, s1c, t1c, w2x, int(nterms), int(S), part, z2t)
    return part, z2t


def trunk_fwd_train(x, trans, w1, b1, s1c, t1c, w2p, s2c, t2c, w3sp, S, z2t=None):
    """-> pmax (B,S,1024), parg, psum (B*S,2,1024), psh (B*S,128).  z2t: pass B's stored z2 (read back instead of
    recomputing layers 1-2) or None."""
    B, _, N = x.shape
    pmax = torch.empty(B, S, 1024, device=x.device, dtype=torch.float32)
    parg = torch.empty(B, S, 1024, device=x.device, dtype=torch.int32)
    psum = torch.empty(B * S, 2, 1024, device=x.device, dtype=torch.float32)
    psh = torch.empty(B * S, 128, device=x.device, dtype=torch.float32)
    _call("pngpd_trunk_fwd_train", x, x, B, N, trans, w1, b1, s1c, t1c, w2p, s2c, t2c, w3sp, int(S), pmax, parg,
          psum, psh, z2t)
    return pmax, parg, psum, psh


def trunk_fwd_train_bf(x, trans, w1, b1, s1c, t1c, w2x, s2c, t2c, w3sx, S, nterms=3, z2t=None):
    """bf16 matrix-core variant of trunk_fwd_train (nterms 3 = bf16x3, 1 = plain bf16); returns (pmax (B,Sx,1024),
    parg, psum (B*Sx,2,1024), psh (B*Sx*2,128), Sx) with Sx = min(S, ceil(N/128)) (128-point tiles).  z2t: pass B's
    stored z2, read back instead of recomputing layers 1-2, or None."""
    B, _, N = x.shape
    S = max(1, min(int(S), (N + 127) // 128))
    if z2t is not None and z2t.dtype != (torch.int16 if int(nterms) == 1 else torch.float32):
        raise RuntimeError("z2t: expected the tiles pngpd_trunk_bn2_stats_bf stored with the same nterms")
    pmax = torch.empty(B, S, 1024, device=x.device, dtype=torch.float32)
    parg = torch.empty(B, S, 1024, device=x.device, dtype=torch.int32)
    psum = torch.empty(B * S, 2, 1024, device=x.device, dtype=torch.float32)
    psh = torch.empty(B * S * 2, 128, device=x.device, dtype=torch.float32)
    _call("pngpd_trunk_fwd_train_bf", x, x, B, N, trans, w1, b1, s1c, t1c, w2x, s2c, t2c, w3sx, int(nterms), int(S),
          pmax, parg, psum, psh, z2t)
    return pmax, parg, psum, psh, S


def _gather_ranges(B, clouds_per_range):
    if clouds_per_range is None:      # 16 workgroups per range: ~256 workgroups at small B, 16 clouds per range at large B
        clouds_per_range = max(1, min(16, B // 16))
    return int(clouds_per_range), (B + clouds_per_range - 1) // clouds_per_range


def trunk_bwd_gather(x, trans, w1, b1, s1c, t1c, w2p, s2c, t2c, idx, coef, clouds_per_range=None):
    B, _, N = x.shape
    clouds_per_range, R = _gather_ranges(B, clouds_per_range)
    Gp = torch.empty(R, 1024, 128, device=x.device, dtype=torch.float32)
    _call("pngpd_trunk_bwd_gather", x, x, B, N, trans, w1, b1, s1c, t1c, w2p, s2c, t2c, idx, coef,
          int(clouds_per_range), Gp)
    return Gp


def trunk_bwd_gather_bf(x, trans, w1, b1, s1c, t1c, w2x, s2c, t2c, idx, coef, nterms, clouds_per_range=None):
    B, _, N = x.shape
    cpr, R = _gather_ranges(B, clouds_per_range)
    Gp = torch.empty(R, 1024, 128, device=x.device, dtype=torch.float32)
    _call("pngpd_trunk_bwd_gather_bf", x, x, B, N, trans, w1, b1, s1c, t1c, w2x, int(nterms), s2c, t2c, idx, coef,
          cpr, Gp)
    return Gp


def trunk_pool_refine(x, trans, w1, b1, s1c, t1c, w2p, s2c, t2c, idx, w3sp=None, w3=None, g3=None, variant=0,
                      clouds_per_range=None):
    """zex (B,1024): the exact fp32 value of z3s at the arg-max points ``idx`` (pngpd_trunk_pool_refine)."""
    B, _, N = x.shape
    cpr = int(clouds_per_range) if clouds_per_range else max(1, min(16, B // 16))
    zex = torch.empty(B, 1024, device=x.device, dtype=torch.float32)
    _call("pngpd_trunk_pool_refine", x, x, B, N, trans, w1, b1, s1c, t1c, w2p, s2c, t2c, w3sp, w3, g3, idx.contiguous(),
          cpr, int(variant), zex)
    return zex


def unpack_mfma_b_128(Ap):
    """pngpd_a_cvec_finalize's MFMA_B-packed 128x128 matrix -> row-major (128,128)."""
    return Ap.view(4, 16, 2, 32, 4).permute(0, 3, 1, 2, 4).reshape(128, 128)


def trunk_bwd_d_bf(x, s2c, t2c, is2, nm2, Ax, cvec, w3, idx, coef, S, z2t, nterms):
    """Pass D on bf16 / bf16x3 operands; Ax = split_pack_bf16 of the (symmetric) matrix A; z2t required."""
    B, _, N = x.shape
    if z2t is None or z2t.dtype != (torch.int16 if int(nterms) == 1 else torch.float32):
        raise RuntimeError("z2t: expected the tiles pngpd_trunk_bn2_stats_bf stored with the same nterms")
    g2t = _tile_buffer(x, nterms)
    pa = torch.empty(B * S, 128, 2, device=x.device, dtype=torch.float32)
    ps2 = torch.empty(B * S, 12 * 1024, device=x.device, dtype=torch.float32)
    _call("pngpd_trunk_bwd_d_bf", x, x, B, N, s2c, t2c, is2, nm2, Ax, int(nterms), cvec, w3, idx, coef, z2t, int(S),
          g2t, pa, ps2)
    return g2t, pa, ps2


def trunk_bwd_e_bf(x, trans, w1, b1, s1c, t1c, is1, nm1, is2, nm2, a1m, a2m, dsc2, w2tx, g2t, S, z2t, nterms):
    """Pass E on bf16 / bf16x3 operands; w2tx = split_pack_bf16(W2^T as (64,128)); z2t required."""
    B, _, N = x.shape
    want = torch.int16 if int(nterms) == 1 else torch.float32
    if z2t is None or z2t.dtype != want or g2t.dtype != want:
        raise RuntimeError("z2t / g2t: expected the tiles the bf16 passes stored with the same nterms")
    pc = torch.empty(B * S, 64, 2, device=x.device, dtype=torch.float32)
    pR = torch.empty(B, S, 64, 3, device=x.device, dtype=torch.float32)
    pW2 = torch.empty(B * S, 128, 64, device=x.device, dtype=torch.float32)
    _call("pngpd_trunk_bwd_e_bf", x, x, B, N, trans, w1, b1, s1c, t1c, is1, nm1, is2, nm2, a1m, a2m, dsc2, w2tx,
          int(nterms), z2t, g2t, int(S), pc, pR, pW2)
    return pc, pR, pW2


def trunk_bwd_d(x, trans, w1, b1, s1c, t1c, w2p, s2c, t2c, is2, nm2, Ap, cvec, w3, idx, coef, S, z2t=None):
    """-> g2t (pass D -> pass E hand-off, opaque tile layout), pa (B*S,128,2), ps2 (B*S,12,16,64)."""
    B, _, N = x.shape
    g2t = torch.empty(_lib.load().pngpd_trunk_g2t_bytes(B, N) // 4, device=x.device, dtype=torch.float32)
    pa = torch.empty(B * S, 128, 2, device=x.device, dtype=torch.float32)
    ps2 = torch.empty(B * S, 12 * 1024, device=x.device, dtype=torch.float32)
    _call("pngpd_trunk_bwd_d", x, x, B, N, trans, w1, b1, s1c, t1c, w2p, s2c, t2c, is2, nm2, Ap, cvec, w3,
          idx, coef, z2t, int(S), g2t, pa, ps2)
    return g2t, pa, ps2


def g2t_to_rows(g2t, B, N):
    """Pass D's lane-major hand-off -> (B,N,128) rows (tests / debugging only)."""
    T = (N + 63) // 64
    t = g2t.view(B, T, 8, 4, 2, 32, 4)                    # [b][tile][q][wave][h][j][e], value v = 4q+e
    t = t.permute(0, 1, 3, 5, 4, 2, 6).reshape(B, T, 4, 32, 2, 32)          # [b][tile][wave][j][h][v]
    v = torch.arange(32, device=g2t.device)
    r = v & 15
    row0 = (r & 3) + 8 * (r >> 2) + 32 * (v >> 4)                            # + 4*h
    out = torch.empty(B, T, 64, 128, device=g2t.device, dtype=g2t.dtype)
    for h in range(2):
        out[:, :, row0 + 4 * h, :] = t[:, :, :, :, h, :].permute(0, 1, 4, 2, 3).reshape(B, T, 32, 128)
    return out.reshape(B, T * 64, 128)[:, :N].contiguous()


def trunk_bwd_e(x, trans, w1, b1, s1c, t1c, w2p, is1, nm1, is2, nm2, a1m, a2m, dsc2, w2tp, g2t, S, z2t=None):
    B, _, N = x.shape
    pc = torch.empty(B * S, 64, 2, device=x.device, dtype=torch.float32)
    pR = torch.empty(B, S, 64, 3, device=x.device, dtype=torch.float32)
    pW2 = torch.empty(B * S, 128, 64, device=x.device, dtype=torch.float32)
    _call("pngpd_trunk_bwd_e", x, x, B, N, trans, w1, b1, s1c, t1c, w2p, is1, nm1, is2, nm2, a1m, a2m, dsc2,
          w2tp, z2t, g2t, int(S), pc, pR, pW2)
    return pc, pR, pW2


def reduce4(*segs):
    """Up to four (tensor, outer, R, n) partial reductions in ONE launch -> list of (outer,n) fp64 tensors."""
    assert 1 <= len(segs) <= 4
    outs, args = [], []
    ref = segs[0][0]
    for t, outer, R, n in segs:
        o = torch.empty(int(outer), int(n), device=t.device, dtype=torch.float64)
        outs.append(o)
        args += [t, int(outer), int(R), int(n), o]
    for _ in range(4 - len(segs)):
        args += [None, 0, 0, 0, None]
    _call("pngpd_reduce_partials4", ref, *args)
    return outs


def fc_bwd(g, inp, W):
    """Backward of ``inp @ W^T + b``: -> (dinp (B,K), dW (Nout,K), db (Nout)) in one launch, no transposed copies."""
    B, K = inp.shape
    Nout = W.shape[0]
    _f32(g, "g", (B, Nout)); _f32(inp, "inp", (B, K)); _f32(W, "W", (Nout, K))
    dinp = torch.empty(B, K, device=g.device, dtype=torch.float32)
    dW = torch.empty(Nout, K, device=g.device, dtype=torch.float32)
    db = torch.empty(Nout, device=g.device, dtype=torch.float32)
    _call("pngpd_fc_bwd", g, g, inp, W, B, K, Nout, dW, dinp, db)
    return dinp, dW, db


def bn1d_fwd_train(z, gamma, beta, eps, relu, momentum=0.1, bufs=None):
    """Batch-statistics BatchNorm1d (+ReLU) over (B,C); ``bufs`` = (running_mean, running_var, num_batches_tracked)
    are updated in place by the same kernel."""
    B, C = z.shape
    y = torch.empty_like(z)
    mean = torch.empty(C, device=z.device, dtype=torch.float32)
    var = torch.empty(C, device=z.device, dtype=torch.float32)
    rm, rv, nbt = bufs if bufs is not None else (None, None, None)
    _call("pngpd_bn1d_fwd_train", z, _f32(z, "z"), B, C, _f32(gamma, "gamma", (C,)), _f32(beta, "beta", (C,)),
          float(eps), int(relu), y, mean, var, float(momentum), rm, rv, nbt)
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


def nll_fwd(logp, target, mean=True):
    """F.nll_loss(logp, target, reduction='mean'|'sum') (main_1v.py:74, :101) -> 0-dim fp32 tensor."""
    B, K = logp.shape
    if not target.is_cuda or target.dtype != torch.int64 or tuple(target.shape) != (B,):
        raise RuntimeError("nll_fwd: target must be a CUDA int64 tensor of shape (B,)")
    loss = torch.empty((), device=logp.device, dtype=torch.float32)
    _call("pngpd_nll_fwd", logp, _f32(logp, "logp", (B, K)), target.contiguous(), B, K, int(bool(mean)), loss)
    return loss


def nll_log_softmax_bwd(g, gloss, target, logp, mean=True):
    """Backward of ``F.nll_loss(F.log_softmax(z), target)`` w.r.t. z in one launch: upstream ``g`` (B,K) on the
    log-probabilities (or None) plus ``gloss`` (0-dim fp32) on the loss."""
    B, K = logp.shape
    out = torch.empty_like(logp)
    _call("pngpd_nll_log_softmax_bwd", logp, None if g is None else _f32(g, "g", (B, K)), gloss.contiguous(),
          target.contiguous(), logp, B, K, int(bool(mean)), out)
    return out
