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
    with torch.cuda.device(w.device):
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
    with torch.cuda.device(x.device):
        _lib.check(lib.pngpd_trunk_fwd_infer(_ptr(x), B, N, _ptr(trans), _ptr(w1), _ptr(b1), _ptr(w2p), _ptr(b2),
                                             _ptr(w3p), _ptr(b3), int(bool(relu_last)), _ptr(out), _ptr(ws),
                                             nbytes, _stream(x)), "trunk_fwd_infer")
    return out


def fc_fwd(inp, W, bias, epilogue):
    """out = epilogue(inp @ W^T + bias).  inp (B,K), W (Nout,K)."""
    lib = _lib.load()
    _req(inp, "in")
    B, K = inp.shape
    Nout = W.shape[0]
    _req(W, "W", (Nout, K)); _req(bias, "bias", (Nout,))
    out = torch.empty(B, Nout, device=inp.device, dtype=torch.float32)
    with torch.cuda.device(inp.device):
        _lib.check(lib.pngpd_fc_fwd(_ptr(inp), B, K, _ptr(W), _ptr(bias), Nout, int(epilogue), _ptr(out),
                                    _stream(inp)), "fc_fwd")
    return out
