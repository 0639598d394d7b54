"""Tensor-level wrappers of the GPD-baseline kernels (include/pngpd.h "GPD baseline").

* ``project_grasps``      — the 60x60 projection images of ``BaseGraspDataset.project_pc`` / ``cal_projection``
                            (PointNetGPD/model/dataset.py:88-198) for a batch of grasps, given the in-box points in the
                            hand frame AND their normals (the reference estimates normals with open3d, which is an input
                            here: SURVEY.md §8c);
* ``register_depth_map`` / ``depth_map_to_cloud`` — PointNetGPD/ycb_cloud_generate.py:60-184;
* ``conv5_pool2``         — one convolution stage of ``GPDClassifier`` (PointNetGPD/model/gpd.py:13-24);
* ``GPDNetFn``            — the classifier as ONE autograd node for train() mode on CUDA: forward and backward
                            (``loss.backward()`` of PointNetGPD/main_1v_gpd.py:105) on libpngpd kernels only.
CUDA tensors only; there is no CPU implementation behind these calls.
"""
import numpy as np
import torch

from . import _lib
from .ops import _call

PROJECT_SIZE, VOXEL_POINT_NUM, PROJECTION_MARGIN = 60, 50, 1       # dataset.py:220-224


def _f64(t, dev):
    return torch.as_tensor(np.ascontiguousarray(t, dtype=np.float64) if isinstance(t, np.ndarray) else t,
                           dtype=torch.float64, device=dev).contiguous()


def project_grasps(points, normals, offsets, widths, chann=3, device=None):
    """points / normals (sum M_g,3) float64 (numpy or CUDA), offsets (G+1) int32, widths (G,) -> (G,60,60,chann)
    float64 CUDA tensor, bit-identical to the reference's ``project_pc`` per grasp.

    PRECONDITION (the reference's own, dataset.py:104-128): the points are IN-BOX crops (``collect_pc`` output), so
    every voxel index ``floor(coord / resolution + 30)`` lies in [0, 60).  A point outside that range is DROPPED by the
    kernel; the reference's numpy code would instead wrap a negative index to the other side of the image or raise
    ``IndexError`` for an index >= 60.  Callers feeding arbitrary clouds must crop first."""
    if chann not in (3, 12):
        raise NotImplementedError("project_chann must be 3 or 12 (dataset.py:218-219)")
    dev = device or (points.device if isinstance(points, torch.Tensor) else torch.device("cuda", torch.cuda.current_device()))
    pts, nrm = _f64(points, dev), _f64(normals, dev)
    off = torch.as_tensor(offsets, dtype=torch.int32, device=dev).contiguous()
    w = _f64(widths, dev).reshape(-1)
    G = off.numel() - 1
    if pts.shape != nrm.shape or pts.dim() != 2 or pts.shape[1] != 3 or w.numel() != G:
        raise RuntimeError("project_grasps: points/normals (M,3), offsets (G+1), widths (G)")
    out = torch.empty(G, PROJECT_SIZE, PROJECT_SIZE, chann, device=dev, dtype=torch.float64)
    _call("pngpd_gpd_projection", pts, pts, nrm, off, w, G, int(chann), PROJECT_SIZE, PROJECTION_MARGIN, VOXEL_POINT_NUM,
          out)
    return out


def register_depth_map(depth, rgb_shape, depthK, rgbK, H_RGBFromDepth, device=None):
    """ycb_cloud_generate.py:60-121 -> registered depth (hr,wr) float64 CUDA tensor."""
    dev = device or (depth.device if isinstance(depth, torch.Tensor) else torch.device("cuda", torch.cuda.current_device()))
    d = _f64(depth, dev)
    hd, wd = d.shape
    hr, wr = int(rgb_shape[0]), int(rgb_shape[1])
    dK, rK, H = np.asarray(depthK, dtype=np.float64), np.asarray(rgbK, dtype=np.float64), np.asarray(H_RGBFromDepth, dtype=np.float64)
    cam = np.concatenate([[dK[0, 0], dK[1, 1], dK[0, 2], dK[1, 2], rK[0, 0], rK[1, 1], rK[0, 2], rK[1, 2]], H[:3, :4].reshape(-1)])
    camd = torch.from_numpy(cam).to(dev)
    out = torch.empty(hr, wr, device=dev, dtype=torch.float64)
    _call("pngpd_depth_register", d, d, int(hd), int(wd), camd, hr, wr, out)
    return out


def depth_map_to_cloud(depth, rgbK, refFromRGB, objFromref, rgb=None, device=None):
    """ycb_cloud_generate.py:124-184 (organized=False) -> xyz (P,3) float64 [, colours (P,3) uint8] CUDA tensors, the
    pixels with depth > 0 in row-major order."""
    dev = device or (depth.device if isinstance(depth, torch.Tensor) else torch.device("cuda", torch.cuda.current_device()))
    d = _f64(depth, dev)
    h, w = d.shape
    rK, A, O = (np.asarray(a, dtype=np.float64) for a in (rgbK, refFromRGB, objFromref))
    cam = np.concatenate([[rK[0, 0], rK[1, 1], rK[0, 2], rK[1, 2]], A[:3, :4].reshape(-1), O[:3, :4].reshape(-1)])
    camd = torch.from_numpy(cam).to(dev)
    xyz = torch.empty(h * w, 3, device=dev, dtype=torch.float64)
    col_in = col_out = None
    if rgb is not None:
        col_in = torch.as_tensor(rgb, device=dev).to(torch.uint8).contiguous()
        if tuple(col_in.shape) != (h, w, 3):
            raise RuntimeError("rgb: expected (h,w,3)")
        col_out = torch.empty(h * w, 3, device=dev, dtype=torch.uint8)
    count = torch.zeros(1, device=dev, dtype=torch.int32)
    nbytes = _lib.load().pngpd_depth_cloud_workspace_bytes(int(h), int(w))
    ws = torch.empty((nbytes + 3) // 4, device=dev, dtype=torch.int32)
    _call("pngpd_depth_to_cloud", d, d, int(h), int(w), camd, col_in, xyz, col_out, count, ws, nbytes)
    n = int(count.item())                                   # the one host sync: the cloud's length
    return (xyz[:n], col_out[:n]) if rgb is not None else xyz[:n]


def depth_map_to_cloud_organized(depth, rgbK, refFromRGB, objFromref, rgb, device=None):
    """ycb_cloud_generate.py:124-184 with ``organized=True`` -> (h,w,6) float64 CUDA tensor: xyz (object frame) and the
    pixel's colour where depth > 0, NaN xyz and zero colour elsewhere (:147-155).  The kernel's ordered emit is scattered
    back to the pixel grid: the compacted rows ARE the depth > 0 pixels in row-major order."""
    dev = device or (depth.device if isinstance(depth, torch.Tensor) else torch.device("cuda", torch.cuda.current_device()))
    d = _f64(depth, dev)
    h, w = d.shape
    xyz, col = depth_map_to_cloud(d, rgbK, refFromRGB, objFromref, rgb=rgb, device=dev)
    out = torch.zeros(h, w, 6, device=dev, dtype=torch.float64)
    out[:, :, :3] = float("nan")
    good = d > 0
    out[good] = torch.cat([xyz, col.to(torch.float64)], 1)
    return out


def conv5_pool2(x, weight, bias):
    """Conv2d(Cin,Cout,5) + bias + MaxPool2d(2,2): x (B,Cin,H,H) fp32 CUDA -> (B,Cout,(H-4)/2,(H-4)/2)."""
    B, Cin, H, W = x.shape
    if H != W or not x.is_cuda or x.dtype != torch.float32:
        raise RuntimeError("conv5_pool2: expected a square fp32 CUDA image batch")
    Cout = weight.shape[0]
    x, weight, bias = x.contiguous(), weight.detach().contiguous(), bias.detach().contiguous()
    out = torch.empty(B, Cout, (H - 4) // 2, (H - 4) // 2, device=x.device, dtype=torch.float32)
    _call("pngpd_conv5_pool2", x, x, B, Cin, H, weight, bias, Cout, out)
    return out


def conv5_pool2_arg(x, weight, bias):
    """``conv5_pool2`` that also returns the window position of every pooled maximum (u8, first on a tie)."""
    B, Cin, H, W = x.shape
    if H != W or not x.is_cuda or x.dtype != torch.float32:
        raise RuntimeError("conv5_pool2_arg: expected a square fp32 CUDA image batch")
    Cout = weight.shape[0]
    out = torch.empty(B, Cout, (H - 4) // 2, (H - 4) // 2, device=x.device, dtype=torch.float32)
    arg = torch.empty(out.shape, device=x.device, dtype=torch.uint8)
    _call("pngpd_conv5_pool2_arg", x, x, B, Cin, H, weight, bias, Cout, out, arg)
    return out, arg


def conv5_pool2_bwd(x, weight, dout, arg, need_input_grad):
    """Backward of one stage: -> (dx or None, dW, db).  Deterministic (slice partials reduced in order)."""
    B, Cin, H, _ = x.shape
    Cout = weight.shape[0]
    dW = torch.empty_like(weight)
    db = torch.empty(Cout, device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x) if need_input_grad else None
    nbytes = _lib.load().pngpd_conv5_pool2_bwd_workspace_bytes(B, Cin, H, Cout)
    ws = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32)
    _call("pngpd_conv5_pool2_bwd", x, x, B, Cin, H, weight, Cout, dout, arg, dW, db, dx, ws, nbytes)
    return dx, dW, db


def fc_fwd_splitk(x, weight, bias, relu):
    """``[relu](x @ W^T + b)`` for a layer with few output tiles and a long contraction (fc1: 7200 -> 500): K split over
    workgroups, deterministic two-launch reduction."""
    B, K = x.shape
    Nout = weight.shape[0]
    out = torch.empty(B, Nout, device=x.device, dtype=torch.float32)
    nbytes = _lib.load().pngpd_fc_fwd_splitk_workspace_bytes(B, K, Nout)
    ws = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32)
    _call("pngpd_fc_fwd_splitk", x, x, B, K, weight, bias, Nout, int(bool(relu)), out, ws, nbytes)
    return out


class GPDNetFn(torch.autograd.Function):
    """GPDClassifier.forward (gpd.py:22-31, dropout off) as one autograd node on libpngpd: two conv+pool stages that
    record their pooling choices, fc1 + ReLU on the split-K FC entry, fc2 + log_softmax on the FC kernel; backward =
    log_softmax_bwd, fc_bwd x2 (one launch each: dW, db, dx), relu_bwd and the sparse convolution backward kernels."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, fw1, fb1, fw2, fb2):
        from . import ops
        if ctx.needs_input_grad[0]:
            raise RuntimeError("GPDNetFn: the gradient with respect to the input images is not implemented "
                               "(main_1v_gpd.py trains the weights only); pass the images without requires_grad")
        x = x.float().contiguous()
        w1, b1, w2, b2 = (t.detach().contiguous() for t in (w1, b1, w2, b2))
        fw1, fb1, fw2, fb2 = (t.detach().contiguous() for t in (fw1, fb1, fw2, fb2))
        p1, a1 = conv5_pool2_arg(x, w1, b1)                          # (B,20,28,28)
        p2, a2 = conv5_pool2_arg(p1, w2, b2)                         # (B,50,12,12)
        flat = p2.view(p2.shape[0], -1)
        h1 = fc_fwd_splitk(flat, fw1, fb1, True)
        logp = ops.fc_fwd(h1, fw2, fb2, ops.EPI_LOG_SOFTMAX)          # K = 500: the FC kernel's half-block tail
        ctx.save_for_backward(x, w1, p1, a1, w2, flat, a2, fw1, h1, fw2, logp)
        return logp

    @staticmethod
    def backward(ctx, g):
        from . import ops
        x, w1, p1, a1, w2, flat, a2, fw1, h1, fw2, logp = ctx.saved_tensors
        dz = ops.log_softmax_bwd(g.float().contiguous(), logp)
        dh1, dfw2, dfb2 = ops.fc_bwd(dz, h1, fw2)
        _call("pngpd_relu_bwd", h1, h1, dh1, h1.numel())
        dflat, dfw1, dfb1 = ops.fc_bwd(dh1, flat, fw1)
        dp1, dw2, db2 = conv5_pool2_bwd(p1, w2, dflat.view(a2.shape), a2, True)
        _, dw1, db1 = conv5_pool2_bwd(x, w1, dp1, a1, False)
        return None, dw1, db1, dw2, db2, dfw1, dfb1, dfw2, dfb2
