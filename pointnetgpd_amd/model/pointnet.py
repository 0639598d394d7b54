"""Drop-in mirror of the reference's ``PointNetGPD/model/pointnet.py`` module surface.

Same class names, constructor signatures, sub-module attribute names and construction
order (hence identical ``state_dict`` keys, identical seeded initialisation, and whole-module
pickles that load across implementations):

    PointNetCls(num_points=2500, input_chann=3, k=2).forward(x: (B,3,N) fp32)
        -> (log_probs (B,k), trans (B,3,3))                      reference pointnet.py:177-194
    PointNetfeat(...).forward(x) -> (global_feat (B,1024), trans)           pointnet.py:123-154
    STN3d(...).forward(x) -> trans (B,3,3)                                  pointnet.py:8-45

Dispatch is by the input's device and is explicit, never silent:

* CUDA tensor  -> libpngpd HIP kernels (``include/pngpd.h``).  A missing / unloadable library
  raises ``RuntimeError``; nothing falls back to ATen.
* CPU tensor   -> the plain ATen composite the reference itself runs on CPU
  (BASELINE config 1 "CPU PyTorch (plumbing, no GPU)").

``DualPointNetCls`` / ``DualPointNetfeat`` / ``SimpleSTN3d`` / ``PointNetDenseCls`` are never
instantiated by any reference script; they are kept as plain-ATen modules only
so that ``from model.pointnet import PointNetCls, DualPointNetCls`` (main_1v.py:16) works.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops, train

_TRUNK_WIDTHS = (64, 128, 1024)

# Inference arithmetic of the fused trunk:
#   "fp32"   exact fp32 MFMA (default; the only mode that meets the 1e-3 log-prob contract by construction)
#   "bf16x3" opt-in: 3-term split-bf16 products on the bf16 matrix cores (~2^-16 relative product error; log-probs
#            within 1e-4 of fp32, not bit-identical)
#   "bf16"   opt-in: plain single-product bf16 operands, fp32 accumulate (BASELINE configs[2]); ~2^-8 per product —
#            does NOT meet 1e-3 in general, measured bounds in tests/test_gpu_bf16.py and DESIGN.md
# In the two bf16 modes the clouds may also be STORED as bf16 ((B,3,N) torch.bfloat16, 6 B/point): the trunk kernel
# reads them directly.  In "fp32" mode a bf16 cloud is widened first.
_INFER_PRECISION = "fp32"
_INFER_REFINE = False
_NTERMS = {"bf16x3": 3, "bf16": 1}


def set_inference_precision(mode, refine=False):
    """``refine`` (reduced-precision modes only): the bf16 / bf16x3 trunk only CHOOSES the arg-max point of every pooled
    value; the value itself is re-evaluated in exact fp32 at that point (pngpd_trunk_pool_refine — layers 1-2 for the
    B*1024 chosen points + one 128-long contraction each), so a pooled feature equals the fp32 path's bit for bit
    wherever the choice agrees.  Costs the arg tracking in the trunk's epilogue plus one gather-sized pass; off by
    default (eval-mode bf16x3 / bf16 already sit at 2e-7 / 6e-5 of the fp32 log-probs on the bench inputs)."""
    global _INFER_PRECISION, _INFER_REFINE
    if mode not in ("fp32", "bf16x3", "bf16"):
        raise ValueError("precision must be 'fp32', 'bf16x3' or 'bf16'")
    _INFER_PRECISION = mode
    _INFER_REFINE = bool(refine) and mode != "fp32"


def get_inference_precision():
    return _INFER_PRECISION


def _trunk_infer(mod, x, trans, relu_last):
    """Eval-mode fused trunk of a module holding conv1..3 / bn1..3 in the selected arithmetic."""
    if _INFER_PRECISION != "fp32" and _INFER_REFINE:
        pooled_bf, arg = ops.trunk_fwd_infer_bf(x, trans, *_trunk_infer_weights_x3(mod, x.device), relu_last=relu_last,
                                                nterms=_NTERMS[_INFER_PRECISION], want_arg=True)
        w1, b1, w2p, b2, _, b3 = _trunk_infer_weights(mod, x.device)
        w3f, ones128, ones1024 = _trunk_refine_weights(mod, x.device)
        zex = ops.trunk_pool_refine(x.float() if x.dtype == torch.bfloat16 else x, trans, w1, b1, None, None, w2p,
                                    ones128, b2, arg, w3=w3f, g3=ones1024, variant=1)
        pooled = zex + b3
        if relu_last:
            pooled = torch.where(pooled < 0, torch.zeros_like(pooled), pooled)
        # a non-finite input coordinate poisons its cloud's row in the bf16 pass (torch.max semantics): keep that
        return torch.where(torch.isnan(pooled_bf), pooled_bf, pooled)
    if _INFER_PRECISION != "fp32":
        return ops.trunk_fwd_infer_bf(x, trans, *_trunk_infer_weights_x3(mod, x.device), relu_last=relu_last,
                                      nterms=_NTERMS[_INFER_PRECISION])
    if x.dtype == torch.bfloat16:
        x = x.float()
    return ops.trunk_fwd_infer(x, trans, *_trunk_infer_weights(mod, x.device), relu_last=relu_last)


def _check_points(x, num_points, input_chann):
    if x.dim() != 3 or x.shape[1] != input_chann:
        raise RuntimeError(f"expected input of shape (B,{input_chann},N), got {tuple(x.shape)}")
    if x.shape[2] != num_points:
        # The reference's MaxPool1d(num_points)+view(-1,1024) silently corrupts the batch
        # dimension when N != num_points; fail instead.
        raise RuntimeError(f"N={x.shape[2]} points per cloud but the model was built with "
                           f"num_points={num_points}")


class _FoldCache:
    """Per-device cache of BN-folded / MFMA-packed inference weights, invalidated by the
    version counters of the tensors it was built from (optimizer steps, load_state_dict)."""

    def __init__(self):
        self.store = {}

    def get(self, key, tensors, build):
        sig = tuple((t.data_ptr(), t._version) for t in tensors)
        hit = self.store.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        val = build()
        self.store[key] = (sig, val)
        return val


class _HipModule(nn.Module):
    """nn.Module with a non-persistent fold cache (never pickled, never in state_dict)."""

    def _cache(self):
        c = self.__dict__.get("_fold_cache")
        if c is None:
            c = _FoldCache()
            self.__dict__["_fold_cache"] = c
        return c

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_fold_cache", None)
        return d


def _bn_tensors(bn):
    return [bn.weight, bn.bias, bn.running_mean, bn.running_var]


def _fold(layer, bn, layout, device):
    if bn is None:
        return ops.fold_conv_bn(layer.weight, layer.bias, layout=layout)
    return ops.fold_conv_bn(layer.weight, layer.bias, bn.weight, bn.bias, bn.running_mean,
                            bn.running_var, eps=bn.eps, layout=layout)


def _trunk_infer_weights(mod, device):
    """Folded weights of a conv1..3 / bn1..3 trunk, in the layouts pngpd_trunk_fwd_infer wants."""
    srcs = []
    for i in (1, 2, 3):
        conv, bn = getattr(mod, f"conv{i}"), getattr(mod, f"bn{i}")
        srcs += [conv.weight, conv.bias] + _bn_tensors(bn)

    def build():
        w1, b1 = _fold(mod.conv1, mod.bn1, ops.LAYOUT_ROWMAJOR, device)
        w2, b2 = _fold(mod.conv2, mod.bn2, ops.LAYOUT_MFMA_B, device)
        w3, b3 = _fold(mod.conv3, mod.bn3, ops.LAYOUT_MFMA_B, device)
        return (w1, b1, w2, b2, w3, b3)

    return mod._cache().get(("trunk", device), srcs, build)


def _trunk_infer_weights_x3(mod, device):
    """Folded weights for pngpd_trunk_fwd_infer_bf: layer 1 fp32, layers 2/3 split into bf16 hi/lo."""
    srcs = []
    for i in (1, 2, 3):
        conv, bn = getattr(mod, f"conv{i}"), getattr(mod, f"bn{i}")
        srcs += [conv.weight, conv.bias] + _bn_tensors(bn)

    def build():
        w1, b1 = _fold(mod.conv1, mod.bn1, ops.LAYOUT_ROWMAJOR, device)
        w2, b2 = _fold(mod.conv2, mod.bn2, ops.LAYOUT_ROWMAJOR, device)
        w3, b3 = _fold(mod.conv3, mod.bn3, ops.LAYOUT_ROWMAJOR, device)
        return (w1, b1, ops.split_pack_bf16(w2), b2, ops.split_pack_bf16(w3), b3)

    return mod._cache().get(("trunk_x3", device), srcs, build)


def _trunk_refine_weights(mod, device):
    """Operands of the eval-mode pool refinement that no other path keeps: the BN-folded conv3 weight row-major, and the
    unit vectors standing in for the train-mode layer-2 scale / the sign of gamma3 (both folded into the weights)."""
    srcs = [mod.conv3.weight, mod.conv3.bias] + _bn_tensors(mod.bn3)

    def build():
        w3, _ = _fold(mod.conv3, mod.bn3, ops.LAYOUT_ROWMAJOR, device)
        return (w3, torch.ones(128, device=device), torch.ones(1024, device=device))

    return mod._cache().get(("trunk_refine", device), srcs, build)


def _fc_infer_weights(mod, names, device):
    """names: [(linear_attr, bn_attr or None), ...] -> [(Wf, bf), ...]."""
    srcs = []
    for lin, bn in names:
        l = getattr(mod, lin)
        srcs += [l.weight, l.bias] + (_bn_tensors(getattr(mod, bn)) if bn else [])

    def build():
        return [_fold(getattr(mod, lin), getattr(mod, bn) if bn else None, ops.LAYOUT_ROWMAJOR, device)
                for lin, bn in names]

    return mod._cache().get(("fc",) + tuple(n for n, _ in names) + (device,), srcs, build)


def _trunk_aten(mod, x, relu_last):
    """The ATen composite of a trunk (CPU plumbing path)."""
    x = F.relu(mod.bn1(mod.conv1(x)))
    x = F.relu(mod.bn2(mod.conv2(x)))
    x = mod.bn3(mod.conv3(x))
    if relu_last:
        x = F.relu(x)
    return torch.max(x, dim=2)[0]


def _check_train_batch(x):
    if x.shape[0] < 2:
        # same condition nn.BatchNorm1d enforces on the (B,C) FC activations in train mode
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")


# ---------------------------------------------------------------------------------------
# hot-path classes
# ---------------------------------------------------------------------------------------
class STN3d(_HipModule):
    """Input transform net (reference pointnet.py:8-45)."""

    def __init__(self, num_points=2500, input_chann=3):
        super().__init__()
        self.num_points = num_points
        widths = (input_chann,) + _TRUNK_WIDTHS
        for i in range(3):
            setattr(self, f"conv{i + 1}", nn.Conv1d(widths[i], widths[i + 1], 1))
        self.mp1 = nn.MaxPool1d(num_points)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, 9)
        self.relu = nn.ReLU()
        for i, c in enumerate(_TRUNK_WIDTHS + (512, 256)):
            setattr(self, f"bn{i + 1}", nn.BatchNorm1d(c))

    def forward(self, x):
        _check_points(x, self.num_points, self.conv1.in_channels)
        if x.is_cuda:
            if self.training:
                return self._forward_hip_train(x)
            return self._forward_hip_infer(x)
        g = _trunk_aten(self, x, relu_last=True)
        g = F.relu(self.bn4(self.fc1(g)))
        g = F.relu(self.bn5(self.fc2(g)))
        g = self.fc3(g)
        return (g + torch.eye(3, dtype=g.dtype, device=g.device).reshape(1, 9)).view(-1, 3, 3)

    def _forward_hip_train(self, x):
        _check_train_batch(x)
        pooled = train.trunk_train(self, x.contiguous(), None, relu_last=True)   # x already fp32 (PointNetfeat)
        return train.head_train(self.fc1, self.bn4, self.fc2, self.bn5, self.fc3, pooled,
                                ops.EPI_ADD_IDEN3).view(-1, 3, 3)

    def _forward_hip_infer(self, x):
        dev = x.device
        x = x.contiguous()
        pooled = _trunk_infer(self, x, None, relu_last=True)
        (w1, b1), (w2, b2), (w3, b3) = _fc_infer_weights(self, [("fc1", "bn4"), ("fc2", "bn5"), ("fc3", None)], dev)
        g = ops.fc_fwd(pooled, w1, b1, ops.EPI_RELU)
        g = ops.fc_fwd(g, w2, b2, ops.EPI_RELU)
        return ops.fc_fwd(g, w3, b3, ops.EPI_ADD_IDEN3).view(-1, 3, 3)


class PointNetfeat(_HipModule):
    """Transform + per-point MLP + global max feature (reference pointnet.py:123-154)."""

    def __init__(self, num_points=2500, input_chann=3, global_feat=True):
        super().__init__()
        self.stn = STN3d(num_points=num_points, input_chann=input_chann)
        widths = (input_chann,) + _TRUNK_WIDTHS
        for i in range(3):
            setattr(self, f"conv{i + 1}", nn.Conv1d(widths[i], widths[i + 1], 1))
        for i, c in enumerate(_TRUNK_WIDTHS):
            setattr(self, f"bn{i + 1}", nn.BatchNorm1d(c))
        self.mp1 = nn.MaxPool1d(num_points)
        self.num_points = num_points
        self.global_feat = global_feat

    def forward(self, x):
        _check_points(x, self.num_points, self.conv1.in_channels)
        if x.is_cuda and self.global_feat:
            x = x.contiguous()
            if self.training and x.dtype == torch.bfloat16:
                x = x.float()     # bf16 cloud storage: the training passes read fp32 — one widening cast per step
            trans = self.stn(x)
            if self.training:
                return train.trunk_train(self, x, trans.contiguous(), relu_last=False), trans
            return _trunk_infer(self, x, trans.contiguous(), relu_last=False), trans
        # ATen composite: CPU plumbing path, and the (never used) global_feat=False branch.
        trans = self.stn(x)
        x = torch.bmm(x.transpose(2, 1), trans).transpose(2, 1)
        x = F.relu(self.bn1(self.conv1(x)))
        pointfeat = x
        x = F.relu(self.bn2(self.conv2(x)))
        x = self.bn3(self.conv3(x))
        g = torch.max(x, dim=2)[0]
        if self.global_feat:
            return g, trans
        g = g.view(-1, 1024, 1).repeat(1, 1, self.num_points)
        return torch.cat([g, pointfeat], 1), trans


class PointNetCls(_HipModule):
    """Grasp classifier (reference pointnet.py:177-194): log-probabilities + the 3x3 transform."""

    def __init__(self, num_points=2500, input_chann=3, k=2):
        super().__init__()
        self.num_points = num_points
        self.feat = PointNetfeat(num_points, input_chann=input_chann, global_feat=True)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, k)
        self.bn1 = nn.BatchNorm1d(512)
        self.bn2 = nn.BatchNorm1d(256)
        self.relu = nn.ReLU()

    def forward(self, x):
        g, trans = self.feat(x)
        if g.is_cuda:
            if self.training:
                return train.head_train(self.fc1, self.bn1, self.fc2, self.bn2, self.fc3, g,
                                        ops.EPI_LOG_SOFTMAX), trans
            (w1, b1), (w2, b2), (w3, b3) = _fc_infer_weights(self, [("fc1", "bn1"), ("fc2", "bn2"), ("fc3", None)],
                                                             g.device)
            g = ops.fc_fwd(g, w1, b1, ops.EPI_RELU)
            g = ops.fc_fwd(g, w2, b2, ops.EPI_RELU)
            return ops.fc_fwd(g, w3, b3, ops.EPI_LOG_SOFTMAX), trans
        g = F.relu(self.bn1(self.fc1(g)))
        g = F.relu(self.bn2(self.fc2(g)))
        return F.log_softmax(self.fc3(g), dim=-1), trans

    def forward_loss(self, x, target, reduction="mean"):
        """``output, trans = model(x); loss = F.nll_loss(output, target, reduction=reduction)`` (main_1v.py:73-74) as
        one call -> (loss, output, trans).  On the HIP training path the loss and its backward run inside the FC head's
        own foreign calls (``pngpd_head_train_fwd/_bwd`` with ``target``): no ATen kernel is left in a training step.
        Everywhere else (CPU, eval) it is literally the two reference lines."""
        if x.is_cuda and self.training:
            g, trans = self.feat(x)
            output, loss = train.head_train(self.fc1, self.bn1, self.fc2, self.bn2, self.fc3, g, ops.EPI_LOG_SOFTMAX,
                                            target=target, reduction=reduction)
            return loss, output, trans
        output, trans = self(x)
        return F.nll_loss(output, target, reduction=reduction), output, trans


# ---------------------------------------------------------------------------------------
# import-compatibility classes (plain ATen; out of the hot path, see module docstring)
# ---------------------------------------------------------------------------------------
class SimpleSTN3d(nn.Module):
    """Small T-Net (reference pointnet.py:48-85); never instantiated by the reference scripts
    except through DualPointNetfeat."""

    def __init__(self, num_points=2500, input_chann=3):
        super().__init__()
        self.num_points = num_points
        widths = (input_chann, 64, 128, 256)
        for i in range(3):
            setattr(self, f"conv{i + 1}", nn.Conv1d(widths[i], widths[i + 1], 1))
        self.mp1 = nn.MaxPool1d(num_points)
        self.fc1 = nn.Linear(256, 128)
        self.fc2 = nn.Linear(128, 64)
        self.fc3 = nn.Linear(64, 9)
        self.relu = nn.ReLU()
        for i, c in enumerate((64, 128, 256, 128, 64)):
            setattr(self, f"bn{i + 1}", nn.BatchNorm1d(c))

    def forward(self, x):
        g = _trunk_aten(self, x, relu_last=True)
        g = F.relu(self.bn4(self.fc1(g)))
        g = F.relu(self.bn5(self.fc2(g)))
        g = self.fc3(g)
        return (g + torch.eye(3, dtype=g.dtype, device=g.device).reshape(1, 9)).view(-1, 3, 3)


class DualPointNetfeat(nn.Module):
    """Two-cloud feature extractor (reference pointnet.py:88-120)."""

    def __init__(self, num_points=2500, input_chann=6, global_feat=True):
        super().__init__()
        self.stn1 = SimpleSTN3d(num_points=num_points, input_chann=input_chann // 2)
        self.stn2 = SimpleSTN3d(num_points=num_points, input_chann=input_chann // 2)
        widths = (input_chann,) + _TRUNK_WIDTHS
        for i in range(3):
            setattr(self, f"conv{i + 1}", nn.Conv1d(widths[i], widths[i + 1], 1))
        for i, c in enumerate(_TRUNK_WIDTHS):
            setattr(self, f"bn{i + 1}", nn.BatchNorm1d(c))
        self.mp1 = nn.MaxPool1d(num_points)
        self.num_points = num_points
        self.global_feat = global_feat

    def forward(self, x):
        t1, t2 = self.stn1(x[:, 0:3, :]), self.stn2(x[:, 3:6, :])
        xt = x.transpose(2, 1)
        x = torch.cat([torch.bmm(xt[..., 0:3], t1), torch.bmm(xt[..., 3:6], t2)], dim=-1).transpose(2, 1)
        x = F.relu(self.bn1(self.conv1(x)))
        pointfeat = x
        x = F.relu(self.bn2(self.conv2(x)))
        g = torch.max(self.bn3(self.conv3(x)), dim=2)[0]
        if self.global_feat:
            return g, t1 + t2
        g = g.view(-1, 1024, 1).repeat(1, 1, self.num_points)
        return torch.cat([g, pointfeat], 1), t1 + t2


class DualPointNetCls(nn.Module):
    """Two-cloud classifier (reference pointnet.py:157-174)."""

    def __init__(self, num_points=2500, input_chann=3, k=2):
        super().__init__()
        self.num_points = num_points
        self.feat = DualPointNetfeat(num_points, input_chann=input_chann, global_feat=True)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, k)
        self.bn1 = nn.BatchNorm1d(512)
        self.bn2 = nn.BatchNorm1d(256)
        self.relu = nn.ReLU()

    def forward(self, x):
        g, trans = self.feat(x)
        g = F.relu(self.bn1(self.fc1(g)))
        g = F.relu(self.bn2(self.fc2(g)))
        return F.log_softmax(self.fc3(g), dim=-1), trans


class PointNetDenseCls(nn.Module):
    """Per-point segmentation head (reference pointnet.py:197-221)."""

    def __init__(self, num_points=2500, input_chann=3, k=2):
        super().__init__()
        self.num_points = num_points
        self.k = k
        self.feat = PointNetfeat(num_points, input_chann=input_chann, global_feat=False)
        widths = (1088, 512, 256, 128, k)
        for i in range(4):
            setattr(self, f"conv{i + 1}", nn.Conv1d(widths[i], widths[i + 1], 1))
        for i, c in enumerate((512, 256, 128)):
            setattr(self, f"bn{i + 1}", nn.BatchNorm1d(c))

    def forward(self, x):
        b = x.size(0)
        x, trans = self.feat(x)
        for i in (1, 2, 3):
            x = F.relu(getattr(self, f"bn{i}")(getattr(self, f"conv{i}")(x)))
        x = self.conv4(x).transpose(2, 1).contiguous()
        x = F.log_softmax(x.view(-1, self.k), dim=-1)
        return x.view(b, self.num_points, self.k), trans
