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
import ctypes
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, arith, ops, train

_TRUNK_WIDTHS = (64, 128, 1024)

# Inference arithmetic of the fused trunk (per model: ``model.set_precision(...)``, arith.py):
#   "fp32"   exact fp32 MFMA (default; the only mode that meets the 1e-3 log-prob contract by construction)
#   "bf16x3" opt-in: 3-term split-bf16 products on the bf16 matrix cores (~2^-16 relative product error; log-probs
#            within 1e-4 of fp32, not bit-identical)
#   "bf16"   opt-in: plain single-product bf16 operands, fp32 accumulate (BASELINE configs[2]); ~2^-8 per product —
#            does NOT meet 1e-3 in general, measured bounds in tests/test_gpu_bf16.py and DESIGN.md
# In the two bf16 modes the clouds may also be STORED as bf16 ((B,3,N) torch.bfloat16, 6 B/point): the trunk kernel
# reads them directly.  In "fp32" mode a bf16 cloud is widened first.
_NTERMS = {"bf16x3": 3, "bf16": 1}


def set_inference_precision(mode, refine=False):
    """Deprecated shim (rounds 1-5): sets the PROCESS DEFAULT that models without their own ``set_precision`` use.
    ``refine`` (reduced-precision modes only): the bf16 / bf16x3 trunk only CHOOSES the arg-max point of every pooled
    value; the value itself is re-evaluated in exact fp32 at that point (pngpd_trunk_pool_refine — layers 1-2 for the
    B*1024 chosen points + one 128-long contraction each), so a pooled feature equals the fp32 path's bit for bit
    wherever the choice agrees.  Costs the arg tracking in the trunk's epilogue plus one gather-sized pass; off by
    default (eval-mode bf16x3 / bf16 already sit at 2e-7 / 6e-5 of the fp32 log-probs on the bench inputs)."""
    arith.set_default(infer=mode, infer_refine=bool(refine) and mode != "fp32")


def get_inference_precision():
    return arith.default("infer")


def _trunk_infer(mod, x, trans, relu_last, folded=None):
    """Eval-mode fused trunk of a module holding conv1..3 / bn1..3 in the module's own arithmetic; ``folded`` = the
    module's entry of the active fold plan (None: fold now — kernel-level tests calling a trunk on its own)."""
    if folded is None:
        with _folded(mod, x.device) as w:
            return _trunk_infer(mod, x, trans, relu_last, w)
    cfg = arith.resolve(mod)
    l1, l2, l3 = folded["trunk"]
    ones = folded.get("ones")
    if cfg.infer != "fp32" and cfg.infer_refine:
        pooled_bf, arg = ops.trunk_fwd_infer_bf(x, trans, l1["row"], l1["bf"], l2["x3"], l2["bf"], l3["x3"], l3["bf"],
                                                relu_last=relu_last, nterms=_NTERMS[cfg.infer], want_arg=True)
        if ones is None:
            # unit vectors standing in for the train-mode layer-2 scale / the sign of gamma3 (both folded into the weights)
            ones = folded["ones"] = (torch.ones(128, device=x.device), torch.ones(1024, device=x.device))
        zex = ops.trunk_pool_refine(x.float() if x.dtype == torch.bfloat16 else x, trans, l1["row"], l1["bf"], None,
                                    None, l2["mfma"], ones[0], l2["bf"], arg, w3=l3["row"], g3=ones[1], w3sp=l3["mfma"],
                                    variant=1)      # w3sp given: layers 1-2 at the DISTINCT arg-max points only
        pooled = zex + l3["bf"]
        if relu_last:
            pooled = torch.where(pooled < 0, torch.zeros_like(pooled), pooled)
        # a non-finite input coordinate poisons its cloud's row in the bf16 pass (torch.max semantics): keep that
        return torch.where(torch.isnan(pooled_bf), pooled_bf, pooled)
    if cfg.infer != "fp32":
        return ops.trunk_fwd_infer_bf(x, trans, l1["row"], l1["bf"], l2["x3"], l2["bf"], l3["x3"], l3["bf"],
                                      relu_last=relu_last, nterms=_NTERMS[cfg.infer])
    if x.dtype == torch.bfloat16:
        x = x.float()
    return ops.trunk_fwd_infer(x, trans, l1["row"], l1["bf"], l2["mfma"], l2["bf"], l3["mfma"], l3["bf"],
                               relu_last=relu_last)


def _check_points(x, num_points, input_chann):
    if x.dim() != 3 or x.shape[1] != input_chann:
        raise RuntimeError(f"expected input of shape (B,{input_chann},N), got {tuple(x.shape)}")
    if x.shape[2] != num_points:
        # The reference's MaxPool1d(num_points)+view(-1,1024) silently corrupts the batch
        # dimension when N != num_points; fail instead.
        raise RuntimeError(f"N={x.shape[2]} points per cloud but the model was built with "
                           f"num_points={num_points}")


# ---------------------------------------------------------------------------------------
# eval-mode weights: folded from the LIVE parameters on every forward, one launch per model
# ---------------------------------------------------------------------------------------
_TLS = threading.local()
_TRUNK_LAYERS = (("conv1", "bn1"), ("conv2", "bn2"), ("conv3", "bn3"))


def _layer_sources(mod, lin, bn):
    l = getattr(mod, lin)
    if bn is None:
        return (l.weight, l.bias, None, None, None, None, 0.0)
    n = getattr(mod, bn)
    return (l.weight, l.bias, n.weight, n.bias, n.running_mean, n.running_var, float(n.eps))


class _FoldPlan:
    """Device buffers + the C descriptor (``pngpd_fold_model_t``) of the eval-mode weights of a module tree.

    The reference reads ``self.convX.weight`` / ``self.bnX.running_mean`` in every forward (pointnet.py:29-31,35-37,
    144-147,191-193); so does this: ``launch()`` re-derives every folded / packed weight of the tree from the live
    tensors in ONE kernel launch (a few microseconds for the 1.6 M weights of a PointNetCls, no host synchronisation,
    capturable in a HIP graph).  Nothing is keyed by tensor version counters — an in-place edit through ``.data``
    (``m.weight.data.normal_()``, ``vector_to_parameters``) does not bump them — so there is no stale-weight state;
    the plan itself (buffers + descriptor) is rebuilt only when a source tensor's ADDRESS or a member's arithmetic
    changes (``signature``)."""

    def __init__(self, root, device, want=None):
        self.device = device
        self.members = {}
        layers = []                      # (sources, C, K, outputs dict)
        for m in root.modules():
            if not isinstance(m, _HipModule):
                continue
            cfg = arith.resolve(m)
            entry = {}
            if hasattr(m, "conv3"):
                w = want if want is not None else _trunk_wants(cfg)
                outs = []
                for i, (lin, bn) in enumerate(_TRUNK_LAYERS):
                    o = {k: None for k in ("row", "mfma", "x3")}
                    for k in w[i]:
                        o[k] = True
                    outs.append(self._add(layers, m, lin, bn, o))
                entry["trunk"] = outs
            fcs = _FC_STACKS.get(type(m).__name__)
            if fcs and want is None:
                entry["fc"] = [self._add(layers, m, lin, bn, {"row": True, "mfma": None, "x3": None}) for lin, bn in fcs]
            if entry:
                self.members[id(m)] = entry
        if len(layers) > _lib.FOLD_MAX_LAYERS:
            raise RuntimeError("fold plan: more layers than PNGPD_FOLD_MAX_LAYERS")
        self.args = _lib.FoldModel()
        self.args.n = len(layers)
        self._keep = layers
        for i, (src, C, K, o) in enumerate(layers):
            L = self.args.layer[i]
            W, b, g, be, rm, rv, eps = src
            L.W, L.b = W.data_ptr(), b.data_ptr() if b is not None else None
            if g is not None:
                L.gamma, L.beta, L.mean, L.var = g.data_ptr(), be.data_ptr(), rm.data_ptr(), rv.data_ptr()
            L.eps, L.C, L.K = eps, C, K
            L.row = o["row"].data_ptr() if o["row"] is not None else None
            L.mfma = o["mfma"].data_ptr() if o["mfma"] is not None else None
            L.x3 = o["x3"].data_ptr() if o["x3"] is not None else None
            L.bf = o["bf"].data_ptr()
        self.signature = _plan_signature(root, device, want)

    def _add(self, layers, m, lin, bn, o):
        src = _layer_sources(m, lin, bn)
        W = src[0]
        for t in src[:6]:
            if t is not None and (t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous()):
                raise RuntimeError("eval-mode HIP forward: parameters and BatchNorm buffers must be contiguous float32 "
                                   f"tensors on {self.device} (got {t.dtype} on {t.device})")
        C, K = W.shape[0], W[0].numel()
        out = {"bf": torch.empty(C, device=self.device)}
        out["row"] = torch.empty(C, K, device=self.device) if o["row"] else None
        out["mfma"] = torch.empty(C * K, device=self.device) if o["mfma"] else None
        out["x3"] = torch.empty(2 * C * K, device=self.device, dtype=torch.int16) if o["x3"] else None
        layers.append((src, C, K, out))
        return out

    def launch(self):
        fn = _FOLD_FN.get("f")
        if fn is None:
            fn = _FOLD_FN["f"] = _lib.load().pngpd_fold_model
        with _lib.device_guard(self.device):
            _lib.check(fn(ctypes.addressof(self.args), torch.cuda.current_stream(self.device).cuda_stream),
                       "pngpd_fold_model")


_FOLD_FN = {}
_FC_STACKS = {"STN3d": (("fc1", "bn4"), ("fc2", "bn5"), ("fc3", None)),
              "PointNetCls": (("fc1", "bn1"), ("fc2", "bn2"), ("fc3", None))}


def _trunk_wants(cfg):
    """Which layouts of conv1..3 the module's eval arithmetic reads (per layer)."""
    if cfg.infer == "fp32":
        return (("row",), ("mfma",), ("mfma",))
    if cfg.infer_refine:
        return (("row",), ("mfma", "x3"), ("row", "mfma", "x3"))
    return (("row",), ("x3",), ("x3",))


def _plan_signature(root, device, want):
    sig = [device, want]
    for m in root.modules():
        if not isinstance(m, _HipModule):
            continue
        own = m.__dict__.get("_arith")
        sig.append((arith.default("infer"), arith.default("infer_refine")) if not own else
                   (own.get("infer", arith.default("infer")), own.get("infer_refine", arith.default("infer_refine"))))
        if hasattr(m, "conv3"):
            for lin, bn in _TRUNK_LAYERS:
                sig += [t.data_ptr() if t is not None else 0 for t in _layer_sources(m, lin, bn)[:6]]
        fcs = _FC_STACKS.get(type(m).__name__)
        if fcs and want is None:
            for lin, bn in fcs:
                sig += [t.data_ptr() if t is not None else 0 for t in _layer_sources(m, lin, bn)[:6]]
    return tuple(sig)


class _folded:
    """``with _folded(module, device) as entry:`` — the module's freshly folded eval-mode weights.  The OUTERMOST HIP
    module of a forward (a PointNetCls; or a PointNetfeat / STN3d called on its own) folds its whole tree in one
    launch; nested modules find themselves in the active plan (thread-local: DataParallel-style threads each hold
    their own) and launch nothing."""

    def __init__(self, mod, device):
        self.mod, self.device = mod, device

    def __enter__(self):
        active = getattr(_TLS, "plan", None)
        self.owner = False
        if active is not None and active.device == self.device and id(self.mod) in active.members:
            return active.members[id(self.mod)]
        plans = self.mod.__dict__.get("_fold_plans")
        if plans is None:
            plans = self.mod.__dict__["_fold_plans"] = {}
        plan = plans.get(self.device)
        sig = _plan_signature(self.mod, self.device, None)
        if plan is None or plan.signature != sig:
            plan = plans[self.device] = _FoldPlan(self.mod, self.device)
        plan.launch()
        self.prev, self.owner = active, True
        _TLS.plan = plan
        return plan.members[id(self.mod)]

    def __exit__(self, *exc):
        if self.owner:
            _TLS.plan = self.prev
        return False


class _HipModule(nn.Module):
    """nn.Module whose eval-mode HIP weights are derived state (never pickled, never in state_dict)."""

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_fold_plans", None)
        d.pop("_fold_cache", None)        # rounds 1-5
        return d

    def set_precision(self, mode=None, *, infer=None, train=None, refine=None, fp32_side_passes=None,
                      refine_pool=None, sequencing=None):
        """Arithmetic of THIS model on the HIP path (and of its ``feat`` / ``stn`` sub-modules): ``mode`` sets the
        eval-mode and the train-mode trunks together ("fp32" exact, default | "bf16x3" | "bf16"); the keyword fields
        set one aspect each (arith.py).  Stored on the module as a dict of builtins: pickled with it, invisible to
        ``state_dict``, independent of every other model in the process.  Returns ``self``."""
        fields = dict(infer=infer if infer is not None else mode, train=train if train is not None else mode,
                      infer_refine=refine, fp32_side_passes=fp32_side_passes, refine_pool=refine_pool,
                      sequencing=sequencing)
        for m in self.modules():
            if isinstance(m, _HipModule):
                arith.set_on(m, **fields)
        return self

    def get_precision(self):
        """The effective arithmetic record (arith.Arith) of this module."""
        return arith.resolve(self)


def _trunk_tuple(outs, kinds):
    return tuple(v for o, k in zip(outs, kinds) for v in (o[k], o["bf"]))


def _trunk_infer_weights(mod, device):
    """(w1, b1, w2p, b2, w3p, b3): freshly folded weights of a conv1..3 / bn1..3 trunk in the layouts
    pngpd_trunk_fwd_infer wants (tools / kernel-level tests; the model's forward goes through ``_folded``)."""
    plan = _FoldPlan(mod, device, want=(("row",), ("mfma",), ("mfma",)))
    plan.launch()
    return _trunk_tuple(plan.members[id(mod)]["trunk"], ("row", "mfma", "mfma"))


def _trunk_infer_weights_x3(mod, device):
    """The same for pngpd_trunk_fwd_infer_bf: layer 1 fp32, layers 2/3 split into bf16 hi/lo fragments."""
    plan = _FoldPlan(mod, device, want=(("row",), ("x3",), ("x3",)))
    plan.launch()
    return _trunk_tuple(plan.members[id(mod)]["trunk"], ("row", "x3", "x3"))


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
                                ops.EPI_ADD_IDEN3, cfg=arith.resolve(self)).view(-1, 3, 3)

    def _forward_hip_infer(self, x):
        x = x.contiguous()
        with _folded(self, x.device) as w:
            pooled = _trunk_infer(self, x, None, True, w)
            f1, f2, f3 = w["fc"]
            g = ops.fc_fwd(pooled, f1["row"], f1["bf"], ops.EPI_RELU)
            g = ops.fc_fwd(g, f2["row"], f2["bf"], ops.EPI_RELU)
            return ops.fc_fwd(g, f3["row"], f3["bf"], ops.EPI_ADD_IDEN3).view(-1, 3, 3)


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
            if self.training:
                trans = self.stn(x)
                return train.trunk_train(self, x, trans.contiguous(), relu_last=False), trans
            with _folded(self, x.device) as w:
                trans = self.stn(x)
                return _trunk_infer(self, x, trans.contiguous(), False, w), trans
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
        if x.is_cuda and not self.training:
            with _folded(self, x.device) as w:        # one fold launch for the whole tree (feat, feat.stn, head)
                g, trans = self.feat(x)
                f1, f2, f3 = w["fc"]
                g = ops.fc_fwd(g, f1["row"], f1["bf"], ops.EPI_RELU)
                g = ops.fc_fwd(g, f2["row"], f2["bf"], ops.EPI_RELU)
                return ops.fc_fwd(g, f3["row"], f3["bf"], ops.EPI_LOG_SOFTMAX), trans
        g, trans = self.feat(x)
        if g.is_cuda:
            return train.head_train(self.fc1, self.bn1, self.fc2, self.bn2, self.fc3, g,
                                    ops.EPI_LOG_SOFTMAX, cfg=arith.resolve(self)), trans
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
                                            target=target, reduction=reduction, cfg=arith.resolve(self))
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
