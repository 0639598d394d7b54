"""``FlatAdam`` — ``torch.optim.Adam`` (the reference's optimizer, PointNetGPD/main_1v.py:61:
``optim.Adam(model.parameters(), lr=args.lr)`` with torch's defaults: betas (0.9, 0.999), eps 1e-8, no weight decay,
no amsgrad) over ONE flat HBM buffer.

The 44 parameters of a ``PointNetCls`` (1,604,363 values, 6.4 MB) are re-pointed at slices of a single buffer; their
gradients and the two Adam moments live in three more.  Consequences:

* ``step()`` is one launch (``pngpd_adam_flat``, 45 MB of streaming traffic) instead of torch's multi-tensor pair;
* the fused training entries (``train.FusedTrunkFn`` / ``FusedHeadFn``) write every parameter gradient straight
  into its slice of the gradient buffer — no per-parameter ``AccumulateGrad``, no ``zero_grad`` fills, and the exactly
  zero conv-bias gradients are written by the kernels;
* data-parallel training all-reduces slices of the SAME buffer (``ddp.GradAverager``): no gather / scatter copies,
  and the trunk gradients can leave while the rest of the backward still runs.

The update is torch's single-tensor Adam in fp32 (torch/optim/adam.py ``_single_tensor_adam``); per-parameter
``state`` entries (``step``, ``exp_avg``, ``exp_avg_sq``) are views of the flat moments, so ``state_dict()`` has the
layout of ``torch.optim.Adam``'s and checkpoints move between the two.
"""
import weakref

import torch
from torch.utils.weak import WeakIdKeyDictionary

from . import _lib

# parameter -> (gradient view, weakref to the owning FlatAdam, index in its parameter list).  Kept HERE and not as an
# attribute of the Parameter: instance attributes of a Parameter are pickled with it (a whole-module checkpoint would
# carry a second 6.4 MB buffer and come back with an orphaned view), and a registry entry dies with its optimizer.
_GRAD_VIEWS = WeakIdKeyDictionary()


def grad_view(param):
    """The flat-buffer gradient view the fused backward may write for ``param`` in place, or None: only while the
    owning FlatAdam is alive AND ``param.grad`` still IS that view (a ``zero_grad(set_to_none=True)`` of the module, a
    foreign optimizer, or a checkpoint round trip all break the identity and fall back to autograd's accumulation)."""
    e = _GRAD_VIEWS.get(param)
    if e is None:
        return None
    view, owner, _ = e
    if owner() is None or param.grad is not view:
        return None
    return view


class GradGroup:
    """The in-place gradient targets of ONE fused piece (a trunk's 12 parameters, an FC stack's 10): the flat
    optimizer's views in the piece's parameter order, resolved once and re-validated per step by identity checks only
    (the registry look-ups cost ~2 us per parameter — 0.1 ms per step at the reference's batch of 64, where the eager
    loop is host-bound)."""
    __slots__ = ("views", "params", "idx", "owner")

    def __init__(self, views, params, idx, owner):
        self.views, self.params, self.idx, self.owner = views, params, idx, owner

    def valid(self, params):
        if self.owner() is None or len(params) != len(self.params):
            return False
        for p, q, v in zip(params, self.params, self.views):
            if p is not q() or p.grad is not v:
                return False
        return True

    def mark_written(self):
        """Called by the fused backward after it has written the gradient slices of the piece in place."""
        opt = self.owner()
        if opt is None:
            return
        w = opt._written
        for i in self.idx:
            if w[i]:
                raise RuntimeError("FlatAdam: a gradient slice was written twice before step() — the fused backward "
                                   "OVERWRITES gradients (no accumulation over micro-batches or repeated module "
                                   "calls); use model.set_precision(sequencing='passes') or torch.optim.Adam for "
                                   "accumulation")
            w[i] = 1


_GROUPS = {}          # id(first parameter of the piece) -> GradGroup (validated by identity on every use)


def grad_group(params):
    """The GradGroup of a fused piece whose parameters ALL have live flat views (``grad_view``), None if none has one;
    a mix raises (the fused backward writes the whole piece's gradients in place)."""
    g = _GROUPS.get(id(params[0]))
    if g is not None and g.valid(params):
        return g
    entries = [_GRAD_VIEWS.get(p) for p in params]
    live = [e is not None and e[1]() is not None and p.grad is e[0] for e, p in zip(entries, params)]
    if not any(live):
        _GROUPS.pop(id(params[0]), None)
        return None
    if not all(live) or len({id(e[1]()) for e in entries}) != 1:
        raise RuntimeError("optim.FlatAdam must own all parameters of a trunk / FC stack or none of them "
                           "(the fused backward writes the whole piece's gradients in place)")
    g = GradGroup([e[0] for e in entries], [weakref.ref(p) for p in params], [e[2] for e in entries], entries[0][1])
    _GROUPS[id(params[0])] = g
    owner = entries[0][1]()
    for i in g.idx:
        owner._grouped[i] = 1     # these slices are OVERWRITTEN by a fused backward; all others accumulate via autograd
    return g


def ungroup(params):
    """Called by the NON-fused call paths (pass-by-pass sequencing, DEBUG_STASH) at forward time: their gradients
    arrive by autograd ACCUMULATION.  A slice a fused backward used to overwrite (``_grouped``) was exempt from
    ``zero_grad`` and may hold a stale gradient — zero it once and hand it back to the accumulate-and-clear regime."""
    for p in params:
        e = _GRAD_VIEWS.get(p)
        if e is None:
            continue
        view, owner, i = e
        opt = owner()
        if opt is not None and opt._grouped[i]:
            opt._grouped[i] = 0
            if not opt._written[i]:
                view.zero_()
    _GROUPS.pop(id(params[0]), None)


class FlatAdam(torch.optim.Optimizer):
    """Bias correction uses ONE step count for the whole buffer (``_step``; torch.optim.Adam keeps one per parameter).
    They agree whenever every parameter takes part in every step — the case for a PointNetCls.  A fused piece that sat
    out k steps (its slices are skipped, as torch skips ``grad is None``) resumes with the GLOBAL count in its bias
    correction where torch would use its own smaller count: after the first few hundred steps the two corrections are
    both within 1e-3 of 1, before that the resumed piece takes slightly smaller steps than torch's would."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        params = [p for p in params]
        if not params:
            raise ValueError("optimizer got an empty parameter list")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=float(eps)))
        ps = self.param_groups[0]["params"]
        dev = ps[0].device
        if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 for p in ps):
            raise RuntimeError("FlatAdam needs float32 parameters on one CUDA device (libpngpd has no CPU path)")
        self.device = dev
        # 256-byte aligned slices (the kernels write gradients with 16-byte stores) behind a 64-float header: slot 0 of
        # the GRADIENT buffer's header carries the kept-sample count of a data-parallel step, so that it travels in the
        # same all-reduce as the adjacent (first) gradient slice (ddp.GradAverager) — one collective fewer per step
        self.header = 64
        self.offsets, off = [], self.header
        for p in ps:
            self.offsets.append(off)
            off += (p.numel() + 63) // 64 * 64
        self.numel = off
        z = lambda: torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_p, self.flat_g, self.flat_m, self.flat_v = z(), z(), z(), z()
        self.capturable = bool(capturable)
        # device-resident step count / learning rate for captured graphs (StepLR rewrites group["lr"] in place)
        self.step_dev = torch.zeros((), device=dev, dtype=torch.float32) if capturable else None
        self._step = 0
        self._views = []
        self._written = bytearray(len(ps))    # per parameter: slice written in place by the fused backward this step
        self._grouped = bytearray(len(ps))    # per parameter: owned by a fused piece (optim.GradGroup) — set on first use
        with torch.no_grad():
            for p, o in zip(ps, self.offsets):
                n = p.numel()
                self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                gv = self.flat_g[o:o + n].view(p.shape)
                p.grad = gv
                _GRAD_VIEWS[p] = (gv, weakref.ref(self), len(self._views))
                self._views.append(gv)
                self.state[p] = {"step": torch.tensor(0.0),
                                 "exp_avg": self.flat_m[o:o + n].view(p.shape),
                                 "exp_avg_sq": self.flat_v[o:o + n].view(p.shape)}
        self._params = ps

    # ---- gradient buffer -------------------------------------------------------------------------------------
    def segment(self, params):
        """(offset, length) of the smallest contiguous range of the flat buffers covering ``params``."""
        idx = [i for i, p in enumerate(self._params) if any(p is q for q in params)]
        lo = min(self.offsets[i] for i in idx)
        hi = max(self.offsets[i] + (self._params[i].numel() + 63) // 64 * 64 for i in idx)
        return lo, hi - lo

    def zero_grad(self, set_to_none=True):
        """The fused backward OVERWRITES the gradient slices of the pieces it owns (``optim.GradGroup``: the trunks and
        FC stacks), so those are never cleared (and the views are never dropped).  Every other parameter — a plain
        autograd parameter next to the fused pieces, e.g. a custom head on a ``PointNetfeat`` — receives its gradient by
        autograd ACCUMULATION into the same view and is zeroed here; with pass-by-pass / ATen sequencing that is every
        parameter (one fill of the whole buffer).  A slice written twice raises (``mark_written``)."""
        for p, gv in zip(self._params, self._views):
            if p.grad is not gv:
                p.grad = gv
        self._written[:] = bytes(len(self._written))
        if any(self._grouped):
            # ``_grouped`` is set when a fused piece registers (optim.grad_group) and cleared when the same parameters
            # go through a non-fused call path again (optim.ungroup): no process-global sequencing flag is consulted
            for i, gv in enumerate(self._views):
                if not self._grouped[i]:
                    gv.zero_()
            return
        # no fused piece has registered yet (first step, or a model without one): clear everything
        self.flat_g.zero_()

    def _settle_gradients(self):
        """Before the update: every parameter's gradient must be IN its slice.  A gradient autograd left elsewhere
        (``.grad`` re-pointed or dropped by foreign code) is copied in / zeroed.  Returns the indices of the parameters
        to leave untouched this step: slices of a fused piece that took no part in this step's backward (an unused or
        eval-mode sub-module) hold a STALE gradient — ``torch.optim.Adam`` skips a parameter whose ``.grad`` is None,
        and so does this update (no momentum-only drift).  Plain autograd parameters always update."""
        skip = []
        for i, (p, gv) in enumerate(zip(self._params, self._views)):
            g = p.grad
            if g is not gv:
                if g is None:
                    gv.zero_()
                else:
                    gv.copy_(g)
                p.grad = gv
            elif self._grouped[i] and not self._written[i] and any(self._written):
                skip.append(i)
        self._written[:] = bytes(len(self._written))
        return skip

    def _update_ranges(self, skip):
        """(offset, length) runs of the flat buffers to update = everything but the slices of ``skip``."""
        if not skip:
            return [(self.header, self.numel - self.header)]
        runs, lo = [], self.header
        for i in skip:
            o = self.offsets[i]
            if o > lo:
                runs.append((lo, o - lo))
            lo = o + (self._params[i].numel() + 63) // 64 * 64
        if self.numel > lo:
            runs.append((lo, self.numel - lo))
        return runs

    # ---- update ----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, grad_div=None):
        """``grad_scale``: host factor on the gradients; ``grad_div``: 0-dim float32 device tensor the gradients are
        divided by (max(., 1)) — the all-reduced global sample count of ``ddp.GradAverager.backward``."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        lr, (b1, b2), eps = g["lr"], g["betas"], g["eps"]
        lib = _lib.load()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        lr_dev = lr.data_ptr() if torch.is_tensor(lr) and lr.is_cuda else None
        lr_host = 0.0 if lr_dev else float(lr)
        runs = self._update_ranges(self._settle_gradients())      # one run (one launch) unless a fused piece sat out
        self._step += 1
        with _lib.device_guard(self.device):
            if self.step_dev is not None:
                _lib.check(lib.pngpd_adam_step_inc(self.step_dev.data_ptr(), stream), "adam_step_inc")
            for lo, n in runs:
                h = lo * 4
                _lib.check(lib.pngpd_adam_flat(self.flat_p.data_ptr() + h, self.flat_g.data_ptr() + h,
                                               self.flat_m.data_ptr() + h, self.flat_v.data_ptr() + h,
                                               n, lr_host, lr_dev, float(b1), float(b2),
                                               float(eps), float(self._step),
                                               self.step_dev.data_ptr() if self.step_dev is not None else None,
                                               float(grad_scale), grad_div.data_ptr() if grad_div is not None else None,
                                               stream), "adam_flat")
        # the kernel rewrote every parameter behind autograd's back: version-keyed caches (the eval-mode fold cache)
        # must see it
        torch.autograd.graph.increment_version(self._params)
        return loss

    # ---- checkpoints (torch.optim.Adam layout) ------------------------------------------------------------------
    def state_dict(self):
        # a captured graph (train.GraphedTrainStep) advances only the device-resident counter: it is the truth then
        step = float(self.step_dev.item()) if self.step_dev is not None else float(self._step)
        self._step = int(step)
        for p in self._params:
            self.state[p]["step"] = torch.tensor(step)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        sd = state_dict["state"]
        steps = [float(v["step"]) for v in sd.values() if "step" in v]
        with torch.no_grad():
            for i, p in enumerate(self._params):
                st = sd.get(i)
                if st is None:
                    continue
                self.state[p]["exp_avg"].copy_(st["exp_avg"])
                self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
        self._step = int(max(steps)) if steps else 0
        if self.step_dev is not None:
            self.step_dev.fill_(float(self._step))
        for g, sg in zip(self.param_groups, state_dict["param_groups"]):
            for k, v in sg.items():
                if k == "params":
                    continue
                if k == "lr" and torch.is_tensor(g.get("lr")):
                    # a capturable optimizer's lr is a device tensor a captured graph reads: fill it, never replace it
                    g["lr"].fill_(float(v))
                else:
                    g[k] = v
