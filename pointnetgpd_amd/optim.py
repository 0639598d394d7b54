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
import torch

from . import _lib


class FlatAdam(torch.optim.Optimizer):
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
        # 256-byte aligned slices (the kernels write gradients with 16-byte stores)
        self.offsets, off = [], 0
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
        with torch.no_grad():
            for p, o in zip(ps, self.offsets):
                n = p.numel()
                self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                gv = self.flat_g[o:o + n].view(p.shape)
                p.grad = gv
                p._pngpd_grad = gv
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
        """The fused backward OVERWRITES every gradient slice, so nothing is cleared (and the views are never
        dropped); with pass-by-pass / ATen sequencing autograd accumulates into the views, which then need zeros."""
        from . import train
        if train._use_fused():
            return
        for p in self._params:
            p.grad = p._pngpd_grad
        self.flat_g.zero_()

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
        self._step += 1
        with _lib.device_guard(self.device):
            if self.step_dev is not None:
                _lib.check(lib.pngpd_adam_step_inc(self.step_dev.data_ptr(), stream), "adam_step_inc")
            _lib.check(lib.pngpd_adam_flat(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                                           self.flat_v.data_ptr(), self.numel, lr_host, lr_dev, float(b1), float(b2),
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
        for p in self._params:
            self.state[p]["step"] = torch.tensor(float(self._step))
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
                if k != "params":
                    g[k] = v
