"""Data-parallel training helper: one process per GPU, gradients combined with RCCL all-reduces over xGMI
(backend "nccl" IS RCCL on ROCm; "gloo" for the CPU tests).

Replaces the reference's single-process ``nn.DataParallel(model, device_ids=[0,1,2,3])``
(PointNetGPD/main_1v.py:158-165, main_fullv.py:104-111):

* DataParallel re-broadcasts the 6.4 MB of parameters on every forward and reduces gradients onto device 0; here
  every rank owns a replica, parameters are broadcast once, and the 1,604,363 gradients travel in place.
* **Per-sample mean over the GLOBAL batch**, as DataParallel's gathered ``nll_loss`` gives: ranks back-propagate the
  SUM of their samples' losses, the all-reduce adds gradients and kept-sample counts, and the optimizer divides by the
  global count.  ``my_collate`` drops ``None`` samples independently on every rank, so per-rank batches are ragged; a
  rank left with fewer than two samples (train-mode BatchNorm needs two) contributes zeros and still joins every
  collective — nobody hangs.
* With ``optim.FlatAdam`` attached the gradients already live in one flat buffer: the all-reduce runs on slices of it
  (no gather / scatter copies) in TWO buckets — the PointNetCls head + PointNetfeat trunk slice leaves as soon as the
  backward has produced ``d loss / d trans`` (a tensor hook on the STN output), overlapping the STN backward; the
  STN slice (plus the sample count) follows at the end.
* BatchNorm statistics stay per replica, exactly as under DataParallel.  Train-mode forwards never
  read running statistics, so nothing is broadcast per step: ``sync_buffers()`` makes rank 0's running statistics
  everyone's before ``eval()`` / a checkpoint — the observable behaviour of DataParallel's replica-0-wins.
"""
import torch
import torch.distributed as dist


class GradAverager:
    def __init__(self, model, process_group=None, optimizer=None, early_bucket_at_world_1=False):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.opt = None
        self._pending = []
        self._hooked = False
        self._early = None          # (offset, length) of the bucket that can leave mid-backward
        self._late = None
        # a single-rank group has nothing to overlap, so the mid-backward bucket normally stays with the late one;
        # the flag sends it from the tensor hook anyway — the way to EXECUTE the multi-rank code path (async all-reduce
        # on a slice of the flat buffer, issued from inside autograd, ordered against the fused backward's stream) on
        # a one-GPU box (tests/test_gpu_rccl.py, ``torchrun --nproc-per-node 1 bench.py``)
        self.early_at_world_1 = bool(early_bucket_at_world_1)
        self.sync_parameters()
        if optimizer is not None:
            self.attach(optimizer)

    # ---- replicas ---------------------------------------------------------------------------------------------
    def sync_parameters(self):
        """Rank 0's parameters and buffers become everyone's (done once, and after a checkpoint load)."""
        with torch.no_grad():
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                dist.broadcast(t, src=0, group=self.group)
                # a collective rewrites the storage without touching the autograd version counter; the eval-mode
                # fold cache (model/pointnet.py) is keyed by (data_ptr, _version), so bump it explicitly
                torch.autograd.graph.increment_version(t)

    def sync_buffers(self):
        """Rank 0's BatchNorm buffers become everyone's: ONE broadcast of the 7,936 running statistics (flattened)
        and one of the 10 ``num_batches_tracked`` counters.  Call before ``eval()`` / saving, not per step."""
        if self.world == 1:
            return
        with torch.no_grad():
            bufs = list(self.model.buffers())
            for dt in sorted({b.dtype for b in bufs}, key=str):      # same order on every rank
                group = [b for b in bufs if b.dtype == dt]
                flat = torch.cat([b.reshape(-1) for b in group])
                dist.broadcast(flat, src=0, group=self.group)
                torch._foreach_copy_(group, [c.view_as(b) for c, b in zip(flat.split([b.numel() for b in group]), group)])
                torch.autograd.graph.increment_version(group)

    # ---- flat-buffer mode -------------------------------------------------------------------------------------
    def attach(self, optimizer):
        """Use ``optimizer``'s (optim.FlatAdam) flat gradient buffer for the collectives, in two buckets."""
        self.opt = optimizer
        self._count = optimizer.flat_g[:1]       # header slot 0 of the flat gradient buffer (optim.FlatAdam)
        stn = getattr(getattr(self.model, "feat", None), "stn", None)
        if stn is not None:
            lo, n = optimizer.segment(list(stn.parameters()))
            rest = [p for p in self.model.parameters() if not any(p is q for q in stn.parameters())]
            lo2, n2 = optimizer.segment(rest)
            if lo + n <= lo2 or lo2 + n2 <= lo:          # the two slices do not interleave
                self._late, self._early = (lo, n), (lo2, n2)
                if not self._hooked:
                    stn.register_forward_hook(self._stn_forward_hook)
                    self._hooked = True
        if self._early is None:
            self._late, self._early = (optimizer.header, optimizer.numel - optimizer.header), None
        # the kept-sample count sits in the header right in front of the first slice: whichever bucket starts there
        # takes the header along, so gradients and count share one all-reduce
        hdr = optimizer.header
        if self._late[0] == hdr:
            self._late = (0, self._late[1] + hdr)
            self._count_with = "late"
        elif self._early is not None and self._early[0] == hdr:
            self._early = (0, self._early[1] + hdr)
            self._count_with = "early"
        else:
            self._count_with = None

    def _stn_forward_hook(self, module, inputs, output):
        if (self.opt is not None and (self.world > 1 or self.early_at_world_1) and torch.is_tensor(output)
                and output.requires_grad):
            output.register_hook(self._early_ready)

    def _early_ready(self, grad):
        # d loss / d trans exists: the PointNetCls head and the PointNetfeat trunk have written their gradient slices.
        # Only inside backward(): a plain loss.backward() followed by average_gradients() must not see a bucket leave
        # early (it would be reduced twice, and unevenly across ranks).
        if self._active and self._early is not None and not self._skip and not self._early_sent:
            lo, n = self._early
            self._early_sent = True
            self._pending.append(dist.all_reduce(self.opt.flat_g[lo:lo + n], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))
        return None

    _skip = False
    _active = False
    _early_sent = False

    def backward(self, loss_sum, n_local):
        """Back-propagate this rank's SUMMED loss (``n_local`` kept samples; ``loss_sum`` None or n_local < 2: the rank
        sits this step out with zero gradients) and combine gradients across ranks.  Returns the global kept-sample
        count as a 0-dim device tensor for the optimizer to divide by (``FlatAdam.step(grad_div=count)``) when a flat
        buffer is attached; without one the parameters' ``.grad`` already hold the global per-sample mean and None is
        returned."""
        self._pending = []
        sit_out = loss_sum is None or n_local < 2
        if self.opt is not None:
            self._skip = False
            # the count is written BEFORE any bucket can leave (the early one goes from inside backward())
            if sit_out:
                self.opt.flat_g.zero_()      # header included: count 0
                self._skip = True            # no hook will fire: send the early bucket here
                if self._early is not None:
                    lo, n = self._early
                    self._pending.append(dist.all_reduce(self.opt.flat_g[lo:lo + n], op=dist.ReduceOp.SUM,
                                                         group=self.group, async_op=True))
            else:
                self._count.fill_(float(n_local))
                self._active, self._early_sent = True, False
                try:
                    if loss_sum.is_cuda:
                        from .train import loss_backward
                        loss_backward(loss_sum)       # cached unit root gradient: no ones_like fill per step
                    else:
                        loss_sum.backward()
                finally:
                    self._active = False
                if self._early is not None and not self._early_sent:
                    # the hook did not fire (no gradient flowed into the STN output): every rank reaches this point
                    # with the same graph, so the bucket is sent here on all of them
                    lo, n = self._early
                    self._pending.append(dist.all_reduce(self.opt.flat_g[lo:lo + n], op=dist.ReduceOp.SUM,
                                                         group=self.group, async_op=True))
            lo, n = self._late
            self._pending.append(dist.all_reduce(self.opt.flat_g[lo:lo + n], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))
            if self._count_with is None:     # (cannot happen with FlatAdam's layout; kept for foreign layouts)
                self._pending.append(dist.all_reduce(self._count, op=dist.ReduceOp.SUM, group=self.group,
                                                     async_op=True))
            for w in self._pending:
                w.wait()                     # stream-ordered for NCCL/RCCL: no host block
            self._pending = []
            return self._count[0]
        # generic parameters (CPU tests, foreign optimizers): one flat bucket built here, count in its last slot
        if sit_out:
            for p in self.params:
                p.grad = torch.zeros_like(p)
        else:
            loss_sum.backward()
        grads = []
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            grads.append(p.grad)
        cnt = torch.tensor([float(0 if sit_out else n_local)], device=grads[0].device, dtype=grads[0].dtype)
        flat = torch.cat([g.reshape(-1) for g in grads] + [cnt])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        total = flat[-1].clamp_min(1.0)
        flat[:-1].div_(total)
        torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat[:-1].split([g.numel() for g in grads]), grads)])
        return None                          # gradients are already the global per-sample mean

    def average_gradients(self):
        """All-reduce(sum)/world of every parameter gradient (equal per-rank batches; kept for callers that run
        ``loss.backward()`` themselves): in place on the flat buffer when one is attached, else through one bucket."""
        if self.world == 1:
            return
        if self.opt is not None:
            g = self.opt.flat_g[self.opt.header:]
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            g.div_(self.world)
            return
        grads = []
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            grads.append(p.grad)
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.div_(self.world)
        torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)])


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_*).  Returns (rank, world, local_rank); world == 1 -> nothing is initialised."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank
