"""Device-resident training data path ("optional GPU-side crop for training batches").

The reference feeds ``main_1v*.py`` through 32 DataLoader workers, each of which re-loads a grasp file and a cloud
file per sample and crops the cloud in numpy (``dataset.py:420-458``).  On an MI355X the whole dataset fits in a
corner of the 288 GB of HBM: every cloud file of every object is uploaded ONCE into a single fp64 arena, every item's
grasp frame (18 doubles) and label are computed ONCE into two per-dataset tables, and a batch is ONE foreign call,
``pngpd_train_batch``: [full-view: the per-sample gather lists] -> crop count/compact -> ``my_collate`` as an ordered
prefix over the keep flags -> resample (mode 0 = the training rule: without replacement iff m > N, ``None`` iff fewer
than 50 in-box points) written straight into the compacted rows.  Nothing but the epoch's permutation and view picks
(uploaded once per epoch) crosses PCIe.

The batches are produced ``prefetch`` batches AHEAD of the consumer on a side stream, so that the crop of batch t+1
runs under the training step of batch t (main_1v.py:59-84 consumes, :120-128 produces) and the only host<->device
hand-shake — the kept count B' the step's launches are sized by — is a pinned 4-byte copy that completed one step
earlier.  ``prefetch=0`` is the serial schedule (same stream, one host sync per batch); both schedules run the same
launches with the same keys, so they yield identical batches.

Per-sample semantics are those of ``PointGraspOneViewDataset.__getitem__`` / ``my_collate``:
view drawn uniformly from the object's NP3 clouds (:425-428, shuffle-then-last), training-style crop with the
object's mesh->cloud transform (:429-433), resample rule (:438-444), label rule (:447-453 / :536-541), samples that
come out ``None`` dropped from the batch (main_1v.py:48-50).  For the full-view datasets (``PointGraspDataset``,
:244-282) a sample's cloud is ``obj_points_num`` rows drawn with replacement from the stack of ``pc_file_used_num``
view files themselves drawn with replacement (:252-254); here that is a (B, obj_points_num) int32 gather list
built on the device (``pngpd_stack_gather_lists``: uniform rows of the stacked length).  The random streams differ
(numpy global RNG in forked workers there; one seeded numpy Generator per epoch + counter-hash device RNGs keyed by
(seed, epoch, position in the epoch) here — independent of the batch size).
"""
import collections
import ctypes

import numpy as np
import torch

from . import _lib, crop


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class _Slot:
    """Scratch of one in-flight batch (reused every ``prefetch + 1`` batches, always on the producing stream); the raw
    pointers are taken once.  ``pin``: the kept count, written by the device into pinned host memory."""

    def __init__(self, dev, B, max_keep, Pg):
        self.counts = torch.empty(B, device=dev, dtype=torch.int32)
        self.idx = torch.empty(B, max_keep, device=dev, dtype=torch.int32)
        self.rows = torch.empty(B, device=dev, dtype=torch.int32)
        self.valid = torch.empty(B, device=dev, dtype=torch.uint8)
        self.gather = torch.empty(B, Pg, device=dev, dtype=torch.int32) if Pg else None
        self.seg = torch.empty(4 * B, device=dev, dtype=torch.int32)      # segment counts of the two-launch scan
        self.pin = torch.zeros((), dtype=torch.int32).pin_memory()
        self.pin_np = self.pin.numpy()                       # the same 4 bytes, read without a tensor op
        self.event = torch.cuda.Event()
        self.counts_ptr, self.idx_ptr, self.rows_ptr = self.counts.data_ptr(), self.idx.data_ptr(), self.rows.data_ptr()
        self.valid_ptr, self.pin_ptr = self.valid.data_ptr(), self.pin.data_ptr()
        self.gather_ptr = self.gather.data_ptr() if Pg else None
        self.seg_ptr = self.seg.data_ptr()


class DeviceGraspLoader:
    segmented_scan = True       # long full-view sample clouds: 4 workgroups per sample, two launches (same batches)

    """Iterable of ``(data (B',3,N) fp32 CUDA, target (B',) int64 CUDA)`` batches over one of the four mirror
    datasets of ``model.dataset`` (one-view and full-view, 2- and 3-class).  ``len()`` = batches per epoch.
    ``last_meta`` holds, for the most recent batch, the item indices, the chosen view files, the in-box counts, the
    keep mask, the labels (-1 = the reference's ``None``) and, for full-view datasets, the gather lists.

    ``prefetch``: batches produced ahead on a side stream (0 = serial, on the consumer's stream).
    ``rank`` / ``world``: one process per GPU — each rank walks its strided share of the epoch's permutation
    (padded by wrap-around to equal lengths, like ``DistributedSampler``)."""

    def __init__(self, dataset, batch_size, device, shuffle=True, seed=0, max_keep=8192, prefetch=2, rank=0, world=1):
        if getattr(dataset, "projection", False):
            raise NotImplementedError("projection=True belongs to the GPD baseline")
        self.ds, self.B, self.device = dataset, int(batch_size), torch.device(device)
        self.shuffle, self.seed, self.max_keep, self.epoch = bool(shuffle), int(seed), int(max_keep), 0
        self.prefetch, self.rank, self.world = max(0, int(prefetch)), int(rank), max(1, int(world))
        if self.device.type != "cuda":
            raise RuntimeError("DeviceGraspLoader needs a CUDA device (the host path is model.dataset + DataLoader)")
        _lib.load()                                           # fail loudly here, not in the first batch
        chunks = self._index(dataset)
        self.arena = torch.from_numpy(np.concatenate(chunks, 0)).to(self.device)
        self.frames_all = torch.from_numpy(self._frames).to(self.device)          # (n_items,18) fp64
        self.labels_all = torch.from_numpy(self._labels).to(self.device)          # (n_items,) int64, -1 = None
        self._meta = None
        self._side = torch.cuda.Stream(device=self.device, priority=-1) if self.prefetch else None
        Pg = int(dataset.obj_points_num) if self.fullview else 0
        self._slots = [_Slot(self.device, self.B, self.max_keep, Pg) for _ in range(self.prefetch + 1)]

    def _index(self, dataset):
        """Host tables (no device work): arena layout, per-item grasp frames and labels, per-object view ranges."""
        self.ds = dataset
        self.fullview = not hasattr(dataset, "minimum_point_amount")
        self.view_range = {}                                  # path -> (start, len) in the arena
        chunks, off = [], 0
        self.files = []
        per = int(dataset.grasp_amount_per_file)
        n_obj = len(dataset.object)
        self._frames = np.empty((n_obj * per, 18))
        self._labels = np.empty(n_obj * per, dtype=np.int64)
        for oi, obj in enumerate(dataset.object):
            files = list(dataset.d_pc[dataset.transform[obj][0]])
            for path in files:
                if path in self.view_range:
                    continue
                pc = np.asarray(np.load(path), dtype=np.float64).reshape(-1, 3)
                self.view_range[path] = (off, len(pc))
                chunks.append(pc)
                off += len(pc)
            g = np.asarray(np.load(dataset.d_grasp[obj]), dtype=np.float64)[:per]
            if len(g) < per:
                # the reference indexes ``np.load(...)[grasp_ind]`` with grasp_ind < grasp_amount_per_file
                # (dataset.py:421-430) and raises IndexError on a short file; so does this loader, up front
                raise IndexError(f"{dataset.d_grasp[obj]}: {len(g)} grasps, grasp_amount_per_file = {per}")
            lab = [dataset._label(r[-2] + r[-1] * 0.01) for r in g]        # dataset.py:446-453 / :535-541, once
            self.files.append(files)
            self._frames[oi * per:oi * per + len(g)] = crop.frames_from_grasps_train(g, dataset.transform[obj][1])
            self._labels[oi * per:oi * per + len(g)] = [-1 if v is None else v for v in lab]
        self.nfiles = np.array([len(f) for f in self.files], dtype=np.int64)
        self.range_tab = np.zeros((n_obj, int(self.nfiles.max()), 2), dtype=np.int32)
        for oi, files in enumerate(self.files):
            self.range_tab[oi, :len(files)] = [self.view_range[f] for f in files]
        return chunks

    def __len__(self):
        return (self._per_rank() + self.B - 1) // self.B

    def _per_rank(self):
        return (len(self.ds) + self.world - 1) // self.world

    @property
    def dataset(self):
        return self.ds

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    # ------------------------------------------------------------------ per epoch (host, once)
    def _epoch_tables(self):
        ds = self.ds
        rng = np.random.default_rng([self.seed, self.epoch])
        order = rng.permutation(len(ds)) if self.shuffle else np.arange(len(ds))
        if self.world > 1:
            total = self._per_rank() * self.world
            order = np.concatenate([order, order[:total - len(order)]])[self.rank:total:self.world]
            rng = np.random.default_rng([self.seed, self.epoch, self.rank + 1])
        order = np.ascontiguousarray(order, dtype=np.int32)
        obj = order // int(ds.grasp_amount_per_file)
        if self.fullview:                                     # k views WITH replacement (:252-253)
            pick = rng.integers(0, self.nfiles[obj][:, None], size=(len(order), int(ds.pc_file_used_num)))
            spans = self.range_tab[obj[:, None], pick]        # (n,k,2)
        else:                                                 # uniform view (:425-428, shuffle-then-last)
            pick = rng.integers(0, self.nfiles[obj])
            spans = self.range_tab[obj, pick]                 # (n,2)
        return order, obj, pick, np.ascontiguousarray(spans, dtype=np.int32)

    def _device_seed(self):
        return ((self.seed * 1000003 + self.epoch) * 100003 + self.rank) & (2 ** 64 - 1)

    # ------------------------------------------------------------------ one batch = one foreign call
    # The eager step at the reference's batch of 64 is host-bound (0.75 ms of Python + ctypes per step), so every
    # microsecond of host work per batch is end-to-end time: the call below is raw pointers computed once per epoch, the
    # outputs of _CHUNK batches come from one allocation, and the kept count is written by the kernel STRAIGHT into pinned
    # host memory (device-visible on this platform) — no copy launch, one event per batch.
    _CHUNK = 8

    def _enqueue(self, bi, slot, ep, stream):
        s = bi * self.B
        G = min(ep["n"], s + self.B) - s
        ci = bi % self._CHUNK
        if ci == 0 or ep.get("chunk") is None:
            with torch.cuda.stream(stream):
                oc = torch.empty(self._CHUNK, self.B, 3, ep["N"], device=self.device, dtype=torch.float32)
                lc = torch.empty(self._CHUNK, self.B, device=self.device, dtype=torch.int64)
            if self.prefetch:                                 # consumed on the caller's stream: once per allocation
                oc.record_stream(ep["main"]); lc.record_stream(ep["main"])
            ep["chunk"] = (oc.unbind(0), lc.unbind(0))
        out, labels_out = ep["chunk"][0][ci], ep["chunk"][1][ci]
        args = (ep["arena"], ep["f64"], ep["P"], ep["frames"], ep["labels"], ep["order"] + 4 * s,
                ep["spans"] + ep["span_stride"] * s, ep["k"], ep["Pg"], slot.gather_ptr, G, self.max_keep, ep["N"],
                ep["min_pts"], ep["seed"], s, slot.counts_ptr, slot.idx_ptr, slot.rows_ptr, slot.valid_ptr,
                slot.seg_ptr if self.segmented_scan else None,
                out.data_ptr(), labels_out.data_ptr(), slot.pin_ptr, stream.cuda_stream)
        if ep["guard"]:                                       # another device is current: the slow, guarded launch
            with torch.cuda.device(self.device):
                code = ep["fn"](*args)
        else:
            code = ep["fn"](*args)
        if code != 0:
            _lib.check(code, "train_batch")
        slot.event.record(stream)
        return dict(bi=bi, s=s, G=G, out=out, labels=labels_out, slot=slot)

    def __iter__(self):
        order, obj, pick, spans = self._epoch_tables()
        n = len(order)
        main = torch.cuda.current_stream(self.device)
        stream = self._side if self.prefetch else main
        if self.prefetch:
            stream.wait_stream(main)                          # the arena / table uploads were issued on `main`
        with torch.cuda.stream(stream):
            order_d = torch.from_numpy(order).to(self.device)
            spans_d = torch.from_numpy(spans).to(self.device)
        ds = self.ds
        ep = dict(n=n, N=int(ds.grasp_points_num), fn=_lib.load().pngpd_train_batch, arena=self.arena.data_ptr(),
                  f64=int(self.arena.dtype == torch.float64), P=self.arena.shape[0], frames=self.frames_all.data_ptr(),
                  labels=self.labels_all.data_ptr(), order=order_d.data_ptr(), spans=spans_d.data_ptr(),
                  span_stride=spans_d.stride(0) * 4, k=int(ds.pc_file_used_num) if self.fullview else 0,
                  Pg=int(ds.obj_points_num) if self.fullview else 0, min_pts=int(ds.min_point_limit),
                  seed=ctypes.c_ulonglong(self._device_seed()), keep=(order_d, spans_d), main=main)
        nb, R = (n + self.B - 1) // self.B, len(self._slots)
        queue, nxt = collections.deque(), 0
        ep["guard"] = torch.cuda.current_device() != self.device.index
        while nxt < nb or queue:
            while nxt < nb and len(queue) < R:
                queue.append(self._enqueue(nxt, self._slots[nxt % R], ep, stream))
                nxt += 1
            b = queue.popleft()
            slot = b["slot"]
            if not slot.event.query():                        # completed one step ago when prefetching
                slot.event.synchronize()
            kept = int(slot.pin_np)
            if self.prefetch:
                main.wait_event(slot.event)                   # (the stream this iterator was created on)
            self._meta = (b, order, obj, pick)
            if kept == b["G"]:
                yield b["out"] if b["G"] == self.B else b["out"][:kept], b["labels"] if b["G"] == self.B else b["labels"][:kept]
            else:
                yield b["out"][:kept], b["labels"][:kept]

    # ------------------------------------------------------------------ introspection (tests, debugging)
    @property
    def last_meta(self):
        """Details of the batch most recently yielded (valid until the iterator is advanced): items, views, counts,
        keep, labels, gather.  Built on demand — the training loop never pays for it."""
        if self._meta is None:
            return None
        b, order, obj, pick = self._meta
        s, G, slot = b["s"], b["G"], b["slot"]
        items = order[s:s + G].astype(np.int64)
        o, pk = obj[s:s + G], pick[s:s + G]
        if self.fullview:
            views = [[self.files[oi][j] for j in row] for oi, row in zip(o, pk)]
        else:
            views = [self.files[oi][j] for oi, j in zip(o, pk)]
        torch.cuda.current_stream(self.device).wait_event(slot.event)
        return dict(items=items, views=views, counts=slot.counts[:G].clone(), keep=slot.rows[:G] >= 0,
                    labels=self._labels[items], gather=None if slot.gather is None else slot.gather[:G].clone())
