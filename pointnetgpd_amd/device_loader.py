"""Device-resident training data path ("optional GPU-side crop for training batches").

The reference feeds ``main_1v*.py`` through 32 DataLoader workers, each of which re-loads a grasp file and a cloud
file per sample and crops the cloud in numpy (``dataset.py:420-458``).  On an MI355X the whole dataset fits in a
corner of the 288 GB of HBM: every cloud file of every object is uploaded ONCE into a single fp64 arena, and a
batch is produced by two kernel launches — ``pngpd_crop_count_compact_ranges`` (each grasp cropped against the
arena range of the view drawn for it) and ``pngpd_crop_resample`` (mode 0 = the training rule: without replacement
iff m > N, ``None`` iff fewer than 50 in-box points) — with only the 18-double grasp frames crossing PCIe.

Per-sample semantics are those of ``PointGraspOneViewDataset.__getitem__`` / ``my_collate``:
view drawn uniformly from the object's NP3 clouds (:425-428, shuffle-then-last), training-style crop with the
object's mesh->cloud transform (:429-433), resample rule (:438-444), label rule (:447-453 / :536-541), samples that
come out ``None`` dropped from the batch (main_1v.py:48-50).  For the full-view datasets (``PointGraspDataset``,
:244-282) a sample's cloud is ``obj_points_num`` rows drawn with replacement from the stack of ``pc_file_used_num``
view files themselves drawn with replacement (:252-254); here that becomes a (B, obj_points_num) int32 gather list
built on the device (view slot ~ its share of the stack, row uniform within the view) and
``pngpd_crop_count_compact_gather``.  The random streams differ (numpy global RNG in forked
workers there; one seeded numpy Generator + a counter-hash device RNG here).
"""
import numpy as np
import torch

from . import crop


class DeviceGraspLoader:
    """Iterable of ``(data (B',3,N) fp32 CUDA, target (B',) int64 CUDA)`` batches over one of the four mirror
    datasets of ``model.dataset`` (one-view and full-view, 2- and 3-class).  ``len()`` = batches per epoch.
    ``last_meta`` holds, for the most recent batch, the item indices, the chosen view files, the in-box counts, the
    keep mask, the labels (-1 = the reference's ``None``) and, for full-view datasets, the gather lists."""

    def __init__(self, dataset, batch_size, device, shuffle=True, seed=0, max_keep=8192):
        if getattr(dataset, "projection", False):
            raise NotImplementedError("projection=True belongs to the GPD baseline")
        self.ds, self.B, self.device = dataset, int(batch_size), torch.device(device)
        self.shuffle, self.seed, self.max_keep, self.epoch = bool(shuffle), int(seed), int(max_keep), 0
        if self.device.type != "cuda":
            raise RuntimeError("DeviceGraspLoader needs a CUDA device (the host path is model.dataset + DataLoader)")
        chunks = self._index(dataset)
        self.arena = torch.from_numpy(np.concatenate(chunks, 0)).to(self.device)
        self.last_meta = None

    def _index(self, dataset):
        """Host tables (no device work): arena layout, per-object grasp rows, labels and view ranges."""
        self.ds = dataset
        self.fullview = not hasattr(dataset, "minimum_point_amount")
        self.view_range = {}                                  # path -> (start, len) in the arena
        chunks, off = [], 0
        self.files, self.grasps, self.labels, self.transforms = [], [], [], []
        for obj in dataset.object:
            files = list(dataset.d_pc[dataset.transform[obj][0]])
            for path in files:
                if path in self.view_range:
                    continue
                pc = np.asarray(np.load(path), dtype=np.float64).reshape(-1, 3)
                self.view_range[path] = (off, len(pc))
                chunks.append(pc)
                off += len(pc)
            g = np.asarray(np.load(dataset.d_grasp[obj]), dtype=np.float64)
            lab = [dataset._label(r[-2] + r[-1] * 0.01) for r in g]        # dataset.py:446-453 / :535-541, once
            self.files.append(files)
            self.grasps.append(g)
            self.labels.append(np.array([-1 if v is None else v for v in lab], dtype=np.int64))
            self.transforms.append(dataset.transform[obj][1])
        self.nfiles = np.array([len(f) for f in self.files], dtype=np.int64)
        self.range_tab = np.zeros((len(self.files), int(self.nfiles.max()), 2), dtype=np.int64)
        for oi, files in enumerate(self.files):
            self.range_tab[oi, :len(files)] = [self.view_range[f] for f in files]
        return chunks

    def _assemble(self, items, rng):
        """Host half of a batch, vectorised per object: frames (n,18), view picks, labels (-1 = the reference's None)."""
        ds = self.ds
        obj_ind, grasp_ind = np.unravel_index(items, (len(ds.object), ds.grasp_amount_per_file))
        n = len(items)
        frames = np.empty((n, 18))
        labels = np.empty(n, dtype=np.int64)
        for oi in np.unique(obj_ind):
            sel = obj_ind == oi
            frames[sel] = crop.frames_from_grasps_train(self.grasps[oi][grasp_ind[sel]], self.transforms[oi])
            labels[sel] = self.labels[oi][grasp_ind[sel]]
        if self.fullview:                                     # k views WITH replacement (:252-253)
            pick = rng.integers(0, self.nfiles[obj_ind][:, None], size=(n, ds.pc_file_used_num))
            spans = self.range_tab[obj_ind[:, None], pick]    # (n,k,2)
            views = [[self.files[o][j] for j in row] for o, row in zip(obj_ind, pick)]
        else:                                                 # uniform view (:425-428, shuffle-then-last)
            pick = rng.integers(0, self.nfiles[obj_ind])
            spans = self.range_tab[obj_ind, pick]             # (n,2)
            views = [self.files[o][j] for o, j in zip(obj_ind, pick)]
        return frames, labels, spans, views

    def _gather_lists(self, view_sets, batch_index):
        """(B, obj_points_num) int32 arena rows: ``pc[np.random.choice(len(pc), size=obj_points_num)]`` of the stacked
        views (:253-254) = pick a view slot with probability len_slot / len_stack, then a uniform row of that view."""
        dev = self.device
        vs = torch.from_numpy(np.ascontiguousarray(view_sets, dtype=np.int64)).to(dev)   # (B,k,2) start, len
        start, length = vs[..., 0], vs[..., 1]
        gen = torch.Generator(device=dev)
        gen.manual_seed((self.seed * 1000003 + self.epoch) * 100003 + batch_index)
        n = self.ds.obj_points_num
        slot = torch.multinomial(length.double(), n, replacement=True, generator=gen)   # (B,n)
        u = torch.rand(slot.shape, device=dev, dtype=torch.float64, generator=gen)
        ln = torch.gather(length, 1, slot)
        row = torch.minimum((u * ln.double()).long(), ln - 1)
        return (torch.gather(start, 1, slot) + row).int().contiguous()

    def __len__(self):
        return (len(self.ds) + self.B - 1) // self.B

    @property
    def dataset(self):
        return self.ds

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __iter__(self):
        ds = self.ds
        rng = np.random.default_rng([self.seed, self.epoch])
        order = rng.permutation(len(ds)) if self.shuffle else np.arange(len(ds))
        for bi, s in enumerate(range(0, len(order), self.B)):
            items = order[s:s + self.B]
            frames, labels, spans, views = self._assemble(items, rng)
            has_label = labels >= 0
            fr = torch.from_numpy(frames).to(self.device)
            if self.fullview:
                gather = self._gather_lists(spans, bi)
                counts, idx = crop.crop_count_compact_gather(self.arena, fr, gather, self.max_keep)
            else:
                gather, rg = None, torch.from_numpy(spans.astype(np.int32)).to(self.device)
                counts, idx = crop.crop_count_compact_ranges(self.arena, fr, rg, self.max_keep)
            out, valid = crop.crop_resample(self.arena, fr, counts, idx, ds.grasp_points_num, crop.MODE_TRAIN,
                                            ds.min_point_limit, seed=(self.seed * 1000003 + self.epoch) * 100003 + bi,
                                            ranges=None if self.fullview else rg, gather=gather)
            keep = valid & torch.from_numpy(has_label).to(self.device)     # my_collate drops the Nones
            self.last_meta = dict(items=items, views=views, counts=counts, keep=keep, labels=labels, gather=gather)
            yield out[keep], torch.from_numpy(labels).to(self.device)[keep]
