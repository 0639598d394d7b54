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
    """Iterable of ``(data (B',3,N) fp32 CUDA, target (B',) int64 CUDA)`` batches over a one-view mirror dataset
    (``model.dataset.PointGraspOneViewDataset`` / ``...MultiClassDataset``).  ``len()`` = batches per epoch.
    ``last_meta`` holds, for the most recent batch, the item indices, the chosen view files and the keep mask."""

    def __init__(self, dataset, batch_size, device, shuffle=True, seed=0, max_keep=8192):
        self.fullview = not hasattr(dataset, "minimum_point_amount")
        if getattr(dataset, "projection", False):
            raise NotImplementedError("projection=True belongs to the GPD baseline")
        self.ds, self.B, self.device = dataset, int(batch_size), torch.device(device)
        self.shuffle, self.seed, self.max_keep, self.epoch = bool(shuffle), int(seed), int(max_keep), 0
        if self.device.type != "cuda":
            raise RuntimeError("DeviceGraspLoader needs a CUDA device (the host path is model.dataset + DataLoader)")
        # ---- the arena: every view of every object, once
        self.view_range = {}                                  # path -> (start, len)
        chunks, off = [], 0
        for obj in dataset.object:
            for path in dataset.d_pc[dataset.transform[obj][0]]:
                if path in self.view_range:
                    continue
                pc = np.asarray(np.load(path), dtype=np.float64).reshape(-1, 3)
                self.view_range[path] = (off, len(pc))
                chunks.append(pc)
                off += len(pc)
        self.arena = torch.from_numpy(np.concatenate(chunks, 0)).to(self.device)
        self.grasps = {obj: np.asarray(np.load(dataset.d_grasp[obj]), dtype=np.float64) for obj in dataset.object}
        self.last_meta = None

    def _gather_lists(self, view_sets, batch_index):
        """(B, obj_points_num) int32 arena rows: ``pc[np.random.choice(len(pc), size=obj_points_num)]`` of the stacked
        views (:253-254) = pick a view slot with probability len_slot / len_stack, then a uniform row of that view."""
        dev = self.device
        vs = torch.tensor(view_sets, dtype=torch.int64, device=dev)                  # (B,k,2) start, len
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
            obj_ind, grasp_ind = np.unravel_index(items, (len(ds.object), ds.grasp_amount_per_file))
            frames = np.empty((len(items), 18))
            ranges = np.empty((len(items), 2), dtype=np.int32)
            labels = np.empty(len(items), dtype=np.int64)
            has_label = np.ones(len(items), dtype=bool)
            views, view_sets = [], []
            for i, (oi, gi) in enumerate(zip(obj_ind, grasp_ind)):
                obj = ds.object[oi]
                files = ds.d_pc[ds.transform[obj][0]]
                if self.fullview:                                         # k views WITH replacement (:252-253)
                    pick = rng.integers(0, len(files), size=ds.pc_file_used_num)
                    view_sets.append([self.view_range[files[j]] for j in pick])
                    views.append([files[j] for j in pick])
                else:
                    view = files[int(rng.integers(0, len(files)))]        # uniform view (:425-428)
                    views.append(view)
                    ranges[i] = self.view_range[view]
                grasp = self.grasps[obj][gi]
                frames[i] = crop.frames_from_grasps_train(grasp[None, :], ds.transform[obj][1])[0]
                lab = ds._label(grasp[-2] + grasp[-1] * 0.01)             # :446-453
                has_label[i] = lab is not None
                labels[i] = -1 if lab is None else lab
            fr = torch.from_numpy(frames).to(self.device)
            if self.fullview:
                gather = self._gather_lists(view_sets, bi)
                counts, idx = crop.crop_count_compact_gather(self.arena, fr, gather, self.max_keep)
            else:
                gather = None
                rg = torch.from_numpy(ranges).to(self.device)
                counts, idx = crop.crop_count_compact_ranges(self.arena, fr, rg, self.max_keep)
            out, valid = crop.crop_resample(self.arena, fr, counts, idx, ds.grasp_points_num, crop.MODE_TRAIN,
                                            ds.min_point_limit, seed=(self.seed * 1000003 + self.epoch) * 100003 + bi)
            keep = valid & torch.from_numpy(has_label).to(self.device)     # my_collate drops the Nones
            self.last_meta = dict(items=items, views=views, counts=counts, keep=keep, labels=labels, gather=gather)
            yield out[keep], torch.from_numpy(labels).to(self.device)[keep]
