"""Drop-in mirror of the reference's ``PointNetGPD/model/dataset.py`` for the PointNet path.

Same classes, constructor keywords, on-disk layout, numpy RNG call sequence and return values:

    PointGraspOneViewDataset(grasp_points_num, grasp_amount_per_file, thresh_good, thresh_bad, tag,
                             with_obj=False, projection=False, ...)          reference dataset.py:375-461
    PointGraspOneViewMultiClassDataset(...)                                  :464-549
    PointGraspDataset(obj_points_num, grasp_points_num, pc_file_used_num, ...)   :201-285
    PointGraspMultiClassDataset(...)                                         :288-372
    __getitem__(i) -> (grasp_pc (3,N) float64, label[, obj_name]) or None    :420-458 etc.
    collect_pc(grasp, pc, transform) -> (M,3) float64 in-box points or None  :15-76

``__getitem__`` runs inside forked DataLoader workers, where a HIP context cannot be used, so —
exactly like the reference — it is numpy on the host (fp64).  The crop arithmetic is shared with
the batched GPU crop (``pointnetgpd_amd.crop``: same frames, same strict box test).  The GPD
projection branch (``projection=True``, dataset.py:78-198: the 60x60 images of the CNN baseline) is implemented on
the host without the reference's per-point Python loop; its surface normals come from open3d when installed, or
from ``dataset.normal_estimator`` (open3d's un-oriented estimate cannot be reproduced bit for bit — the images are
pinned given the normals, tests/golden/gpd_projection.npz).  The batched GPU version is ``gpd_ops.project_grasps``.

Differences that do not change results: ``.npy`` files are opened memory-mapped and kept in a small
per-process LRU (the reference re-reads the whole grasp file for every sample).
"""
import glob
import os
import pickle
from collections import OrderedDict

import numpy as np
import torch.utils.data

from .. import crop

_ORDERS = ((0, 1, 2), (1, 2, 0), (0, 2, 1))        # dataset.py:104,110,113


class _NpyCache:
    """Tiny per-process LRU of memory-mapped .npy files."""

    def __init__(self, capacity=64):
        self.capacity = capacity
        self.items = OrderedDict()

    def get(self, path):
        arr = self.items.get(path)
        if arr is None:
            arr = np.load(path, mmap_mode="r")
            self.items[path] = arr
            if len(self.items) > self.capacity:
                self.items.popitem(last=False)
        else:
            self.items.move_to_end(path)
        return arr


class BaseGraspDataset(torch.utils.data.Dataset):
    """Holds the mesh->cloud transforms and the crop (reference dataset.py:10-76)."""

    def __init__(self):
        self.pointnetgpd_dir = os.environ["PointNetGPD_FOLDER"]
        with open(f"{self.pointnetgpd_dir}/PointNetGPD/data/google2cloud.pkl", "rb") as f:
            self.transform = pickle.load(f)
        self._npy = _NpyCache()

    def collect_pc(self, grasp, pc, transform):
        frame = crop.frames_from_grasps_train(np.asarray(grasp, dtype=np.float64)[None, :], transform)[0]
        self.in_ind, pts = crop.collect_pc_numpy(frame, pc)
        if len(self.in_ind) < self.min_point_limit:
            return None
        if self.projection:
            f = np.asarray(frame, dtype=np.float64)
            pc_t = (f[3:12].reshape(3, 3).dot((np.asarray(pc) - f[0:3]).T)).T      # the whole cloud in the hand frame
            return self.project_pc(pc_t, np.asarray(grasp, dtype=np.float64)[6])  # width = grasp[6], dataset.py:22
        return pts

    # ---- GPD baseline images (reference dataset.py:78-198) -------------------------------------------------
    normal_estimator = None     # callable(points (P,3)) -> normals (P,3); None = open3d's estimate_normals (:78-86)

    def get_normal(self, points, radius=0.1, max_nn=30):
        if self.normal_estimator is not None:
            return np.asarray(self.normal_estimator(points), dtype=np.float64)
        try:
            import open3d as o3d
            pcd = o3d.geometry.PointCloud()
            pcd.points = o3d.utility.Vector3dVector(points)
            pcd.estimate_normals(search_param=o3d.geometry.KDTreeSearchParamHybrid(radius=radius, max_nn=max_nn))
            return np.asarray(pcd.normals)
        except (ImportError, AttributeError) as e:
            raise RuntimeError("projection=True needs surface normals: open3d is not installed here — set "
                               "`dataset.normal_estimator = fn(points) -> normals`") from e

    def project_pc(self, pc, gripper_width):
        """dataset.py:88-118: (60,60,3|12) float64 images of the in-box points ``pc[self.in_ind]``."""
        nrm = self.get_normal(pc)
        pts, nrm = np.asarray(pc, dtype=np.float64)[self.in_ind], np.asarray(nrm, dtype=np.float64)[self.in_ind]
        ok = ~np.isnan(nrm).any(axis=1)                              # rows with a NaN normal are deleted (:97-101)
        pts, nrm = pts[ok], nrm[ok]
        occ1, n1 = self.cal_projection(pts, self.project_size, self.projection_margin, nrm, _ORDERS[0], gripper_width)
        if self.project_chann == 3:
            return n1
        occ2, n2 = self.cal_projection(pts, self.project_size, self.projection_margin, nrm, _ORDERS[1], gripper_width)
        occ3, n3 = self.cal_projection(pts, self.project_size, self.projection_margin, nrm, _ORDERS[2], gripper_width)
        return np.dstack([occ1, n1, occ2, n2, occ3, n3])

    def cal_projection(self, point_cloud_voxel, m_width_of_pic, margin, surface_normal, order, gripper_width):
        """dataset.py:139-198 without the Python loop over points: voxels are ranked by a stable sort, the float32
        normal sums are accumulated rank by rank (sequential per voxel, like the reference's buffer fill), and the
        (x, y) pixel keeps the voxel with the largest z index (numpy's last-write-wins on the sorted unique list)."""
        S = m_width_of_pic
        occupy, norm = np.zeros((S, S, 1)), np.zeros((S, S, 3))
        p = np.asarray(point_cloud_voxel, dtype=np.float64)
        a, b = p[:, order[0]], p[:, order[1]]
        if max(a.max() - a.min(), b.max() - b.min()) == 0:
            return occupy, norm
        res = gripper_width / (S - margin)
        vox = np.stack([np.floor(p[:, o] / res + S / 2).astype(np.int64) for o in order], 1)
        uniq, inv = np.unique(vox, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        first = np.argsort(inv, kind="stable")                       # points grouped by voxel, input order kept
        start = np.searchsorted(inv[first], np.arange(len(uniq)))
        rank = np.empty(len(p), dtype=np.int64)
        rank[first] = np.arange(len(p)) - start[inv[first]]
        number = np.minimum(np.bincount(inv, minlength=len(uniq)), self.voxel_point_num)
        acc = np.zeros((len(uniq), 3), dtype=np.float32)
        n32 = np.asarray(surface_normal, dtype=np.float64).astype(np.float32)
        for r in range(int(number.max())):                           # <= voxel_point_num rounds, each one float32 add
            sel = np.nonzero(rank == r)[0]
            acc[inv[sel]] = acc[inv[sel]] + n32[sel]
        norm[uniq[:, 0], uniq[:, 1], :] = acc.astype(np.float64) / number[:, None].astype(np.float64)
        occupy[uniq[:, 0], uniq[:, 1], 0] = number
        return occupy / occupy.max(), norm

    # ---- shared pieces of the four concrete datasets
    def _init_common(self, grasp_points_num, grasp_amount_per_file, thresh_good, thresh_bad, tag, with_obj,
                     projection, project_chann, project_size):
        self.grasp_points_num = grasp_points_num
        self.grasp_amount_per_file = grasp_amount_per_file
        self.tag = tag
        self.thresh_good = thresh_good
        self.thresh_bad = thresh_bad
        self.with_obj = with_obj
        self.min_point_limit = 50
        self.projection = projection
        self.project_chann = project_chann
        if self.project_chann not in [3, 12]:
            raise NotImplementedError
        self.project_size = project_size
        if self.project_size != 60:
            raise NotImplementedError
        self.voxel_point_num = 50
        self.projection_margin = 1

    def _index_files(self, cloud_glob, sort_clouds):
        fl_grasp = glob.glob(f"{self.pointnetgpd_dir}/PointNetGPD/data/ycb_grasp/{self.tag}/*.npy")
        fl_pc = glob.glob(f"{self.pointnetgpd_dir}/data/ycb-tools/models/ycb/*/rgbd/clouds/{cloud_glob}")
        self.d_pc, self.d_grasp = {}, {}
        for path in fl_pc:
            self.d_pc.setdefault(path.split("/")[-4], []).append(path)
        if sort_clouds:
            for k in self.d_pc:
                self.d_pc[k].sort()
        for path in fl_grasp:
            self.d_grasp[path.split("/")[-1].split(".")[0]] = path
        self.object = list(set(self.d_grasp.keys()).intersection(set(self.transform.keys())))
        self.amount = len(self.object) * self.grasp_amount_per_file

    def _resample(self, grasp_pc):
        n = self.grasp_points_num
        replace = not (len(grasp_pc) > n)                       # dataset.py:439-444
        return grasp_pc[np.random.choice(len(grasp_pc), size=n, replace=replace)].T

    def _label(self, score):
        raise NotImplementedError

    def _finish(self, grasp, grasp_pc, obj_grasp):
        if grasp_pc is None:
            return None
        level_score, refine_score = grasp[-2:]
        grasp_pc = grasp_pc.transpose((2, 1, 0)) if self.projection else self._resample(grasp_pc)   # dataset.py:262-270
        label = self._label(level_score + refine_score * 0.01)
        if label is None:
            return None
        if self.with_obj:
            return grasp_pc, label, obj_grasp
        return grasp_pc, label

    def __len__(self):
        return self.amount


def _label_2class(self, score):
    """dataset.py:447-453: bad -> 0, good -> 1, in between -> sample dropped."""
    if score >= self.thresh_bad:
        return 0
    if score <= self.thresh_good:
        return 1
    return None


def _label_3class(self, score):
    """dataset.py:535-541: bad -> 0, good -> 2, in between -> 1."""
    if score >= self.thresh_bad:
        return 0
    if score <= self.thresh_good:
        return 2
    return 1


class _FullView(BaseGraspDataset):
    """Multi-view object cloud (reference dataset.py:201-285)."""

    def __init__(self, obj_points_num, grasp_points_num, pc_file_used_num, grasp_amount_per_file, thresh_good,
                 thresh_bad, tag, with_obj=False, projection=False, project_chann=3, project_size=60):
        super().__init__()
        self.obj_points_num = obj_points_num
        self.pc_file_used_num = pc_file_used_num
        self._init_common(grasp_points_num, grasp_amount_per_file, thresh_good, thresh_bad, tag, with_obj,
                          projection, project_chann, project_size)
        self._index_files("*.npy", sort_clouds=False)

    def __getitem__(self, index):
        obj_ind, grasp_ind = np.unravel_index(index, (len(self.object), self.grasp_amount_per_file))
        obj_grasp = self.object[obj_ind]
        obj_pc = self.transform[obj_grasp][0]
        fl_pc = np.array(self.d_pc[obj_pc])
        fl_pc = fl_pc[np.random.choice(len(fl_pc), size=self.pc_file_used_num)]
        grasp = np.array(self._npy.get(self.d_grasp[obj_grasp])[grasp_ind])
        pc = np.vstack([self._npy.get(str(i)) for i in fl_pc])
        pc = pc[np.random.choice(len(pc), size=self.obj_points_num)]
        t = self.transform[obj_grasp][1]
        return self._finish(grasp, self.collect_pc(grasp, pc, t), obj_grasp)


class _OneView(BaseGraspDataset):
    """Single-view cloud, NP3 camera only (reference dataset.py:375-461)."""

    def __init__(self, grasp_points_num, grasp_amount_per_file, thresh_good, thresh_bad, tag, with_obj=False,
                 projection=False, project_chann=3, project_size=60):
        super().__init__()
        self._init_common(grasp_points_num, grasp_amount_per_file, thresh_good, thresh_bad, tag, with_obj,
                          projection, project_chann, project_size)
        self.minimum_point_amount = 150
        self._index_files("pc_NP3_NP5*.npy", sort_clouds=True)

    def __getitem__(self, index):
        obj_ind, grasp_ind = np.unravel_index(index, (len(self.object), self.grasp_amount_per_file))
        obj_grasp = self.object[obj_ind]
        obj_pc = self.transform[obj_grasp][0]
        fl_pc = np.array(self.d_pc[obj_pc])
        np.random.shuffle(fl_pc)
        grasp = np.array(self._npy.get(self.d_grasp[obj_grasp])[grasp_ind])
        pc = np.asarray(self._npy.get(str(fl_pc[-1])))
        t = self.transform[obj_grasp][1]
        return self._finish(grasp, self.collect_pc(grasp, pc, t), obj_grasp)


class PointGraspDataset(_FullView):
    _label = _label_2class


class PointGraspMultiClassDataset(_FullView):
    _label = _label_3class


class PointGraspOneViewDataset(_OneView):
    _label = _label_2class


class PointGraspOneViewMultiClassDataset(_OneView):
    _label = _label_3class
