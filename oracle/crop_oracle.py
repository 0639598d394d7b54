"""CPU oracle (numpy, fp64) for the in-gripper crop / resample / label step.

TEST INFRASTRUCTURE ONLY — see the header of ``oracle/pointnet_oracle.py``.

Restates (citations relative to /root/reference/):

* training-style crop  ``BaseGraspDataset.collect_pc``  PointNetGPD/model/dataset.py:15-76
* inference-style crop ``check_collision_square`` (branch ``use_dataset_py``)
  dex-net/apps/kinect2grasp.py:178-235 and ``collect_pc`` :238-258, with the
  gripper constants of dex-net/data/grippers/robotiq_85/params.json
* resample + label     ``__getitem__``  dataset.py:438-453 (2-class), :536-541 (3-class)

Parity pin: ``collect_pc`` of the unmodified reference is importable in the
build container (open3d stubbed); ``oracle/make_golden.py`` runs it on seeded
synthetic clouds and commits inputs+outputs under ``tests/golden/``.
``kinect2grasp.py`` cannot be imported (rospy, pcl, node start-up at module level), but its
two crop functions only need numpy and a sampler object: ``oracle/make_golden_gpg.py`` cuts their
unmodified source segments out of the reference file with ``ast`` and executes them; the record
(102 grasps on 3 scenes: index sets, hand-frame points) is ``tests/golden/crop_infer.npz`` and pins
``collect_pc_infer`` below (``tests/test_gpg_cpu.py``).
"""
from __future__ import annotations

import numpy as np

# dex-net/data/grippers/robotiq_85/params.json
ROBOTIQ_85 = dict(hand_outer_diameter=0.218, finger_width=0.0255, hand_depth=0.125,
                  hand_height=0.030, max_width=0.085)

MIN_POINT_LIMIT = 50          # dataset.py:212,386
MIN_POINTS_TO_NET = 20        # kinect2grasp.py:47  minimal_points_send_to_point_net


def grasp_frame_train(grasp, transform):
    """dataset.py:16-50.  grasp: (>=8,) = [center3, axis3, width, angle, ...];
    transform: (4,4).  Returns (center (3,), M (3,3) rows=[approach,binormal,minor], width)."""
    grasp = np.asarray(grasp, dtype=np.float64)
    transform = np.asarray(transform, dtype=np.float64)
    center = grasp[0:3]
    axis = grasp[3:6]
    width = grasp[6]
    angle = grasp[7]
    axis = axis / np.linalg.norm(axis)                      # :21
    binormal = axis
    cos_t, sin_t = np.cos(angle), np.sin(angle)             # :24-25
    # np.c_ of three lists builds COLUMNS  :26
    R1 = np.array([[cos_t, 0.0, -sin_t],
                   [0.0, 1.0, 0.0],
                   [sin_t, 0.0, cos_t]])
    axis_y = axis
    axis_x = np.array([axis_y[1], -axis_y[0], 0.0])         # :28
    if np.linalg.norm(axis_x) == 0:                         # :29
        axis_x = np.array([1.0, 0.0, 0.0])
    axis_x = axis_x / np.linalg.norm(axis_x)
    axis_y = axis_y / np.linalg.norm(axis_y)
    axis_z = np.cross(axis_x, axis_y)
    R2 = np.stack([axis_x, axis_y, axis_z], axis=1)         # columns  :34
    approach = R2.dot(R1)[:, 0]                             # :35
    approach = approach / np.linalg.norm(approach)
    minor = np.cross(axis, approach)                        # :37
    hom = lambda v, w: transform.dot(np.array([v[0], v[1], v[2], w]))[:3]
    center_t = hom(center, 1.0)                             # :46
    binormal_t = hom(binormal, 0.0)                         # :47
    approach_t = hom(approach, 0.0)                         # :48
    minor_t = hom(minor, 0.0)                               # :49
    M = np.stack([approach_t, binormal_t, minor_t], axis=0)  # :51 (hstack of columns).T
    return center_t, M, width


def box_mask(pc_t, lo, hi):
    """Strict inequalities on all three axes (dataset.py:61-69; kinect2grasp.py:222-229)."""
    return ((pc_t[:, 0] > lo[0]) & (pc_t[:, 0] < hi[0]) &
            (pc_t[:, 1] > lo[1]) & (pc_t[:, 1] < hi[1]) &
            (pc_t[:, 2] > lo[2]) & (pc_t[:, 2] < hi[2]))


def collect_pc_train(grasp, pc, transform, min_point_limit=MIN_POINT_LIMIT):
    """dataset.py:15-76 with projection=False.  Returns (points_in_box (M,3) f64 or None,
    in_ind (M,) int64, pc_t (P,3))."""
    pc = np.asarray(pc, dtype=np.float64)
    center, M, width = grasp_frame_train(grasp, transform)
    pc_t = (M.dot((pc - center).T)).T                       # :53
    lim = np.array([width / 4, width / 2, width / 4])       # :57-59
    in_ind = np.where(box_mask(pc_t, -lim, lim))[0]
    if len(in_ind) < min_point_limit:                       # :71
        return None, in_ind, pc_t
    return pc_t[in_ind], in_ind, pc_t


def grasp_frame_infer(grasp5x3):
    """kinect2grasp.py:180-187.  grasp5x3 rows = [bottom_center, approach, binormal, minor,
    bottom_center_modified] (grasp_sampler.py:1616-1618).  Each direction is re-normalised."""
    g = np.asarray(grasp5x3, dtype=np.float64).reshape(5, 3)
    a = g[1] / np.linalg.norm(g[1])
    b = g[2] / np.linalg.norm(g[2])
    m = g[3] / np.linalg.norm(g[3])
    return g[0], np.stack([a, b, m], axis=0)


def infer_box(gripper=ROBOTIQ_85):
    """kinect2grasp.py:218-221: x in (0, hand_depth), |y| < w/2, |z| < w/4,
    w = hand_outer_diameter - 2*finger_width."""
    w = gripper["hand_outer_diameter"] - 2 * gripper["finger_width"]
    lo = np.array([0.0, -w / 2, -w / 4])
    hi = np.array([gripper["hand_depth"], w / 2, w / 4])
    return lo, hi


def collect_pc_infer(grasps, pc, gripper=ROBOTIQ_85):
    """kinect2grasp.py:238-258.  grasps: (G,5,3) or (G,15); pc: (P,3) (fp32 at :112,
    promoted to fp64 by the subtraction with an fp64 centre).  Returns
    (list of index arrays, list of (Mi,3) fp64 point arrays in the hand frame)."""
    pc = np.asarray(pc)
    grasps = np.asarray(grasps, dtype=np.float64).reshape(-1, 5, 3)
    lo, hi = infer_box(gripper)
    in_ind, in_pts = [], []
    for g in grasps:
        bottom, M = grasp_frame_infer(g)
        pts = pc - bottom.reshape(1, 3)                     # :188
        points_g = M.dot(pts.T).T                           # :189-190
        idx = np.where(box_mask(points_g, lo, hi))[0]
        in_ind.append(idx)
        in_pts.append(points_g[idx])
    return in_ind, in_pts


def resample_rule(m, n):
    """dataset.py:438-444: without replacement iff m > n.
    (kinect2grasp.py:474-478 / main_test.py:83-86 use >=; both reported.)"""
    return dict(replace_train=not (m > n), replace_infer=not (m >= n))


def resample_with_indices(points, idx):
    """Apply a caller-supplied index vector (the RNG stream of np.random.choice cannot be
    matched on a GPU, SURVEY.md §7 'RNG parity'): returns (3,N) like ``grasp_pc[idx].T``."""
    return np.asarray(points)[np.asarray(idx)].T


def label_2class(level_score, refine_score, thresh_good, thresh_bad):
    """dataset.py:447-453.  Returns 0 / 1 / None."""
    score = level_score + refine_score * 0.01
    if score >= thresh_bad:
        return 0
    if score <= thresh_good:
        return 1
    return None


def label_3class(level_score, refine_score, thresh_good, thresh_bad):
    """dataset.py:535-541 (PointGraspOneViewMultiClassDataset): bad→0, good→2, else 1."""
    score = level_score + refine_score * 0.01
    if score >= thresh_bad:
        return 0
    if score <= thresh_good:
        return 2
    return 1
