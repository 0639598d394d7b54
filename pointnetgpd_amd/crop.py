"""Batched in-gripper crop / resample: host-side frame construction (numpy, fp64) + the
libpngpd crop kernels.

Reference semantics reproduced (paths relative to the reference root):

* training-style frames  ``BaseGraspDataset.collect_pc``      PointNetGPD/model/dataset.py:15-76
* inference-style frames ``check_collision_square``/``collect_pc``  dex-net/apps/kinect2grasp.py:178-258
  with the gripper constants of dex-net/data/grippers/robotiq_85/params.json
* resampling rule        dataset.py:438-444 (without replacement iff M > N),
                         kinect2grasp.py:473-478 / main_test.py:83-86 (iff M >= N)

A *frame* is 18 doubles: origin[3], M[9] (rows approach, binormal, minor), lo[3], hi[3].
"""
import ctypes

import numpy as np
import torch

from . import _lib

# dex-net/data/grippers/robotiq_85/params.json
ROBOTIQ_85 = dict(hand_outer_diameter=0.218, finger_width=0.0255, hand_depth=0.125,
                  hand_height=0.030, max_width=0.085)
MIN_POINT_LIMIT = 50      # dataset.py:212,386  (training: fewer in-box points -> sample dropped)
MIN_POINTS_TO_NET = 20    # kinect2grasp.py:47  (inference: fewer -> grasp marked bad)
MODE_TRAIN, MODE_INFER = 0, 1


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def frames_from_grasps_train(grasps, transform):
    """(G,>=8) grasp vectors [center3, axis3, width, angle, ...] + the object's (4,4) mesh->cloud
    transform  ->  (G,18) frames.  Vectorised restatement of dataset.py:16-59."""
    g = np.atleast_2d(np.asarray(grasps, dtype=np.float64))
    T = np.asarray(transform, dtype=np.float64)
    center, axis, width, angle = g[:, 0:3], g[:, 3:6], g[:, 6], g[:, 7]
    axis = _unit(axis)                                           # :21
    cos_t, sin_t = np.cos(angle), np.sin(angle)
    # axis_x = [a_y, -a_x, 0], replaced by [1,0,0] when it vanishes (:28-30)
    ax = np.stack([axis[:, 1], -axis[:, 0], np.zeros(len(g))], axis=1)
    deg = np.linalg.norm(ax, axis=1) == 0
    ax[deg] = np.array([1.0, 0.0, 0.0])
    ax = _unit(ax)
    ay = _unit(axis)
    az = np.cross(ax, ay)
    # approach = (R2 @ R1)[:, 0] with R2 = [ax ay az] (columns), R1[:,0] = [cos, 0, sin]  (:26,:34-35)
    approach = ax * cos_t[:, None] + az * sin_t[:, None]
    approach = _unit(approach)
    minor = np.cross(axis, approach)                             # :37
    R, t = T[:3, :3], T[:3, 3]
    center_t = center @ R.T + t                                  # w = 1  (:46)
    rows = np.stack([approach @ R.T, axis @ R.T, minor @ R.T], axis=1)   # w = 0  (:47-51)
    lim = np.stack([width / 4, width / 2, width / 4], axis=1)    # :57-59
    return np.concatenate([center_t, rows.reshape(len(g), 9), -lim, lim], axis=1)


def frames_from_grasps_infer(grasps, gripper=ROBOTIQ_85):
    """(G,5,3)|(G,15) sampler output rows [bottom_center, approach, binormal, minor, bottom_modified]
    (grasp_sampler.py:1616-1618)  ->  (G,18) frames.  kinect2grasp.py:180-187,218-221."""
    g = np.asarray(grasps, dtype=np.float64).reshape(-1, 5, 3)
    rows = np.stack([_unit(g[:, 1]), _unit(g[:, 2]), _unit(g[:, 3])], axis=1)
    w = gripper["hand_outer_diameter"] - 2 * gripper["finger_width"]
    lo = np.array([0.0, -w / 2, -w / 4]); hi = np.array([gripper["hand_depth"], w / 2, w / 4])
    n = len(g)
    return np.concatenate([g[:, 0], rows.reshape(n, 9), np.tile(lo, (n, 1)), np.tile(hi, (n, 1))], axis=1)


def collect_pc_numpy(frame, pc):
    """Host (numpy) crop of ONE grasp: returns (in_ind, points_in_hand_frame[in_ind]).  Used by the
    Dataset mirror inside forked DataLoader workers (no HIP after fork)."""
    f = np.asarray(frame, dtype=np.float64)
    M = f[3:12].reshape(3, 3)
    pc_t = (M.dot((np.asarray(pc) - f[0:3]).T)).T
    lo, hi = f[12:15], f[15:18]
    mask = ((pc_t > lo) & (pc_t < hi)).all(axis=1)
    ind = np.where(mask)[0]
    return ind, pc_t[ind]


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def crop_count_compact(cloud, frames, max_keep=4096):
    """cloud (P,3) CUDA fp32|fp64, frames (G,18) CUDA fp64 -> counts (G) int32, idx (G,max_keep) int32."""
    lib = _lib.load()
    if not cloud.is_cuda or cloud.dim() != 2 or cloud.shape[1] != 3 or cloud.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("cloud: expected a CUDA (P,3) float32/float64 tensor")
    if not frames.is_cuda or frames.dtype != torch.float64 or frames.dim() != 2 or frames.shape[1] != 18:
        raise RuntimeError("frames: expected a CUDA (G,18) float64 tensor")
    cloud, frames = cloud.contiguous(), frames.contiguous()
    P, G = cloud.shape[0], frames.shape[0]
    counts = torch.empty(G, device=cloud.device, dtype=torch.int32)
    idx = torch.empty(G, max_keep, device=cloud.device, dtype=torch.int32)
    with _lib.device_guard(cloud.device):
        _lib.check(lib.pngpd_crop_count_compact(_p(cloud), int(cloud.dtype == torch.float64), P, _p(frames), G,
                                                int(max_keep), _p(counts), _p(idx), _stream(cloud)),
                   "crop_count_compact")
    return counts, idx


def crop_count_compact_ranges(arena, frames, ranges, max_keep=4096):
    """arena (P,3) CUDA fp32|fp64 holding many clouds; frames (G,18) CUDA fp64; ranges (G,2) CUDA int32 = [start, len]
    of the cloud each grasp is cropped against -> counts (G) int32, idx (G,max_keep) int32 ARENA-absolute."""
    lib = _lib.load()
    if not arena.is_cuda or arena.dim() != 2 or arena.shape[1] != 3 or arena.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("arena: expected a CUDA (P,3) float32/float64 tensor")
    if not frames.is_cuda or frames.dtype != torch.float64 or frames.dim() != 2 or frames.shape[1] != 18:
        raise RuntimeError("frames: expected a CUDA (G,18) float64 tensor")
    G = frames.shape[0]
    if not ranges.is_cuda or ranges.dtype != torch.int32 or tuple(ranges.shape) != (G, 2):
        raise RuntimeError("ranges: expected a CUDA (G,2) int32 tensor")
    arena, frames, ranges = arena.contiguous(), frames.contiguous(), ranges.contiguous()
    counts = torch.empty(G, device=arena.device, dtype=torch.int32)
    idx = torch.empty(G, max_keep, device=arena.device, dtype=torch.int32)
    with _lib.device_guard(arena.device):
        _lib.check(lib.pngpd_crop_count_compact_ranges(_p(arena), int(arena.dtype == torch.float64), arena.shape[0],
                                                       _p(frames), _p(ranges), G, int(max_keep), _p(counts), _p(idx),
                                                       _stream(arena)), "crop_count_compact_ranges")
    return counts, idx


def crop_count_compact_gather(arena, frames, gather, max_keep=4096):
    """arena (P,3) CUDA; frames (G,18) CUDA fp64; gather (G,Pg) CUDA int32 arena rows forming each grasp's own cloud
    (duplicates allowed) -> counts (G) int32, idx (G,max_keep) int32 arena-absolute, in the order of ``gather``."""
    lib = _lib.load()
    if not arena.is_cuda or arena.dim() != 2 or arena.shape[1] != 3 or arena.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("arena: expected a CUDA (P,3) float32/float64 tensor")
    if not frames.is_cuda or frames.dtype != torch.float64 or frames.dim() != 2 or frames.shape[1] != 18:
        raise RuntimeError("frames: expected a CUDA (G,18) float64 tensor")
    G = frames.shape[0]
    if not gather.is_cuda or gather.dtype != torch.int32 or gather.dim() != 2 or gather.shape[0] != G:
        raise RuntimeError("gather: expected a CUDA (G,Pg) int32 tensor")
    arena, frames, gather = arena.contiguous(), frames.contiguous(), gather.contiguous()
    counts = torch.empty(G, device=arena.device, dtype=torch.int32)
    idx = torch.empty(G, max_keep, device=arena.device, dtype=torch.int32)
    with _lib.device_guard(arena.device):
        _lib.check(lib.pngpd_crop_count_compact_gather(_p(arena), int(arena.dtype == torch.float64), arena.shape[0],
                                                       _p(frames), _p(gather), gather.shape[1], G, int(max_keep),
                                                       _p(counts), _p(idx), _stream(arena)), "crop_count_compact_gather")
    return counts, idx


def crop_resample(cloud, frames, counts, idx, num_points, mode=MODE_INFER, min_points=MIN_POINTS_TO_NET,
                  seed=0, sel=None, ranges=None, gather=None, g_base=0, rows=None, out=None):
    """-> out (G,3,num_points) fp32 in the hand frame, valid (G) bool.

    The draw of grasp g depends on ``(seed, g_base + g)`` only: pass the GLOBAL index of ``frames[0]`` as ``g_base``
    and a candidate draws the same points in every sharding / batching of the candidate list.
    ``rows`` (G) int32 (``batch_keep_rows``): grasp g is written to ``out[rows[g]]``, nothing when ``rows[g] < 0``;
    ``out`` may then be a caller-owned buffer of at least G rows (the worst-case kept count: ``rows`` lives on the
    device, so its maximum cannot be checked here without a synchronisation, and the kernel does not bound it).

    ``ranges`` / ``gather``: the same per-grasp cloud description the count pass was given
    (``crop_count_compact_ranges`` / ``_gather``).  They matter only for grasps holding MORE than ``max_keep`` in-box
    points: the draw stays uniform over all of them (kinect2grasp.py:473-478, dataset.py:438-444) by re-scanning
    the grasp's own cloud instead of using the truncated index list."""
    lib = _lib.load()
    cloud, frames = cloud.contiguous(), frames.contiguous()
    G, max_keep = idx.shape
    if out is None:
        out = torch.empty(G, 3, num_points, device=cloud.device, dtype=torch.float32)
    elif (not out.is_cuda or out.dtype != torch.float32 or not out.is_contiguous() or out.dim() != 3
          or tuple(out.shape[1:]) != (3, num_points) or out.shape[0] < G):
        raise RuntimeError("out: expected a contiguous CUDA (>= G,3,N) float32 tensor")
    valid = torch.empty(G, device=cloud.device, dtype=torch.uint8)
    if rows is not None:
        if not rows.is_cuda or rows.dtype != torch.int32 or tuple(rows.shape) != (G,):
            raise RuntimeError("rows: expected a CUDA (G,) int32 tensor")
        rows = rows.contiguous()
    if sel is not None:
        if not sel.is_cuda or sel.dtype != torch.int32 or tuple(sel.shape) != (G, num_points):
            raise RuntimeError("sel: expected a CUDA (G,N) int32 tensor")
        sel = sel.contiguous()
    Pg = 0
    if ranges is not None:
        if not ranges.is_cuda or ranges.dtype != torch.int32 or tuple(ranges.shape) != (G, 2):
            raise RuntimeError("ranges: expected a CUDA (G,2) int32 tensor")
        ranges = ranges.contiguous()
    if gather is not None:
        if not gather.is_cuda or gather.dtype != torch.int32 or gather.dim() != 2 or gather.shape[0] != G:
            raise RuntimeError("gather: expected a CUDA (G,Pg) int32 tensor")
        gather = gather.contiguous()
        Pg = gather.shape[1]
    with _lib.device_guard(cloud.device):
        _lib.check(lib.pngpd_crop_resample(_p(cloud), int(cloud.dtype == torch.float64), cloud.shape[0], _p(frames),
                                           _p(ranges), _p(gather), int(Pg), G, _p(counts), _p(idx), int(max_keep),
                                           int(num_points), int(mode), int(min_points),
                                           ctypes.c_ulonglong(int(seed) & (2 ** 64 - 1)),
                                           ctypes.c_longlong(int(g_base)), _p(rows), _p(sel), _p(out),
                                           _p(valid), _stream(cloud)), "crop_resample")
    return out, valid.bool()


def crop_count_compact_indexed(index, frames, max_keep=4096):
    """``crop_count_compact`` over a spatial index (``gpg.CloudIndex`` of the scene): counts (G) int32 as before, idx
    (G,max_keep) int32 = positions in ``index.cloud`` (the Morton-sorted scene), ascending — resample with
    ``crop_resample(index.cloud, ...)``.  Only the chunks whose bounding sphere meets a hand's box are evaluated."""
    lib = _lib.load()
    if not frames.is_cuda or frames.dtype != torch.float64 or frames.dim() != 2 or frames.shape[1] != 18:
        raise RuntimeError("frames: expected a CUDA (G,18) float64 tensor")
    frames = frames.contiguous()
    c, G = index.cloud, frames.shape[0]
    counts = torch.empty(G, device=c.device, dtype=torch.int32)
    idx = torch.empty(G, max_keep, device=c.device, dtype=torch.int32)
    with _lib.device_guard(c.device):
        _lib.check(lib.pngpd_crop_count_compact_indexed(_p(c), int(c.dtype == torch.float64), index.P, _p(index.spheres),
                                                        index.C, _p(frames), G, int(max_keep), _p(counts), _p(idx),
                                                        _stream(c)), "crop_count_compact_indexed")
    return counts, idx


def crop_indexed(index, frames, num_points, mode=MODE_INFER, min_points=MIN_POINTS_TO_NET, seed=0, g_base=0,
                 max_keep=4096):
    """Crop + resample of G hands against one indexed scene in ONE launch (``pngpd_crop_indexed``): the in-box index
    list of a hand stays in its workgroup's LDS.  -> out (G,3,num_points) fp32, counts (G) int32, valid (G) bool —
    bit-identical to ``crop_count_compact_indexed`` + ``crop_resample(index.cloud, ...)`` with the same seed / g_base."""
    lib = _lib.load()
    if not frames.is_cuda or frames.dtype != torch.float64 or frames.dim() != 2 or frames.shape[1] != 18:
        raise RuntimeError("frames: expected a CUDA (G,18) float64 tensor")
    frames = frames.contiguous()
    c, G = index.cloud, frames.shape[0]
    counts = torch.empty(G, device=c.device, dtype=torch.int32)
    out = torch.empty(G, 3, num_points, device=c.device, dtype=torch.float32)
    valid = torch.empty(G, device=c.device, dtype=torch.uint8)
    with _lib.device_guard(c.device):
        _lib.check(lib.pngpd_crop_indexed(_p(c), int(c.dtype == torch.float64), index.P, _p(index.spheres), index.C,
                                          _p(frames), G, int(max_keep), int(num_points), int(mode), int(min_points),
                                          ctypes.c_ulonglong(int(seed) & (2 ** 64 - 1)), ctypes.c_longlong(int(g_base)),
                                          _p(counts), _p(out), _p(valid), _stream(c)), "crop_indexed")
    return out, counts, valid.bool()


def batch_keep_rows(counts, labels, min_points):
    """counts (G) int32 CUDA, labels (G) int64 CUDA (-1 = the reference's ``None``) -> rows (G) int32 (row in the
    compacted batch, -1 = dropped), labels_out (G) int64 (first n entries valid), n_keep () int32 CUDA — ``my_collate``
    (main_1v.py:48-50) with the kept count left on the device."""
    lib = _lib.load()
    G = counts.shape[0]
    if not counts.is_cuda or counts.dtype != torch.int32 or not labels.is_cuda or labels.dtype != torch.int64 \
            or tuple(labels.shape) != (G,):
        raise RuntimeError("batch_keep_rows: counts (G) int32 and labels (G) int64 CUDA tensors expected")
    counts, labels = counts.contiguous(), labels.contiguous()
    rows = torch.empty(G, device=counts.device, dtype=torch.int32)
    labels_out = torch.empty(G, device=counts.device, dtype=torch.int64)
    n_keep = torch.empty((), device=counts.device, dtype=torch.int32)
    with _lib.device_guard(counts.device):
        _lib.check(lib.pngpd_batch_keep_rows(_p(counts), _p(labels), G, int(min_points), _p(rows), _p(labels_out),
                                             _p(n_keep), _stream(counts)), "batch_keep_rows")
    return rows, labels_out, n_keep


def crop_grasps(cloud, frames, num_points, mode=MODE_INFER, min_points=None, seed=0, max_keep=4096, sel=None):
    """Crop + resample in one call.  ``frames`` may be a numpy (G,18) array (uploaded here).
    Returns (out (G,3,N) fp32, counts (G) int32, valid (G) bool)."""
    if isinstance(frames, np.ndarray):
        frames = torch.from_numpy(np.ascontiguousarray(frames, dtype=np.float64)).to(cloud.device)
    if min_points is None:
        min_points = MIN_POINT_LIMIT if mode == MODE_TRAIN else MIN_POINTS_TO_NET
    counts, idx = crop_count_compact(cloud, frames, max_keep)
    out, valid = crop_resample(cloud, frames, counts, idx, num_points, mode, min_points, seed, sel)
    return out, counts, valid
