"""GPG grasp-candidate sampler, batched on the GPU — the upstream of the in-gripper crop.

Mirror of ``GpgGraspSamplerPcl.sample_grasps`` (dex-net/src/dexnet/grasping/grasp_sampler.py:1383-1656), the
sampler ``kinect2grasp.py:150`` calls.  The reference walks the sample points one at a time and, for each, runs
19 rotations x 21 lateral offsets x 2-4 ``check_collision_square`` numpy passes over the whole cloud, then up to
25 push-in steps x 7 passes per surviving pose.  Here all sample points of a call are processed together:

1. ``pngpd_gpg_normal_moments``   one workgroup per sample point: r-ball / 100-NN selection and M = sum n n^T
2. host: ONE batched ``np.linalg.eig`` over the K 3x3 matrices — the same LAPACK call as the reference, because the
   arbitrary eigenvector SIGNS it returns decide the enumeration order of the sweep — and the local frames
3. ``pngpd_gpg_enumerate``        rotation about the minor axis x lateral offsets -> the 19 x 21 poses per sample point
4. ``pngpd_hand_box_counts*``     one wave per pose, four hand boxes per transformed point, the whole sweep in ONE launch
5. ``pngpd_gpg_select``           middle admissible offset per rotation, 30-degree rule, ordered list of potential grasps
6. ``pngpd_gpg_pushin``           every push-in step and its backed-off, table-corrected twin (hand corners on the fly)
7. ``pngpd_hand_box_counts*``     collision at every step / twin; the number of valid poses stays on the device
8. ``pngpd_gpg_finish``           first accepted step per potential grasp, rows packed in the reference's output order
9. host: ``num_grasps`` / ``max_num_samples`` stop rule over the per-sample-point counts

Per round of sample points: one upload (sample points), one download of the K moment matrices, one upload of the K
frames, one download of the packed result — the pose enumeration, selection and push-in logic never touch the host.

Reference quirks reproduced on purpose (see oracle/gpg_oracle.py for the executed-reference pin):
* the "rotation by dtheta" is ``rotation_from_quaternion([dtheta_rad, minor])`` on an un-normalised quaternion;
* ``all_normal[ind]`` indexes the FULL-cloud normals with an index into ``points_for_sample`` (:1510);
* the table back-off distance is the Frobenius norm of the stacked (lowest corner, table point) pair (:1605);
* a sample point whose M is all-zero consumes a draw but does not count towards ``max_num_samples`` (:1486-1489)
  — the reference can loop forever there; this implementation gives up after 10 x max_num_samples draws.

The reference reseeds numpy from the OS before every draw (:1455) and so has no reproducible stream; here the
draws come from ``sample_indices`` (explicit), else ``numpy.random.default_rng(seed)``.
"""
import collections
import ctypes
import os
import threading

import numpy as np
import torch

from . import _lib
from .crop import ROBOTIQ_85 as _CROP_GRIPPER
from .ops import _call

# dex-net/data/grippers/robotiq_85/params.json
ROBOTIQ_85 = dict(_CROP_GRIPPER, init_bite=0.01)

NUM_DY, DTHETA, RANGE_DTHETA = 10, 10, 90      # grasp_sampler.py:1413-1415
APPROACH_STEP = 0.005                          # :1418
MAX_NN = 100                                   # :1475
TABLE_CLEARANCE = 0.01                         # :1599 safety_dis_above_table
MIN_OPEN_POINTS = 10                           # :1614
BOX_OPEN, BOX_LEFT, BOX_RIGHT, BOX_BOTTOM = 0, 1, 2, 3
_TLS = threading.local()                       # per-thread pool of pinned staging buffers (two downloads per round)


def _pin_pool():
    pool = getattr(_TLS, "pool", None)
    if pool is None:
        pool = _TLS.pool = []
    return pool


class _RoundWorkspace:
    """Device scratch of one sampler round: ONE allocation, carved per round.  The per-round tensors are hundreds of MB
    (poses2 alone: 0.75 GB at 8,192 sample points) of slightly different sizes from round to round (the live count
    varies, the last round is short): handed to the caching allocator they fragment its large pool, and a fresh 400 MB
    hipMalloc costs 60-90 ms — measured as a stall per call that moved from stage to stage.  Rounds in flight use
    different slots; a slot is reused only by a later round on the SAME stream, i.e. behind its previous user."""

    def __init__(self):
        self.buf, self.off = None, 0

    def reset(self, dev, nbytes):
        if self.buf is None or self.buf.device != dev or self.buf.numel() < nbytes:
            self.buf = None                                   # drop first: never hold two generations
            self.buf = torch.empty(int(nbytes * 1.25) + 4096, device=dev, dtype=torch.uint8)
        self.off = 0

    def take(self, shape, dtype):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        t = self.buf[self.off:self.off + nbytes].view(dtype).view(*shape)
        self.off += (nbytes + 255) & ~255
        return t


def _ws_ring(dev):
    ring = getattr(_TLS, "ws", None)
    if ring is None:
        ring = _TLS.ws = {}
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
    if key not in ring:
        ring[key] = [_RoundWorkspace() for _ in range(4)]
    return ring[key]


_EIG_POOL = None


def _eig_workers():
    from .hostbudget import threads_per_rank
    return threads_per_rank(cap=8)


def _batched_eig(M):
    """``np.linalg.eig`` of a (n,3,3) stack — the very LAPACK call of the reference (:1493; its eigenvector signs decide
    the sweep's enumeration order), one matrix at a time inside numpy's gufunc loop.  Each matrix is solved on its own, so
    slicing the stack over a few host threads returns the same bits (numpy releases the GIL inside the loop): 20,000
    matrices take 28 ms on one core of the GPU box and 4.3 ms on eight.  The pool is this RANK's share of the node
    (hostbudget.threads_per_rank: affinity mask and cgroup quota divided by the ranks on the node — eight ranks of a
    config-5 run do not start 64 threads on a host that has 8 cores for all of them)."""
    global _EIG_POOL
    n = M.shape[0]
    workers = _eig_workers()
    if n < 512 or workers < 2:
        return np.linalg.eig(M)
    if _EIG_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _EIG_POOL = ThreadPoolExecutor(workers, thread_name_prefix="pngpd-eig")
    parts = np.array_split(np.arange(n), workers)
    res = list(_EIG_POOL.map(lambda ix: np.linalg.eig(M[ix[0]:ix[-1] + 1]), [ix for ix in parts if ix.size]))
    return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])


def _gripper_dict(gripper):
    keys = ("hand_outer_diameter", "finger_width", "hand_depth", "hand_height", "init_bite")
    if isinstance(gripper, dict):
        return {k: float(gripper[k]) for k in keys}
    return {k: float(getattr(gripper, k)) for k in keys}     # RobotGripper-like object


# The hand model (:287-321) as a construction table: corner = parent + axis * scale.  Axes: a = approach,
# b = binormal, m = unit(a x b).  Built in this order so that every corner sees the same sequence of roundings
# as the reference's p1..p20 (negations and the factor 0.5 are exact).
def _hand_table(g):
    hh, fw, hd = g["hand_height"], g["finger_width"], g["hand_depth"]
    ow = g["hand_outer_diameter"] - fw * 2
    t = [("u", "c", "m", hh * 0.5), ("d", "c", "m", -(hh * 0.5)),
         (5, "u", "b", -(ow * 0.5)), (6, "u", "b", ow * 0.5), (7, "d", "b", ow * 0.5), (8, "d", "b", -(ow * 0.5)),
         (1, 5, "a", hd), (2, 6, "a", hd), (3, 7, "a", hd), (4, 8, "a", hd),
         (9, 1, "b", -fw), (10, 4, "b", -fw), (11, 5, "b", -fw), (12, 8, "b", -fw),
         (13, 2, "b", fw), (14, 3, "b", fw), (15, 6, "b", fw), (16, 7, "b", fw),
         (17, 11, "a", -hh), (18, 15, "a", -hh), (19, 16, "a", -hh), (20, 12, "a", -hh)]
    return t


def hand_corners(g, center, approach, binormal):
    """(..., 3) arrays -> (..., 20, 3): corners p1..p20 of the hand model in the world frame."""
    center, approach, binormal = (np.asarray(v, dtype=np.float64) for v in (center, approach, binormal))
    m = np.cross(approach, binormal)
    m = m / np.linalg.norm(m, axis=-1, keepdims=True)
    axes = {"a": approach, "b": binormal, "m": m}
    pts = {"c": center}
    for name, parent, ax, scale in _hand_table(g):
        pts[name] = axes[ax] * scale + pts[parent]
    return np.stack([pts[i] for i in range(1, 21)], axis=-2)


def hand_boxes(g):
    """(4,6) strict bounds [x_lo, x_hi, y_lo, y_hi, z_lo, z_hi] of p_open, p_left, p_right, p_bottom (:361-377)."""
    c = hand_corners(g, np.zeros(3), np.array([1.0, 0, 0]), np.array([0, 1.0, 0]))
    p = {i + 1: c[i] for i in range(20)}
    sel = [(p[1], p[2], p[4], p[8]), (p[9], p[1], p[10], p[12]), (p[2], p[13], p[3], p[7]), (p[11], p[15], p[12], p[20])]
    return np.array([[s8[0], s4[0], s1[1], s2[1], s4[2], s1[2]] for s1, s2, s4, s8 in sel])


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


# ------------------------------------------------------------------------------------------------
# device entry points
# ------------------------------------------------------------------------------------------------
def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _check_cloud(cloud):
    if not cloud.is_cuda or cloud.dim() != 2 or cloud.shape[1] != 3 or cloud.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("cloud: expected a CUDA (P,3) float32/float64 tensor")
    return cloud.contiguous()


def normal_moments(cloud, normals, queries, radius, max_nn=MAX_NN, index=None):
    """cloud (P,3) CUDA f32|f64, normals (P,3) CUDA f64, queries (K,3) CUDA f64 -> M (K,3,3) f64, nsel (K) int32.
    ``index`` (the cloud's ``CloudIndex``): only the chunks that can hold one of the max_nn nearest points are scanned
    (``pngpd_gpg_normal_moments_indexed``); same selection, same order of additions: M is bit-identical."""
    lib = _lib.load()
    cloud = _check_cloud(cloud)
    if not normals.is_cuda or normals.dtype != torch.float64 or tuple(normals.shape) != tuple(cloud.shape):
        raise RuntimeError("normals: expected a CUDA (P,3) float64 tensor")
    if not queries.is_cuda or queries.dtype != torch.float64 or queries.dim() != 2 or queries.shape[1] != 3:
        raise RuntimeError("queries: expected a CUDA (K,3) float64 tensor")
    normals, queries = normals.contiguous(), queries.contiguous()
    K = queries.shape[0]
    M = torch.empty(K, 3, 3, device=cloud.device, dtype=torch.float64)
    nsel = torch.empty(K, device=cloud.device, dtype=torch.int32)
    with _lib.device_guard(cloud.device):
        if index is not None and max_nn <= 1024:
            c = index.cloud
            _lib.check(lib.pngpd_gpg_normal_moments_indexed(_p(c), int(c.dtype == torch.float64), _p(index.order),
                                                            _p(normals), index.P, _p(index.spheres), index.C,
                                                            _p(queries), K, float(radius), int(max_nn), _p(M), _p(nsel),
                                                            _stream(c)), "gpg_normal_moments_indexed")
        else:
            _lib.check(lib.pngpd_gpg_normal_moments(_p(cloud), int(cloud.dtype == torch.float64), _p(normals),
                                                    cloud.shape[0], _p(queries), K, float(radius), int(max_nn), _p(M),
                                                    _p(nsel), _stream(cloud)), "gpg_normal_moments")
    return M, nsel


def local_frames(M, normals_at, points):
    """grasp_sampler.py:1486-1506 on the device (``pngpd_gpg_frames``): M (K,3,3), normals_at (K,3), points (K,3) CUDA f64
    -> frames (K,12) f64 = minor, normal, major, sample point; flags (K,) int32 (1 = zero moment matrix: the point is
    skipped, 2 = LAPACK's complex-pair case, 4 / 8 = not converged / out of range).  ``np.linalg.eig`` is evaluated as
    LAPACK's DGEEV evaluates it (same eigenvalue order and eigenvector signs: csrc/pngpd_gpg_eig3.h)."""
    for name, t, shp in (("M", M, (3, 3)), ("normals_at", normals_at, (3,)), ("points", points, (3,))):
        if not t.is_cuda or t.dtype != torch.float64 or tuple(t.shape[1:]) != shp or t.shape[0] != M.shape[0]:
            raise RuntimeError(f"{name}: expected a CUDA (K,{','.join(map(str, shp))}) float64 tensor")
    K = M.shape[0]
    frames = torch.empty(K, 12, device=M.device, dtype=torch.float64)
    flags = torch.empty(K, device=M.device, dtype=torch.int32)
    if K:
        _call("pngpd_gpg_frames", M, M.contiguous(), normals_at.contiguous(), points.contiguous(), K, frames, flags)
    return frames, flags


def _spread3(v):
    """10-bit integers -> bits spread to every third position (Morton interleave helper), int64 tensors."""
    v = v & 0x3FF
    v = (v | (v << 16)) & 0x30000FF
    v = (v | (v << 8)) & 0x300F00F
    v = (v | (v << 4)) & 0x30C30C3
    v = (v | (v << 2)) & 0x9249249
    return v


class CloudIndex:
    """Spatial index of a scene cloud for ``hand_box_counts``: the points re-ordered along a 30-bit Morton curve
    (consecutive points are neighbours) and the bounding sphere of every 64-point chunk.  Built once per scene from a
    handful of torch ops on the device; point VALUES are untouched, so counts are identical to the un-indexed path."""

    def __init__(self, cloud):
        cloud = _check_cloud(cloud)
        P = cloud.shape[0]
        pts = cloud.double()
        # A non-finite point can be inside no box (every face test is a strict comparison that NaN / Inf fail — the
        # reference drops exactly that point, kinect2grasp.py:218-229), but it must not poison the index: it takes no
        # part in the bounding box or in its chunk's sphere (a stand-in finite point does) and sorts to the end.  The
        # point VALUES in ``self.cloud`` are untouched, so the narrow phase still rejects it and only it.  No host sync.
        fin = torch.isfinite(pts).all(1)
        anchor = pts[torch.argmax(fin.to(torch.uint8))]
        safe = torch.where(fin[:, None], pts, anchor[None, :])
        lo = safe.min(0).values
        ext = (safe.max(0).values - lo).clamp_min(1e-30)
        q = ((safe - lo) / ext * 1023.0).long().clamp_(0, 1023)
        code = _spread3(q[:, 0]) | (_spread3(q[:, 1]) << 1) | (_spread3(q[:, 2]) << 2)
        code = torch.where(fin, code, torch.full_like(code, 1 << 30))
        # stable: points with equal codes keep their input order, so the sorted cloud — and with it every keyed
        # resampling draw downstream — is the same on every rank and in every run
        order = torch.argsort(code, stable=True)
        self.order = order.int().contiguous()              # sorted position -> original index
        self.cloud = cloud[order].contiguous()
        C = (P + 63) // 64
        sp = safe[order]
        if C * 64 != P:                                   # pad the last chunk with its own last point
            sp = torch.cat([sp, sp[-1:].expand(C * 64 - P, 3)], 0)
        sp = sp.view(C, 64, 3)
        centre = (sp.min(1).values + sp.max(1).values) * 0.5
        radius = (sp - centre[:, None, :]).norm(dim=2).max(1).values
        self.spheres = torch.cat([centre, radius[:, None]], 1).contiguous()      # (C,4) f64
        self.P, self.C = P, C


def hand_box_counts(cloud, poses, boxes, index=None, valid_units=None, per_unit=1):
    """cloud (P,3) CUDA; poses (Q,12) CUDA f64 [centre, approach, binormal, minor] (unit axes); boxes (NB,6) CUDA f64,
    NB in {1,4} -> counts (Q,NB) int32: cloud points strictly inside each box of each pose.
    ``index`` (a ``CloudIndex`` of the same cloud) selects the sphere-culling kernel: same counts, less work.
    ``valid_units`` (CUDA int32 scalar) with ``per_unit``: only the first valid_units * per_unit poses are evaluated —
    the count lives on the device, the launch covers the buffer's capacity (indexed kernel only; the brute-force
    kernel evaluates the whole buffer, rows past the count are ignored by the caller)."""
    lib = _lib.load()
    if index is not None:
        if not poses.is_cuda or poses.dtype != torch.float64 or poses.dim() != 2 or poses.shape[1] != 12:
            raise RuntimeError("poses: expected a CUDA (Q,12) float64 tensor")
        if not boxes.is_cuda or boxes.dtype != torch.float64 or boxes.dim() != 2 or boxes.shape[1] != 6:
            raise RuntimeError("boxes: expected a CUDA (NB,6) float64 tensor")
        poses, boxes = poses.contiguous(), boxes.contiguous()
        Q, NB = poses.shape[0], boxes.shape[0]
        counts = torch.empty(Q, NB, device=poses.device, dtype=torch.int32)
        if Q == 0:
            return counts
        c = index.cloud
        with _lib.device_guard(c.device):
            _lib.check(lib.pngpd_hand_box_counts_indexed_n(_p(c), int(c.dtype == torch.float64), index.P,
                                                           _p(index.spheres), index.C, _p(poses), Q, _p(boxes), NB,
                                                           _p(valid_units) if valid_units is not None else None,
                                                           int(per_unit), _p(counts), _stream(c)),
                       "hand_box_counts_indexed")
        return counts
    cloud = _check_cloud(cloud)
    if not poses.is_cuda or poses.dtype != torch.float64 or poses.dim() != 2 or poses.shape[1] != 12:
        raise RuntimeError("poses: expected a CUDA (Q,12) float64 tensor")
    if not boxes.is_cuda or boxes.dtype != torch.float64 or boxes.dim() != 2 or boxes.shape[1] != 6:
        raise RuntimeError("boxes: expected a CUDA (NB,6) float64 tensor")
    poses, boxes = poses.contiguous(), boxes.contiguous()
    Q, NB = poses.shape[0], boxes.shape[0]
    counts = torch.empty(Q, NB, device=cloud.device, dtype=torch.int32)
    if Q == 0:
        return counts
    with _lib.device_guard(cloud.device):
        _lib.check(lib.pngpd_hand_box_counts(_p(cloud), int(cloud.dtype == torch.float64), cloud.shape[0], _p(poses),
                                             Q, _p(boxes), NB, _p(counts), _stream(cloud)), "hand_box_counts")
    return counts


def pushin_sweep(index, poses2, total, L, R, S, boxes, min_open, found, sfirst, tol=1e-9, stats=None):
    """All push-in poses of every potential grasp in one launch (``pngpd_gpg_pushin_sweep``): poses2 (L*R*S*2,12) from
    ``pngpd_gpg_pushin``, total (1) int32 on the device -> found / sfirst (L*R) int32 filled in place, identical to
    ``hand_box_counts(..., index=index)`` + the first-accept rule of ``pngpd_gpg_finish``."""
    lib = _lib.load()
    c = index.cloud
    with _lib.device_guard(c.device):
        _lib.check(lib.pngpd_gpg_pushin_sweep(_p(c), int(c.dtype == torch.float64), index.P, _p(index.spheres), index.C,
                                              _p(poses2), _p(total), int(L), int(R), int(S), _p(boxes), int(min_open),
                                              float(tol), _p(found), _p(sfirst),
                                              _p(stats) if stats is not None else None, _stream(c)), "gpg_pushin_sweep")


def sweep_select(index, poses, ab, L, R, D, boxes, prm, tol=1e-9, want_masks=False, stats=None, ibuf=None):
    """The lateral sweep + selection of all (sample point, rotation) units in one launch
    (``pngpd_gpg_sweep_select``): index = the scene's ``CloudIndex``; poses (L*R*D,12), ab (L*R,6) from
    ``pngpd_gpg_enumerate``; boxes (4,6); prm the sampler's parameter block.
    -> flag, dsel, list (L*R) int32, total (1) int32 [, masks (L*R,2) int32: opening / collision bits per offset] —
    identical to ``hand_box_counts(..., index=index)`` + ``pngpd_gpg_select``."""
    lib = _lib.load()
    cap = L * R
    dev = poses.device
    if ibuf is None:
        ibuf = torch.empty(3 * cap + 1, device=dev, dtype=torch.int32)
    flag, dsel, plist, total = ibuf[:cap], ibuf[cap:2 * cap], ibuf[2 * cap:3 * cap], ibuf[3 * cap:3 * cap + 1]
    masks = torch.empty(cap, 2, device=dev, dtype=torch.int32) if want_masks else None
    c = index.cloud
    with _lib.device_guard(dev):
        _lib.check(lib.pngpd_gpg_sweep_select(_p(c), int(c.dtype == torch.float64), index.P, _p(index.spheres), index.C,
                                              _p(poses), _p(ab), int(L), int(R), int(D), _p(boxes), _p(prm), float(tol),
                                              _p(flag), _p(dsel), _p(plist), _p(total),
                                              _p(masks) if masks is not None else None,
                                              _p(stats) if stats is not None else None, _stream(c)),
                   "gpg_sweep_select")
    return (flag, dsel, plist, total, masks) if want_masks else (flag, dsel, plist, total)


# ------------------------------------------------------------------------------------------------
# the sampler
# ------------------------------------------------------------------------------------------------
class GpgGraspSamplerPcl:
    """Drop-in for ``dexnet.grasping.GpgGraspSamplerPcl``: ``sample_grasps(point_cloud, points_for_sample,
    all_normal, num_grasps, max_num_samples)`` -> list of ``[bottom_center, approach, binormal, minor,
    bottom_center_modified]`` (five (3,) float64 arrays per grasp; :1616-1618) in the reference's order.

    gripper: dict or RobotGripper-like object with hand_outer_diameter, finger_width, hand_depth, hand_height,
    init_bite (default: robotiq_85).  ``config`` is accepted for signature compatibility and unused, as in the
    Pcl sampler."""

    def __init__(self, gripper=None, config=None, device=None, use_index=True, batch_samples=4096, fused_sweep=True,
                 eig="device"):
        # sphere-culled collision kernel (identical counts).  False = brute force (debugging): it cannot skip the unused
        # tail of the capacity-sized push-in buffer and is slower even on 3,000-point clouds (3.4 vs 2.6 ms per scene).
        self.use_index = bool(use_index)
        # lateral sweep + selection per (sample point, rotation) in one launch (pngpd_gpg_sweep_select; needs the index).
        # False = one wave per pose (pngpd_hand_box_counts_indexed) + pngpd_gpg_select: same flag / dsel, ~5x the work.
        self.fused_sweep = bool(fused_sweep) and self.use_index
        self.batch_samples = int(batch_samples)   # sample points per device round (399 poses each; bounds host memory)
        # where np.linalg.eig(M) (:1493) runs.  "device" (default): pngpd_gpg_frames — LAPACK's DGEEV restated for 3x3
        # (same eigenvalue order and eigenvector signs), a round is then ONE uninterrupted device chain: no download of
        # M, no host thread pool, no upload of the frames.  "lapack": the library call itself on the host, as rounds 1-5.
        if eig not in ("device", "lapack"):
            raise ValueError("eig: 'device' or 'lapack'")
        self.eig = eig
        self.gripper = gripper if gripper is not None else ROBOTIQ_85
        self.config = config
        self.device = torch.device(device) if device is not None else None
        self.last_stats = {}
        self._const_cache = {}
        self._prm_dev = {}                        # the sweep parameter block on the device (eig="device": nothing is uploaded per round)
        self.sweep_stats = None                   # a CUDA int64 (4,) tensor: pngpd_gpg_sweep_select adds its diagnostics
        self.pushin_stats = None                  # the same for pngpd_gpg_pushin_sweep
        self.profile = None                       # set to {} to collect per-stage times (synchronising; diagnostics only)

    # -- device work for one batch of draws --------------------------------------------------
    def _constants(self, g, dev):
        """Gripper-derived constants (hand boxes on the device, sweep parameter block): pure functions of the gripper
        dict, built once per sampler and device instead of once per scene."""
        key = (tuple(sorted(g.items())), str(dev))
        c = self._const_cache.get(key)
        if c is None:
            prm, R, D, S = self._params(g)
            c = self._const_cache[key] = (torch.from_numpy(hand_boxes(g)).to(dev), prm, R, D, S)
            self._prm_dev[key] = torch.from_numpy(prm).to(dev)
        return c

    def _tick(self, name, dev):
        """Stage timing for tools/bench_gpg_scale.py: with ``self.profile`` a dict, every stage boundary synchronises and
        adds the elapsed host time to its entry (None = start a batch).  Off (the default): no-op."""
        if self.profile is None:
            return
        import time
        torch.cuda.synchronize(dev)
        now = time.perf_counter()
        if name is not None:
            self.profile[name] = self.profile.get(name, 0.0) + (now - self._t_last)
        self._t_last = now

    def _pinned(self, n):
        """A pinned host staging buffer of >= n doubles from a thread-local pool (page-locking costs milliseconds; samplers
        are short-lived — scoring.detect_grasps builds one per scene — so the pool outlives them).  Several rounds are in
        flight at once, each with its own buffers; ``_unpin`` hands a buffer back once its copy has been consumed."""
        pool = _pin_pool()
        best = -1
        for i, t in enumerate(pool):                         # best fit: a small request must not take the big buffer
            if t.numel() >= n and (best < 0 or t.numel() < pool[best].numel()):
                best = i
        if best >= 0:
            return pool.pop(best)
        return torch.empty(max(n + n // 8, 1 << 15), dtype=torch.float64).pin_memory()

    @staticmethod
    def _unpin(t):
        if t is not None:
            _pin_pool().append(t)

    @staticmethod
    def _params(g):
        """Gripper / sweep constants of the device kernels, computed with numpy exactly as the reference computes them
        (layout: pngpd_gpg.hip)."""
        hh, fw, hd = g["hand_height"], g["finger_width"], g["hand_depth"]
        ow = g["hand_outer_diameter"] - fw * 2
        dth = np.arange(-RANGE_DTHETA, RANGE_DTHETA + 1, DTHETA).astype(np.float64) / 180 * np.pi    # :1524-1529
        dys = np.arange(-NUM_DY * fw, (NUM_DY + 1) * fw, fw)                                         # :1531
        S = int(hd / APPROACH_STEP)                                                                  # :1576
        if len(dth) > 32 or len(dys) > 32 or S > 64:
            raise RuntimeError("sweep larger than the kernels' parameter block")
        prm = np.zeros(160)
        prm[0:13] = [g["init_bite"], hd, hd * 0.5, APPROACH_STEP, TABLE_CLEARANCE, hh * 0.5, -(hh * 0.5), -(ow * 0.5),
                     ow * 0.5, -fw, fw, -hh, 3.0]
        prm[16:16 + len(dth)] = dth
        prm[48:48 + len(dys)] = dys
        prm[80:80 + S] = np.arange(S, dtype=np.float64)
        return prm, len(dth), len(dys), S

    # -- one round of draws = three stages, so that consecutive rounds overlap (see sample_grasps) --------------------
    def _stage_moments(self, g, cloud_d, normals_d, sel_pts, normals_at_ind, scene):
        """[device] r-ball / 100-NN moment matrices of the round's sample points + their download (asynchronous)."""
        dev = cloud_d.device
        K = sel_pts.shape[0]
        if self.use_index and scene.get("index") is None:
            scene["index"] = CloudIndex(cloud_d)          # once per scene: ~0.4 ms of device work, nobody waits for it
        index = scene.get("index")
        self._tick(None, dev)
        fw, hd = g["finger_width"], g["hand_depth"]
        r_ball = max(g["hand_outer_diameter"] - fw, hd, g["hand_height"] / 2.0)                  # :1464
        # uploads go through pinned staging buffers: a pageable copy would block the host until the device has drained
        # everything queued before it — i.e. the previous round's whole chain — and serialise the pipeline
        dev_eig = self.eig == "device"
        nq = K * (6 if dev_eig else 3)
        q_h = self._pinned(nq)
        q_h.numpy()[:K * 3] = np.ascontiguousarray(sel_pts).reshape(-1)      # numpy's memcpy: see _stage_chain's upload
        if dev_eig:
            q_h.numpy()[K * 3:K * 6] = np.ascontiguousarray(normals_at_ind).reshape(-1)
        q2_d = torch.empty(nq // 3, 3, device=dev, dtype=torch.float64)
        q2_d.view(-1).copy_(q_h[:nq], non_blocking=True)
        q_d = q2_d[:K]
        M_d, _ = normal_moments(cloud_d, normals_d, q_d, r_ball, MAX_NN, index=index)
        if dev_eig:
            # :1486-1506 on the device: the round stays one device chain (no M download, no host eig, no frame upload)
            frames_d, flags_d = local_frames(M_d, q2_d[K:], q_d)
            return dict(K=K, sel_pts=sel_pts, normals_at_ind=normals_at_ind, q_h=q_h, frames_d=frames_d, flags_d=flags_d)
        M_h = self._pinned(K * 9)
        M_h[:K * 9].copy_(M_d.view(-1), non_blocking=True)                                      # download 1: K x 9 doubles
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        return dict(K=K, sel_pts=sel_pts, normals_at_ind=normals_at_ind, M_h=M_h, M_d=M_d, ev_m=ev, q_h=q_h)

    def _stage_chain(self, rd, g, cloud_d, scene):
        """[host] local frames of the round (np.linalg.eig, exactly as the reference calls it), then [device] everything
        from the pose enumeration to the packed result and its download (asynchronous).  Fills rd["m_zero"]."""
        dev = cloud_d.device
        K, sel_pts, normals_at_ind = rd["K"], rd["sel_pts"], rd["normals_at_ind"]
        tick = self._tick
        boxes_d, prm, R, D, S = self._constants(g, dev)
        if self.use_index and scene.get("index") is None:
            scene["index"] = CloudIndex(cloud_d)          # enqueued behind the first moments kernel; the host does not wait
        index = scene.get("index")
        if self.eig == "device":
            tick("moments+frames", dev)
            L = K
            rd["live"], rd["L"], rd["m_zero"] = np.arange(K), K, None      # the flags arrive with the packed result
            frames_d = rd["frames_d"].view(-1)
            up_d = frames_d
            prm_d = self._prm_dev[(tuple(sorted(g.items())), str(dev))]
            self._chain_device(rd, up_d, frames_d, prm_d, L, R, D, S, boxes_d, index, cloud_d, dev)
            return
        rd["ev_m"].synchronize()
        tick("moments+download", dev)
        M = rd["M_h"][:K * 9].numpy().reshape(K, 3, 3).copy()
        self._unpin(rd.pop("M_h")); rd.pop("M_d"); self._unpin(rd.pop("q_h"))
        m_zero = M.sum((1, 2)) == 0                                                             # :1486
        rd["m_zero"] = m_zero
        live = np.nonzero(~m_zero)[0]
        rd["live"], rd["L"] = live, live.size
        if live.size == 0:
            return
        # local frames (:1493-1512) — np.linalg.eig exactly as the reference calls it, once for the whole round
        eigval, eigvec = _batched_eig(M[live])
        eigval, eigvec = np.real(eigval), np.real(eigvec)
        ar = np.arange(live.size)
        minor = _unit(eigvec[ar, :, np.argmin(eigval, 1)])
        normal = _unit(eigvec[ar, :, np.argmax(eigval, 1)])
        major = np.cross(minor, normal)
        nm = np.linalg.norm(major, axis=1, keepdims=True)
        major = np.where(nm != 0, major / np.where(nm != 0, nm, 1.0), major)
        flip = (normals_at_ind[live] * normal).sum(1) < 0
        normal = np.where(flip[:, None], -normal, normal)
        minor = np.where(flip[:, None], -minor, minor)
        L = live.size
        up = np.concatenate([np.concatenate([minor, normal, major, sel_pts[live]], 1).reshape(-1), prm])
        tick("host eig+frames", dev)
        up_h = self._pinned(up.size)
        # host-side staging copies go through numpy, not torch: torch parallelises a CPU copy above 32,768 elements over its
        # OpenMP pool (128 threads on the GPU box) whose workers then spin — under the container's 16-CPU cgroup quota that
        # exhausted the period and throttled the whole process for ~80 ms, once per call, at rounds of >= 4,096 sample points
        up_h.numpy()[:up.size] = up
        up_d = torch.empty(up.size, device=dev, dtype=torch.float64)
        up_d.copy_(up_h[:up.size], non_blocking=True)                                           # upload: frames + constants
        rd["up_h"] = up_h
        frames_d, prm_d = up_d[:L * 12], up_d[L * 12:]
        self._chain_device(rd, up_d, frames_d, prm_d, L, R, D, S, boxes_d, index, cloud_d, dev)

    def _chain_device(self, rd, up_d, frames_d, prm_d, L, R, D, S, boxes_d, index, cloud_d, dev):
        """[device] pose enumeration -> lateral sweep + selection -> push-in -> packed result and its download."""
        tick = self._tick
        K = rd["K"]
        nflag = K if self.eig == "device" else 0
        cap = L * R
        f64, i32 = torch.float64, torch.int32
        nres = 1 + L + cap * 15 + 1
        ws = _ws_ring(dev)[rd["slot"] % 4]
        ws.reset(dev, 8 * (cap * D * 12 + cap * 6 + cap * S * 2 * 12 + 2 * cap * S * 3 + nres + nflag) + 4 * 2 * (3 * cap + 1) + 4096)
        poses = ws.take((cap * D, 12), f64)
        ab = ws.take((cap, 6), f64)
        _call("pngpd_gpg_enumerate", up_d, frames_d, L, R, D, prm_d, poses, ab)
        tick("upload+enumerate", dev)
        if self.fused_sweep and index is not None:
            flag, dsel, plist, total = sweep_select(index, poses, ab, L, R, D, boxes_d, prm_d, stats=self.sweep_stats,
                                                    ibuf=ws.take((3 * cap + 1,), i32))
        else:
            cnt = hand_box_counts(cloud_d, poses, boxes_d, index=index)                         # (L*R*D,4)
            ibuf = ws.take((3 * cap + 1,), i32)
            flag, dsel, plist, total = ibuf[:cap], ibuf[cap:2 * cap], ibuf[2 * cap:3 * cap], ibuf[3 * cap:]
            _call("pngpd_gpg_select", up_d, cnt, poses, ab, L, R, D, prm_d, flag, dsel, plist, total)
        tick("sweep+select", dev)
        poses2 = ws.take((cap * S * 2, 12), f64)
        bm = ws.take((2 * cap * S, 3), f64)
        back, mod = bm[:cap * S], bm[cap * S:]
        _call("pngpd_gpg_pushin", up_d, plist, total, dsel, poses, ab, frames_d, L, R, D, S, prm_d, poses2, back, mod)
        jbuf = ws.take((3 * cap + 1,), i32)
        found, sfirst, olist, ototal = jbuf[:cap], jbuf[cap:2 * cap], jbuf[2 * cap:3 * cap], jbuf[3 * cap:]
        if self.fused_sweep and index is not None and S <= 32:
            pushin_sweep(index, poses2, total, L, R, S, boxes_d, MIN_OPEN_POINTS, found, sfirst, stats=self.pushin_stats)
            cnt2 = None                                    # pngpd_gpg_finish then starts from found / sfirst
        else:
            cnt2 = hand_box_counts(cloud_d, poses2, boxes_d, index=index, valid_units=total, per_unit=2 * S)
        tick("pushin+sweep2", dev)
        out = ws.take((nres + nflag,), f64)
        _call("pngpd_gpg_finish", up_d, cnt2, plist, total, ab, frames_d, back, mod, L, R, S, MIN_OPEN_POINTS, found,
              sfirst, olist, ototal, out)
        out[nres - 1:nres].copy_(total)     # the potential-grasp count rides in the same download (it was a third sync)
        if nflag:
            out[nres:].copy_(rd["flags_d"])  # eig="device": so do the per-point flags (zero moment matrix, complex pair)
        host_t = self._pinned(nres + nflag)
        host_t[:nres + nflag].copy_(out, non_blocking=True)                                     # download 2: packed result
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        rd.update(host_t=host_t, nres=nres, out_d=out, ev_r=ev)

    def _stage_collect(self, rd, dev):
        """[host] wait for the round's packed result -> (m_zero (K,) bool, counts (K,) grasps per draw, grasps (n,5,3) rows
        of all draws in draw order)."""
        K, m_zero = rd["K"], rd["m_zero"]
        counts = np.zeros(K, dtype=np.int64)
        if rd["L"] == 0:
            return m_zero, counts, np.zeros((0, 5, 3))
        rd["ev_r"].synchronize()
        self._tick("finish+download", dev)
        L, live = rd["L"], rd["live"]
        host = rd["host_t"][:rd["nres"]].numpy()
        if m_zero is None:                                      # eig="device": the frame flags came with the result
            fl = rd["host_t"][rd["nres"]:rd["nres"] + K].numpy().astype(np.int64)
            m_zero = (fl & 1) != 0
            if (fl & 12).any():
                # (np.linalg.eig raises LinAlgError on non-finite input too; DGEEV's rescaling / DLAQR0 paths are not restated)
                raise RuntimeError(f"pngpd_gpg_frames: {int(((fl & 8) != 0).sum())} moment matrices are non-finite (NaN "
                                   f"normals?) or outside DGEEV's unscaled range, {int(((fl & 4) != 0).sum())} did not "
                                   f"converge in 300 QR sweeps; eig='lapack' runs the library itself")
            self.last_stats["eig_complex_pairs"] = self.last_stats.get("eig_complex_pairs", 0) + int(((fl & 2) != 0).sum())
            rd.pop("frames_d", None); rd.pop("flags_d", None); self._unpin(rd.pop("q_h", None))
        self.last_stats["potential"] = self.last_stats.get("potential", 0) + int(host[-1])
        n = int(host[0])
        counts[live] = host[1:1 + L].astype(np.int64)
        # rows are packed in (live sample point, rotation) order = draw order: no per-draw regrouping needed
        grasps = host[1 + L:1 + L + n * 15].reshape(n, 5, 3).copy()
        self._unpin(rd.pop("host_t")); rd.pop("out_d"); self._unpin(rd.pop("up_h", None))
        return m_zero, counts, grasps

    def sample_grasps(self, point_cloud, points_for_sample, all_normal, num_grasps=20, max_num_samples=200,
                      show_final_grasp=False, sample_indices=None, seed=None, as_array=False, scene_index=None,
                      **kwargs):
        chunks = list(self.iter_rounds(point_cloud, points_for_sample, all_normal, num_grasps, max_num_samples,
                                       sample_indices=sample_indices, seed=seed, scene_index=scene_index))
        out = np.concatenate(chunks, 0) if chunks else np.zeros((0, 5, 3))
        if as_array:
            return out
        return [[v.copy() for v in gr] for gr in out]

    def iter_rounds(self, point_cloud, points_for_sample, all_normal, num_grasps=20, max_num_samples=200,
                    sample_indices=None, seed=None, scene_index=None):
        """``sample_grasps`` as a generator: yields the kept grasps of every round, (n,5,3) float64 in draw order, as
        soon as the round's packed result has been downloaded — while the device already works on the next round's
        chain (the rounds' own pipeline below).  A consumer that enqueues each chunk's crop + scoring on ANOTHER
        stream (``scoring.GraspScorer.score_chunks``) overlaps scoring with sampling (kinect2grasp.py:141-150 feeding
        :443-514); concatenating the chunks is exactly ``sample_grasps(..., as_array=True)``."""
        g = _gripper_dict(self.gripper)
        self.last_stats = {"draws": 0, "sampled": 0, "potential": 0}
        if isinstance(point_cloud, torch.Tensor):
            cloud_d = point_cloud if self.device is None else point_cloud.to(self.device)
        else:
            pc = point_cloud.to_array() if hasattr(point_cloud, "to_array") else np.asarray(point_cloud)
            if pc.dtype not in (np.float32, np.float64):
                pc = pc.astype(np.float64)
            dev = self.device or torch.device("cuda", torch.cuda.current_device())
            cloud_d = torch.from_numpy(np.ascontiguousarray(pc)).to(dev)
        if not cloud_d.is_cuda:
            raise RuntimeError("GpgGraspSamplerPcl runs on the GPU: pass a CUDA cloud or construct with device='cuda'")
        dev = cloud_d.device
        all_normal = np.asarray(all_normal.cpu() if isinstance(all_normal, torch.Tensor) else all_normal, dtype=np.float64)
        pfs = np.asarray(points_for_sample.cpu() if isinstance(points_for_sample, torch.Tensor) else points_for_sample,
                         dtype=np.float64).reshape(-1, 3)
        normals_d = torch.from_numpy(np.ascontiguousarray(all_normal)).to(dev)
        scene = {"index": scene_index if self.use_index else None}       # CloudIndex: built once per scene, lazily
        if num_grasps <= 0 or max_num_samples <= 0 or pfs.shape[0] == 0:                       # :1432 loop never entered
            return
        rng = np.random.default_rng(seed)
        explicit = None if sample_indices is None else np.asarray(sample_indices, dtype=np.int64).reshape(-1)
        # Rounds of up to ``batch_samples`` draws run as a three-stage pipeline on ONE stream, two rounds ahead
        # (eig="lapack"; with eig="device" the host's eig(k) slot is empty — frames(k) runs right behind mom(k) on the
        # device — and the same schedule simply keeps two rounds of device work queued ahead of the host's collect):
        #     device:  mom(0) mom(1) | chain(0) mom(2) | chain(1) mom(3) | ...
        #     host:                    eig(0)          | eig(1) collect(0) | eig(2) collect(1) | ...
        # The moments of round k+2 are enqueued behind chain(k), so their download is complete when the host turns to
        # eig(k+1) — which then runs while the device works on chain(k): neither side waits for the other, where the
        # serial schedule alternated (host eig with an idle device, then the device chain with an idle host).  Rounds are
        # sized as if no draw of the rounds in flight had a zero moment matrix (those do not count, :1486-1489); the
        # stop rule is applied to the rounds in order, one round behind, and a stop abandons the rounds in flight.
        sampled, found, done = 0, 0, False
        st = dict(pos=0, issued=0, credit=0, rounds=0)       # credit: draws issued but not yet known to count (or not)

        def issue():
            want = min(self.batch_samples, max_num_samples - sampled - st["credit"])
            if want <= 0:
                return None
            if explicit is not None:
                draws = explicit[st["pos"]:st["pos"] + want]
                if draws.size == 0:
                    return None
            else:
                if st["issued"] >= 10 * max_num_samples:
                    return None                                      # degenerate cloud: the reference would spin here
                draws = rng.integers(0, pfs.shape[0], size=want)
            st["pos"] += draws.size; st["issued"] += draws.size; st["credit"] += draws.size
            rd = self._stage_moments(g, cloud_d, normals_d, pfs[draws], all_normal[draws], scene)
            rd["draws"], rd["slot"] = draws, st["rounds"]
            st["rounds"] += 1
            return rd

        lookahead = 0 if self.profile is not None else 2             # the stage profile wants one round at a time
        pending, chained = collections.deque(), collections.deque()
        for _ in range(max(1, lookahead)):
            rd = issue()
            if rd is not None:
                pending.append(rd)
        try:
            while (pending or chained) and not done:
                issued_here = False
                if pending:
                    rd = pending.popleft()
                    self._stage_chain(rd, g, cloud_d, scene)
                    chained.append(rd)
                if len(chained) > (1 if pending and lookahead else 0):
                    rd = chained.popleft()
                    m_zero, counts, grasps = self._stage_collect(rd, dev)
                    # the reference's loop over the draws (:1486-1489, :1639), without a Python iteration per draw: a draw
                    # whose M is zero consumes a draw but is not counted; stop behind the first counted draw at which
                    # num_grasps grasps have been found or max_num_samples sample points have been processed
                    st["credit"] -= rd["K"]
                    live_cum = sampled + np.cumsum(~m_zero)
                    found_cum = found + np.cumsum(counts)
                    stop = np.nonzero(~m_zero & ((found_cum >= num_grasps) | (live_cum >= max_num_samples)))[0]
                    last = int(stop[0]) if stop.size else rd["K"] - 1
                    keep = int(found_cum[last] - found)
                    self.last_stats["draws"] += last + 1
                    sampled, found = int(live_cum[last]), int(found_cum[last])
                    done = bool(stop.size)
                    self.last_stats["sampled"] = sampled
                    if not done:
                        # the next round is issued BEFORE the consumer sees this one: its moments kernel is queued while
                        # the consumer enqueues the chunk's scoring
                        rd2 = issue()
                        if rd2 is not None:
                            pending.append(rd2)
                        issued_here = True
                    if keep:
                        yield grasps[:keep]
                    if not done and issued_here:
                        continue
                if not done:
                    rd = issue()
                    if rd is not None:
                        pending.append(rd)
        finally:
            # (also when the consumer closes the generator early)
            if pending or chained:
                # rounds in flight were abandoned by the stop rule: their copies must land before the buffers are reused
                torch.cuda.current_stream(dev).synchronize()
                for rd in list(pending) + list(chained):
                    for key in ("M_h", "host_t", "q_h", "up_h"):
                        self._unpin(rd.get(key))
            self.last_stats["sampled"] = sampled
