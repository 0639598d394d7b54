"""CPU oracle (numpy, fp64) for the GPG grasp-candidate sampler.

TEST INFRASTRUCTURE ONLY — see the header of ``oracle/pointnet_oracle.py``: only ``tests/``, ``smoke()`` and
``bench.py``'s CPU-baseline leg may import this; the product (``pointnetgpd_amd/gpg.py``) never does.

Restates (citations relative to /root/reference/dex-net/src/dexnet/grasping/grasp_sampler.py):

* ``GpgGraspSamplerPcl.sample_grasps``   :1383-1656  (the sampler ``kinect2grasp.py:150`` calls)
* ``GraspSampler.get_hand_points``       :287-321
* ``GraspSampler.check_collision_square`` :336-393
* ``GraspSampler.check_collide``          :405-421

Third-party pieces that are absent from /root/reference (``requirements.txt:18`` ``autolab-core``, un-pinned;
``open3d``, un-pinned) and restated from their published algorithms:

* ``autolab_core.RigidTransform.rotation_from_quaternion(q_wxyz)`` = Gohlke's ``quaternion_matrix`` on the
  NORMALISED quaternion.  The reference passes ``[dtheta_rad, minor_x, minor_y, minor_z]`` (:1503) — an angle in
  the scalar slot of an un-normalised quaternion — so the 19 "rotations" are by ``2*atan2(1, dtheta_rad)`` about
  the minor axis (180 deg at dtheta = 0), not by dtheta.  Reproduced as is.
* ``open3d.geometry.KDTreeFlann.search_hybrid_vector_3d(query, radius, max_nn)``: the (up to) ``max_nn`` nearest
  points with squared distance < radius^2, ascending.  The reference indexes its result as 2-D ``[0, i]`` arrays
  (:1475-1480, the pcl-era shape); the stand-in used to run the reference here returns that shape.

Parity pin: ``oracle/make_golden_gpg.py`` executes the UNMODIFIED reference file in the build container with
those two modules (and the unused dexnet/mayavi/rospy imports) stubbed, feeding the sample-point draws through
``np.random.seed``; inputs, draws and the returned grasps are committed under ``tests/golden/gpg_*.npz``.

Deliberate difference: the reference normalises ``all_normal`` rows IN PLACE while it accumulates M (:1478-1483,
a view of the caller's array); the oracle leaves the caller's array untouched (values differ by <= 1 ulp).
"""
from __future__ import annotations

import numpy as np

# dex-net/data/grippers/robotiq_85/params.json
ROBOTIQ_85 = dict(hand_outer_diameter=0.218, finger_width=0.0255, hand_depth=0.125, hand_height=0.030,
                  init_bite=0.01)

NUM_DY = 10            # params['num_dy']        :1413
DTHETA = 10            # params['dtheta']        :1414
RANGE_DTHETA = 90      # params['range_dtheta']  :1415
APPROACH_STEP = 0.005  # params['approach_step'] :1418
MAX_NN = 100           # :1475
WAYS = ("p_open", "p_left", "p_right", "p_bottom")


def rotation_from_quaternion(q_wxyz):
    """autolab_core.RigidTransform.rotation_from_quaternion -> transformations.quaternion_matrix (xyzw)."""
    q = np.array([q_wxyz[1], q_wxyz[2], q_wxyz[3], q_wxyz[0]], dtype=np.float64)
    nq = np.dot(q, q)
    if nq < np.finfo(float).eps * 4.0:
        return np.identity(3)
    q *= np.sqrt(2.0 / nq)
    q = np.outer(q, q)
    return np.array([[1.0 - q[1, 1] - q[2, 2], q[0, 1] - q[2, 3], q[0, 2] + q[1, 3]],
                     [q[0, 1] + q[2, 3], 1.0 - q[0, 0] - q[2, 2], q[1, 2] - q[0, 3]],
                     [q[0, 2] - q[1, 3], q[1, 2] + q[0, 3], 1.0 - q[0, 0] - q[1, 1]]])


def hand_points(g, bottom_center, approach, binormal):
    """:287-321.  21 rows: row 0 is the origin, rows 1..20 the corners p1..p20 of the hand model."""
    hh, fw, hod, hd = g["hand_height"], g["finger_width"], g["hand_outer_diameter"], g["hand_depth"]
    open_w = hod - fw * 2
    minor = np.cross(approach, binormal)
    minor = minor / np.linalg.norm(minor)
    p5_p6 = minor * hh * 0.5 + bottom_center
    p7_p8 = -minor * hh * 0.5 + bottom_center
    p5 = -binormal * open_w * 0.5 + p5_p6
    p6 = binormal * open_w * 0.5 + p5_p6
    p7 = binormal * open_w * 0.5 + p7_p8
    p8 = -binormal * open_w * 0.5 + p7_p8
    p1, p2, p3, p4 = (approach * hd + p for p in (p5, p6, p7, p8))
    p9, p10, p11, p12 = (-binormal * fw + p for p in (p1, p4, p5, p8))
    p13, p14, p15, p16 = (binormal * fw + p for p in (p2, p3, p6, p7))
    p17, p18, p19, p20 = (-approach * hh + p for p in (p11, p15, p16, p12))
    return np.vstack([np.zeros(3), p1, p2, p3, p4, p5, p6, p7, p8, p9, p10, p11, p12, p13, p14, p15, p16,
                      p17, p18, p19, p20])


def way_box(p, way):
    """:361-377.  Strict bounds (x_lo, x_hi, y_lo, y_hi, z_lo, z_hi) of one hand region in the grasp frame."""
    s1, s2, s4, s8 = {"p_open": (p[1], p[2], p[4], p[8]), "p_left": (p[9], p[1], p[10], p[12]),
                      "p_right": (p[2], p[13], p[3], p[7]), "p_bottom": (p[11], p[15], p[12], p[20])}[way]
    return np.array([s8[0], s4[0], s1[1], s2[1], s4[2], s1[2]])


def points_in_way(center, approach, binormal, minor, points, p, way):
    """check_collision_square :336-393 -> indices of the cloud points inside region ``way``."""
    a = approach.reshape(1, 3) / np.linalg.norm(approach)
    b = binormal.reshape(1, 3) / np.linalg.norm(binormal)
    m = minor.reshape(1, 3) / np.linalg.norm(minor)
    grasp_matrix = np.hstack([a.T, b.T, m.T]).T
    pg = np.dot(grasp_matrix, (points - center.reshape(1, 3)).T).T
    bx = way_box(p, way)
    inside = ((bx[2] < pg[:, 1]) & (bx[3] > pg[:, 1]) & (bx[5] > pg[:, 2]) & (bx[4] < pg[:, 2]) &
              (bx[1] > pg[:, 0]) & (bx[0] < pg[:, 0]))
    return np.where(inside)[0]


def collides(center, approach, binormal, minor, points, p):
    """check_collide :405-421: any cloud point in the bottom plate or either finger."""
    return any(len(points_in_way(center, approach, binormal, minor, points, p, w)) > 0
               for w in ("p_bottom", "p_left", "p_right"))


def neighbours(points, query, radius, max_nn=MAX_NN):
    """search_hybrid_vector_3d stand-in: (indices, squared distances), ascending, at most ``max_nn``."""
    d = points - query.reshape(1, 3)
    d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
    idx = np.where(d2 < radius * radius)[0]
    idx = idx[np.argsort(d2[idx], kind="stable")][:max_nn]
    return idx, d2[idx]


def normal_moment(points, normals, query, radius):
    """:1471-1485.  M = sum of n n^T over the neighbours with non-zero distance."""
    M = np.zeros((3, 3))
    idx, d2 = neighbours(points, query, radius)
    for i, dd in zip(idx, d2):
        if dd != 0:
            n = np.array(normals[i], dtype=np.float64).reshape(-1, 1)
            if np.linalg.norm(n) != 0:
                n = n / np.linalg.norm(n)
            M += np.matmul(n, n.T)
    return M


def local_frame(M, normal_at_ind):
    """:1493-1508 -> (new_normal, major_pc, minor_pc)."""
    eigval, eigvec = np.linalg.eig(M)
    minor_pc = eigvec[:, np.argmin(eigval)].reshape(3)
    minor_pc = minor_pc / np.linalg.norm(minor_pc)
    new_normal = eigvec[:, np.argmax(eigval)].reshape(3)
    new_normal = new_normal / np.linalg.norm(new_normal)
    major_pc = np.cross(minor_pc, new_normal)
    if np.linalg.norm(major_pc) != 0:
        major_pc = major_pc / np.linalg.norm(major_pc)
    if np.dot(np.asarray(normal_at_ind).reshape(3), new_normal) < 0:
        new_normal = -new_normal
        minor_pc = -minor_pc
    return new_normal, major_pc, minor_pc


def sample_grasps(points, points_for_sample, all_normal, sample_indices, num_grasps=20, max_num_samples=200,
                  gripper=ROBOTIQ_85, trace=None):
    """:1383-1656 with the per-iteration ``np.random.choice`` draw (:1456) replaced by ``sample_indices`` (consumed
    in order; the reference reseeds from the OS every iteration, :1455, so it has no reproducible stream of its
    own).  Returns the list of ``[bottom_center, approach, binormal, minor, bottom_center_modified]``.
    ``trace`` (a list) receives one dict per consumed draw."""
    g = gripper
    points = np.asarray(points)
    hp = hand_points(g, np.array([0, 0, 0]), np.array([1, 0, 0]), np.array([0, 1, 0]))
    r_ball = max(g["hand_outer_diameter"] - g["finger_width"], g["hand_depth"], g["hand_height"] / 2.0)  # :1464
    fw, hd = g["finger_width"], g["hand_depth"]
    out = []
    sampled = 0
    for ind in sample_indices:
        if not (len(out) < num_grasps and sampled < max_num_samples):
            break
        sel = np.asarray(points_for_sample[ind]).reshape(3)
        M = normal_moment(points, all_normal, sel, r_ball)
        rec = dict(ind=int(ind), M=M.copy(), potential=0, found=0)
        if trace is not None:
            trace.append(rec)
        if sum(sum(M)) == 0:                                                                     # :1486
            continue
        new_normal, major_pc, minor_pc = local_frame(M, all_normal[ind])
        rec.update(normal=new_normal, major=major_pc, minor=minor_pc)
        potential = []
        for dtheta in np.arange(-RANGE_DTHETA, RANGE_DTHETA + 1, DTHETA):                         # :1524
            x, y, z = minor_pc
            rot = rotation_from_quaternion(np.array([np.float64(dtheta) / 180 * np.pi, x, y, z]))
            dy_ok = []
            for dy in np.arange(-NUM_DY * fw, (NUM_DY + 1) * fw, fw):                             # :1531
                binormal = np.dot(rot, major_pc)
                approach = np.dot(rot, new_normal)
                bottom = sel + binormal * dy
                bottom = g["init_bite"] * (-approach) + bottom                                   # :1539
                has = {w: len(points_in_way(bottom, approach, binormal, minor_pc, points, hp, w)) > 0
                       for w in WAYS}
                if has["p_open"] and not has["p_bottom"] and not has["p_left"] and not has["p_right"]:
                    dy_ok.append([bottom, approach, binormal, minor_pc])
            if dy_ok:
                c = dy_ok[int(np.ceil(len(dy_ok) / 2) - 1)]                                      # :1567
                finger_top = c[0] + c[1] * hd
                if finger_top[2] < c[0][2] - hd * 0.5:                                           # :1572
                    potential.append(c)
        rec["potential"] = len(potential)
        n_before = len(out)
        for bottom0, approach, binormal, minor in potential:
            for s in range(int(hd / APPROACH_STEP)):                                             # :1576
                bottom = approach * s * APPROACH_STEP + bottom0
                if collides(bottom, approach, binormal, minor, points, hp):
                    bottom = bottom + (-approach) * APPROACH_STEP * 3                            # :1589
                    hp_w = hand_points(g, bottom, approach, binormal)[1:]
                    zmin = hp_w[:, 2].min()
                    if zmin < 0.01:                                                              # :1600
                        low = hp_w[np.where(hp_w[:, 2] == zmin)[0][0]]
                        with np.errstate(divide="ignore", invalid="ignore"):
                            tx = -low[2] * approach[0] / approach[2] + low[0]
                            ty = -low[2] * approach[1] / approach[2] + low[1]
                        p_table = np.array([tx, ty, 0])
                        back = np.linalg.norm([low, p_table]) + 0.01                             # :1605 (2x3 Frobenius)
                        bottom_mod = bottom - approach * back
                    else:
                        bottom_mod = bottom
                    n_open = len(points_in_way(bottom_mod, approach, binormal, minor, points, hp, "p_open"))
                    if n_open > 10 and not collides(bottom_mod, approach, binormal, minor, points, hp):
                        out.append([bottom, approach, binormal, minor, bottom_mod])
                        break        # :1625 — ONLY after an accepted grasp; a rejected back-off tries the next step
        rec["found"] = len(out) - n_before
        sampled += 1
    return out


def synth_scene(kind, P, seed):
    """Deterministic table-top object cloud with outward unit normals (+ small noise) for the sampler tests.
    Returns (points (P,3) f32-rounded f64, normals (P,3) f64)."""
    rng = np.random.default_rng(seed)
    if kind == "cylinder":                      # upright can: radius 3 cm, 14 cm tall, standing on z = 0
        r, h = 0.03, 0.14
        n_top = P // 5
        n_side = P - n_top
        th = rng.uniform(0, 2 * np.pi, n_side)
        z = rng.uniform(0.0, h, n_side)
        side = np.stack([r * np.cos(th), r * np.sin(th), z], 1)
        nside = np.stack([np.cos(th), np.sin(th), np.zeros(n_side)], 1)
        rr = r * np.sqrt(rng.uniform(0, 1, n_top)); tt = rng.uniform(0, 2 * np.pi, n_top)
        top = np.stack([rr * np.cos(tt), rr * np.sin(tt), np.full(n_top, h)], 1)
        ntop = np.tile([0.0, 0.0, 1.0], (n_top, 1))
        pts, nrm = np.concatenate([side, top]), np.concatenate([nside, ntop])
    elif kind == "box":                         # 5 x 9 x 11 cm box, slightly rotated about z
        ext = np.array([0.025, 0.045, 0.055])
        face = rng.integers(0, 5, P)            # +-x, +-y, +z
        u = rng.uniform(-1, 1, (P, 3)) * ext
        nrm = np.zeros((P, 3))
        for f, (ax, sg) in enumerate([(0, 1), (0, -1), (1, 1), (1, -1), (2, 1)]):
            mk = face == f
            u[mk, ax] = sg * ext[ax]
            nrm[mk, ax] = sg
        u[:, 2] += ext[2]
        c, s = np.cos(0.4), np.sin(0.4)
        Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
        pts, nrm = u @ Rz.T, nrm @ Rz.T
    elif kind == "ellipsoid":                   # curved everywhere: well separated eigenvalues of M
        ax = np.array([0.028, 0.04, 0.06])
        v = rng.normal(size=(P, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
        v[:, 2] = np.abs(v[:, 2]) * 0.98 - 0.1
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        pts = v * ax + np.array([0, 0, 0.07])
        nrm = v / ax
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    else:
        raise ValueError(kind)
    pts = pts + rng.normal(scale=2e-4, size=pts.shape)
    nrm = nrm + rng.normal(scale=0.05, size=nrm.shape)
    nrm *= rng.uniform(0.5, 1.5, (P, 1))        # un-normalised on purpose (:1481-1482 normalises)
    perm = rng.permutation(P)
    return pts[perm].astype(np.float32).astype(np.float64), nrm[perm]
