"""CPU oracle for the GPD-baseline row of SURVEY.md §8f-4 (projection images, depth registration, GPDClassifier).

TEST INFRASTRUCTURE ONLY (see oracle/pointnet_oracle.py for the rule): only ``tests/`` may import it.

Restates, in vectorised numpy / functional torch (citations relative to /root/reference/):

* ``cal_projection`` / ``project_pc``              PointNetGPD/model/dataset.py:88-198
* ``registerDepthMap``                             PointNetGPD/ycb_cloud_generate.py:60-121
* ``registeredDepthMapToPointCloud`` (unorganised and organised) PointNetGPD/ycb_cloud_generate.py:124-184
* ``GPDClassifier.forward``                        PointNetGPD/model/gpd.py:21-31

Parity pin: ``oracle/make_golden_gpd.py`` EXECUTES the unmodified reference functions in the build container (the
dataset module imported with open3d stubbed and ``get_normal`` — open3d's estimator — replaced by supplied normals;
the two ycb_cloud_generate functions cut out with ``ast``; gpd.py imported as is) and commits
``tests/golden/gpd_*.npz``; ``tests/test_gpd_cpu.py`` checks this restatement against those records everywhere.
"""
import numpy as np

PROJECT_SIZE = 60          # dataset.py:220-222
VOXEL_POINT_NUM = 50       # dataset.py:223
PROJECTION_MARGIN = 1      # dataset.py:224
ORDERS = ((0, 1, 2), (1, 2, 0), (0, 2, 1))   # dataset.py:104,110,113


def cal_projection(points, normals, order, gripper_width, size=PROJECT_SIZE, margin=PROJECTION_MARGIN,
                   vpn=VOXEL_POINT_NUM):
    """dataset.py:139-198.  points/normals (M,3) f64 -> occupy (size,size,1), norm (size,size,3), both f64.

    What the reference's loop amounts to: voxel index = floor(coord/res + size/2) per axis of ``order``; per voxel the
    first ``vpn`` points in input order are kept, their normals are stored as float32 and summed sequentially in
    float32; ``np.unique`` sorts the voxels lexicographically, and the fancy assignment ``pic[x, y] = value`` lets the
    LAST voxel of each (x, y) — the one with the largest z index — win."""
    occupy = np.zeros((size, size, 1)); norm = np.zeros((size, size, 3))
    p = np.asarray(points, dtype=np.float64); n = np.asarray(normals, dtype=np.float64)
    o0, o1, o2 = order
    tmp = max(p[:, o0].max() - p[:, o0].min(), p[:, o1].max() - p[:, o1].min())
    if tmp == 0:
        return occupy, norm                                            # dataset.py:150-153
    res = gripper_width / (size - margin)
    vox = np.stack([np.floor(p[:, o] / res + size / 2).astype(np.int64) for o in (o0, o1, o2)], 1)
    uniq, inv = np.unique(vox, axis=0, return_inverse=True)            # lexicographic, like coordinate_buffer
    inv = inv.reshape(-1)
    K = len(uniq)
    number = np.zeros(K, dtype=np.int64)
    acc = np.zeros((K, 3), dtype=np.float32)
    n32 = n.astype(np.float32)
    for i in range(len(p)):                                            # sequential float32 accumulation per voxel
        k = inv[i]
        if number[k] < vpn:
            acc[k] = acc[k] + n32[i]
            number[k] += 1
    mean = acc.astype(np.float64) / number[:, None].astype(np.float64)   # float32 / int64 -> float64 in numpy
    norm[uniq[:, 0], uniq[:, 1], :] = mean                             # duplicates: the last (largest z) wins
    occupy[uniq[:, 0], uniq[:, 1], 0] = number
    return occupy / occupy.max(), norm


def project_pc(points, normals, gripper_width, chann=3):
    """dataset.py:88-118 given the in-box points (hand frame) and their normals: rows with a NaN normal component
    are deleted first (:97-101)."""
    p = np.asarray(points, dtype=np.float64); n = np.asarray(normals, dtype=np.float64)
    bad = np.isnan(n).any(1)
    p, n = p[~bad], n[~bad]
    occ1, nrm1 = cal_projection(p, n, ORDERS[0], gripper_width)
    if chann == 3:
        return nrm1
    if chann != 12:
        raise NotImplementedError
    occ2, nrm2 = cal_projection(p, n, ORDERS[1], gripper_width)
    occ3, nrm3 = cal_projection(p, n, ORDERS[2], gripper_width)
    return np.dstack([occ1, nrm1, occ2, nrm2, occ3, nrm3])


def register_depth_map(depth, rgb_shape, depthK, rgbK, H):
    """ycb_cloud_generate.py:60-121, vectorised: every depth pixel is carried into the RGB camera and the largest
    depth per RGB pixel survives (a z-buffer that keeps the FARTHEST sample, as the reference's ``>`` does)."""
    hd, wd = depth.shape
    hr, wr = rgb_shape[:2]
    v, u = np.mgrid[0:hd, 0:wd]
    d = depth.astype(np.float64)
    inv_fx, inv_fy = 1.0 / depthK[0, 0], 1.0 / depthK[1, 1]
    x = ((u - depthK[0, 2]) * d) * inv_fx
    y = ((v - depthK[1, 2]) * d) * inv_fy
    z = d
    X = H[0, 0] * x + H[0, 1] * y + H[0, 2] * z + H[0, 3]
    Y = H[1, 0] * x + H[1, 1] * y + H[1, 2] * z + H[1, 3]
    Z = H[2, 0] * x + H[2, 1] * y + H[2, 2] * z + H[2, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / Z
        uu = (rgbK[0, 0] * X) * inv + rgbK[0, 2]
        vv = (rgbK[1, 1] * Y) * inv + rgbK[1, 2]
    ok = d != 0
    ur = np.trunc(np.where(ok, uu, 0) + 0.5).astype(np.int64)      # int(): truncation toward zero
    vr = np.trunc(np.where(ok, vv, 0) + 0.5).astype(np.int64)
    ok &= (ur >= 0) & (ur < wr) & (vr >= 0) & (vr < hr)
    out = np.zeros((hr, wr))
    np.maximum.at(out, (vr[ok], ur[ok]), Z[ok])                    # only depths > the running value (init 0) are written
    return out


def depth_map_to_cloud(depth, rgbK, refFromRGB, objFromref):
    """ycb_cloud_generate.py:124-184 with organized=False: (P,3) xyz of the pixels with depth > 0 in row-major
    order (the colour columns are a plain gather of the image and are not restated)."""
    v, u = np.nonzero(depth > 0)
    d = depth[v, u].astype(np.float64)
    x = (u - rgbK[0, 2]) * d * (1.0 / rgbK[0, 0])
    y = (v - rgbK[1, 2]) * d * (1.0 / rgbK[1, 1])
    z = d
    A = refFromRGB
    x1 = A[0, 0] * x + A[0, 1] * y + A[0, 2] * z + A[0, 3]
    y1 = A[1, 0] * x + A[1, 1] * y + A[1, 2] * z + A[1, 3]
    z1 = A[2, 0] * x + A[2, 1] * y + A[2, 2] * z + A[2, 3]
    O = objFromref
    return np.stack([O[0, 0] * x1 + O[0, 1] * y1 + O[0, 2] * z1 + O[0, 3],
                     O[1, 0] * x1 + O[1, 1] * y1 + O[1, 2] * z1 + O[1, 3],
                     O[2, 0] * x1 + O[2, 1] * y1 + O[2, 2] * z1 + O[2, 3]], 1)


def depth_map_to_cloud_organized(depth, rgb, rgbK, refFromRGB, objFromref):
    """ycb_cloud_generate.py:124-184 with organized=True: (h,w,6) — xyz + colour where depth > 0, NaN xyz and zero colour
    where depth <= 0 (:147-155)."""
    h, w = depth.shape
    out = np.zeros((h, w, 6))
    out[:, :, :3] = np.nan
    good = depth > 0
    out[good, :3] = depth_map_to_cloud(depth, rgbK, refFromRGB, objFromref)
    out[good, 3:] = np.asarray(rgb)[good]
    return out


def gpd_forward_torch(sd, x):
    """GPDClassifier.forward (gpd.py:21-31), eval mode (dropout off), on the reference's ATen ops."""
    import torch.nn.functional as F
    x = F.max_pool2d(F.conv2d(x, sd["conv1.weight"], sd["conv1.bias"]), 2, stride=2)
    x = F.max_pool2d(F.conv2d(x, sd["conv2.weight"], sd["conv2.bias"]), 2, stride=2)
    x = x.view(-1, 7200)
    x = F.relu(F.linear(x, sd["fc1.weight"], sd["fc1.bias"]))
    return F.log_softmax(F.linear(x, sd["fc2.weight"], sd["fc2.bias"]), dim=-1)
