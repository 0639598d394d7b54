"""Deterministic synthetic inputs for the depth-registration goldens (test infrastructure): used by
oracle/make_golden_gpd.py (to feed the executed reference) and by the tests (to feed the oracle restatement and the
HIP kernels with the very same arrays without storing megabytes of inputs in the repo)."""
import numpy as np

SCENES = {"small": ((60, 80), (90, 120), 41), "vga": ((480, 640), (480, 640), 43)}


def cloudgen_scene(tag):
    (hd, wd), (hr, wr), seed = SCENES[tag]
    rng = np.random.default_rng(seed)
    fx = 0.9 * wd
    depthK = np.array([[fx, 0, wd / 2 - 0.37], [0, fx * 1.01, hd / 2 + 0.21], [0, 0, 1.0]])
    fr = 0.95 * wr
    rgbK = np.array([[fr, 0, wr / 2 + 0.4], [0, fr * 0.99, hr / 2 - 0.3], [0, 0, 1.0]])
    ang = 0.03
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
    H = np.eye(4); H[:3, :3] = R; H[:3, 3] = [0.025, -0.004, 0.002]
    yy, xx = np.mgrid[0:hd, 0:wd]
    depth = 0.7 + 0.1 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + rng.normal(size=(hd, wd)) * 1e-3
    depth[rng.random((hd, wd)) < 0.15] = 0.0                 # holes (filterDiscontinuities output)
    depth[: hd // 8] = 0.0
    rgb = rng.integers(0, 256, size=(hr, wr, 3), dtype=np.uint8)
    mask = rng.random((hr, wr)) < 0.3                        # generate(): registeredDepthMap[pbmImage == 255] = 0
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    a, b, c, d = q
    Rr = np.array([[1 - 2 * (c * c + d * d), 2 * (b * c - a * d), 2 * (b * d + a * c)],
                   [2 * (b * c + a * d), 1 - 2 * (b * b + d * d), 2 * (c * d - a * b)],
                   [2 * (b * d - a * c), 2 * (c * d + a * b), 1 - 2 * (b * b + c * c)]])
    refFromRGB = np.eye(4); refFromRGB[:3, :3] = Rr; refFromRGB[:3, 3] = [0.1, -0.2, 0.05]
    objFromref = np.eye(4); objFromref[:3, :3] = Rr.T @ R; objFromref[:3, 3] = [-0.3, 0.02, 0.6]
    return dict(depth=depth, rgb=rgb, depthK=depthK, rgbK=rgbK, H=H, mask=mask, refFromRGB=refFromRGB,
                objFromref=objFromref)


def sample_rows(n, count=4096, seed=7):
    """Fixed pseudo-random row subset used to spot-check large outputs."""
    if n <= count:
        return np.arange(n)
    return np.sort(np.random.default_rng(seed).choice(n, count, replace=False))
