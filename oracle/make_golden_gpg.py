"""Generate tests/golden/gpg_*.npz by EXECUTING the unmodified reference sampler
(/root/reference/dex-net/src/dexnet/grasping/grasp_sampler.py :: GpgGraspSamplerPcl.sample_grasps) in the
build container.  TEST INFRASTRUCTURE ONLY; runs where /root/reference exists, never on the GPU box.

The file is loaded by path under stand-ins for the imports it cannot resolve here:
  dexnet.grasping (names only, unused by the Pcl sampler), autolab_core.RigidTransform.rotation_from_quaternion
  (restated from autolab_core's transformations.quaternion_matrix), open3d (brute-force radius/kNN search that
  returns the 2-D [0, i] arrays the reference indexes), scipy.random.seed (removed from SciPy; a no-op here so
  that the np.random.seed below fixes the sample-point draws), rospy/mayavi absent -> the module's own fallbacks.

Usage:  python oracle/make_golden_gpg.py
"""
import contextlib
import importlib.util
import io
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gpg_oracle as go  # noqa: E402

REF_FILE = "/root/reference/dex-net/src/dexnet/grasping/grasp_sampler.py"
OUT = os.path.join(ROOT, "tests", "golden")


def _install_stubs():
    grasping = types.ModuleType("dexnet.grasping")
    for n in ("Grasp", "Contact3D", "ParallelJawPtGrasp3D", "PointGraspMetrics3D", "GraspableObject3D"):
        setattr(grasping, n, type(n, (), {}))
    dexnet = types.ModuleType("dexnet")
    dexnet.grasping = grasping
    sys.modules["dexnet"] = dexnet
    sys.modules["dexnet.grasping"] = grasping

    autolab = types.ModuleType("autolab_core")

    class RigidTransform:
        @staticmethod
        def rotation_from_quaternion(q_wxyz):
            return go.rotation_from_quaternion(q_wxyz)
    autolab.RigidTransform = RigidTransform
    sys.modules["autolab_core"] = autolab

    o3d = types.ModuleType("open3d")
    o3d.geometry = types.SimpleNamespace()
    o3d.utility = types.SimpleNamespace()

    class PointCloud:
        points = None
    o3d.geometry.PointCloud = PointCloud
    o3d.utility.Vector3dVector = lambda a: np.asarray(a, dtype=np.float64)

    class KDTreeFlann:
        def __init__(self, pcd):
            self.pts = np.asarray(pcd.points, dtype=np.float64)

        def search_hybrid_vector_3d(self, query, radius, max_nn):
            idx, d2 = go.neighbours(self.pts, np.asarray(query, dtype=np.float64), radius, max_nn)
            return len(idx), idx.reshape(1, -1), d2.reshape(1, -1)
    o3d.geometry.KDTreeFlann = KDTreeFlann
    sys.modules["open3d"] = o3d

    import scipy
    scipy.random = types.SimpleNamespace(seed=lambda *a, **k: None)


def load_reference():
    _install_stubs()
    spec = importlib.util.spec_from_file_location("_ref_grasp_sampler", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class Cloud(np.ndarray):
    """The reference calls ``point_cloud.to_array()`` (pcl era) AND hands the same object to numpy/open3d."""
    def to_array(self):
        return np.asarray(self)


def run_reference(mod, points, points_for_sample, normals, seed, num_grasps, max_num_samples):
    sampler = object.__new__(mod.GpgGraspSamplerPcl)          # skip the YAML-driven _configure(); unused here
    sampler.gripper = types.SimpleNamespace(**go.ROBOTIQ_85)
    draws = []
    real_choice = np.random.choice

    def recording_choice(a, size=None, replace=True, p=None):
        r = real_choice(a, size=size, replace=replace, p=p)
        draws.append(int(np.asarray(r).reshape(-1)[0]))
        return r
    np.random.seed(seed)
    np.random.choice = recording_choice
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            res = sampler.sample_grasps(points.view(Cloud), points_for_sample, normals.copy(),
                                        num_grasps=num_grasps, max_num_samples=max_num_samples)
    finally:
        np.random.choice = real_choice
    arr = np.array([[np.asarray(v, dtype=np.float64).reshape(3) for v in g] for g in res]).reshape(-1, 5, 3)
    return arr, np.array(draws, dtype=np.int64)


CASES = [  # tag, kind, P, seed_scene, seed_draws, num_grasps, max_num_samples
    ("cyl", "cylinder", 1500, 11, 101, 1000, 30),
    ("box", "box", 2000, 12, 102, 1000, 30),
    ("ell", "ellipsoid", 1200, 13, 103, 1000, 30),
    ("cyl_stop", "cylinder", 1500, 11, 104, 6, 40),        # early stop on num_grasps
    ("box_big", "box", 6000, 14, 105, 1000, 16),
]


KINECT_FILE = "/root/reference/dex-net/apps/kinect2grasp.py"


def load_kinect2grasp_crop(mod):
    """kinect2grasp.py cannot be imported (rospy, pcl, argparse + node start-up at module level), but its two crop
    functions (check_collision_square :178-235, collect_pc :238-258) only touch numpy and the module-global sampler
    ``ags``.  Their UNMODIFIED source segments are cut out of the reference file with ``ast`` at run time and
    executed in a namespace holding numpy and a reference sampler object — nothing is copied into the repo."""
    import ast
    src = open(KINECT_FILE).read()
    tree = ast.parse(src)
    want = {"check_collision_square", "collect_pc"}
    segs = [ast.get_source_segment(src, n) for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert len(segs) == 2
    ags = object.__new__(mod.GpgGraspSamplerPcl)
    ags.gripper = types.SimpleNamespace(**go.ROBOTIQ_85)
    ns = {"np": np, "ags": ags}
    for seg in segs:
        exec(compile(seg, KINECT_FILE, "exec"), ns)
    return ns["collect_pc"]


def infer_crop_cases(mod):
    """Inference-style crop record: scenes x grasp rows (the sampler's own output + random frames) ->
    per-grasp in-box index sets and hand-frame points, from the executed reference functions."""
    collect_pc = load_kinect2grasp_crop(mod)
    rec = {}
    rng = np.random.default_rng(99)
    for c, (tag, kind, P, ss) in enumerate([("cyl", "cylinder", 1500, 11), ("box", "box", 2000, 12),
                                            ("ell", "ellipsoid", 1200, 13)]):
        pts, _ = go.synth_scene(kind, P, ss)
        pts32 = pts.astype(np.float32)                                         # kinect2grasp.py:112
        sampled = np.load(os.path.join(OUT, f"gpg_{tag}.npz"))["grasps"]
        G2 = 12
        a = rng.normal(size=(G2, 3)); a /= np.linalg.norm(a, axis=1, keepdims=True)
        b = np.cross(a, rng.normal(size=(G2, 3))); b /= np.linalg.norm(b, axis=1, keepdims=True)
        b *= rng.uniform(0.5, 2.0, (G2, 1))                                    # un-normalised axes are normalised (:180-185)
        m = np.cross(a, b)
        bottom = pts[rng.integers(0, P, G2)] - 0.05 * a
        rand = np.stack([bottom, a, b, m, bottom], 1)
        grasps = np.concatenate([sampled, rand], 0)
        ind, inpts = collect_pc([list(g) for g in grasps], pts32)
        rec[f"kind_{c}"] = kind; rec[f"P_{c}"] = P; rec[f"seed_{c}"] = ss
        rec[f"grasps_{c}"] = grasps
        rec[f"counts_{c}"] = np.array([len(i) for i in ind])
        rec[f"ind_{c}"] = np.concatenate(ind).astype(np.int16)              # P < 32768
        # hand-frame coordinates: first 3 rows per grasp + per-grasp column sums (the full record would be 2 MB)
        rec[f"pts_head_{c}"] = np.stack([np.concatenate([p_[:3], np.zeros((3 - min(3, len(p_)), 3))]) for p_ in inpts])
        rec[f"pts_sum_{c}"] = np.stack([p_.sum(0) for p_ in inpts])
        print(f"crop_infer {tag}: {len(grasps)} grasps, counts {rec[f'counts_{c}'].tolist()}")
    rec["n_cases"] = 3
    np.savez_compressed(os.path.join(OUT, "crop_infer.npz"), **rec)


def main():
    mod = load_reference()
    os.makedirs(OUT, exist_ok=True)
    for tag, kind, P, ss, sd, ng, mx in CASES:
        pts, nrm = go.synth_scene(kind, P, ss)
        pfs = pts[pts[:, 2] > 0.010]                          # kinect2grasp.py:141
        grasps, draws = run_reference(mod, pts, pfs, nrm, sd, ng, mx)
        np.savez_compressed(os.path.join(OUT, f"gpg_{tag}.npz"), kind=kind, P=P, seed_scene=ss, draws=draws,
                            num_grasps=ng, max_num_samples=mx, grasps=grasps,
                            pts_sum=pts.sum(), nrm_sum=nrm.sum())
        print(f"{tag}: {len(draws)} draws -> {len(grasps)} grasps")
    infer_crop_cases(mod)


if __name__ == "__main__":
    main()
