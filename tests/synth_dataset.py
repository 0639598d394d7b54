"""Deterministic miniature of the reference's on-disk training data layout (test infrastructure).

    <root>/PointNetGPD/data/google2cloud.pkl                       {obj: (cloud_obj_name, 4x4 T)}
    <root>/PointNetGPD/data/ycb_grasp/<tag>/<obj>.npy              (G,12) f64 grasp rows
    <root>/data/ycb-tools/models/ycb/<obj>/rgbd/clouds/pc_NP{1,3}_NP5_<deg>.npy   (P,3) f64 clouds

Grasp row = [center3, axis3, width, angle, jaw_width, min_width, fc_level, canny]
(dex-net/apps/generate-dataset-canny.py:48-54).  Used both by oracle/make_golden.py (to record what
the reference's Dataset classes return) and by tests (to check the mirror against that record).
"""
import os
import pickle

import numpy as np

OBJECTS = ["003_cracker_box", "011_banana", "025_mug"]


def build(root, grasps_per_obj=12, points=3000, seed=5):
    rng = np.random.default_rng(seed)
    os.makedirs(f"{root}/PointNetGPD/data", exist_ok=True)
    transforms = {}
    for oi, obj in enumerate(OBJECTS):
        # a rigid transform mesh frame -> cloud frame
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = rng.uniform(-0.05, 0.05, size=3)
        transforms[obj] = (obj, T)
        cdir = f"{root}/data/ycb-tools/models/ycb/{obj}/rgbd/clouds"
        os.makedirs(cdir, exist_ok=True)
        for cam in ("NP1", "NP3"):
            for deg in (0, 120, 240):
                pc = rng.uniform(-0.07, 0.07, size=(points, 3)) + T[:3, 3]
                np.save(f"{cdir}/pc_{cam}_NP5_{deg}.npy", pc)
        for tag in ("train", "test"):
            gdir = f"{root}/PointNetGPD/data/ycb_grasp/{tag}"
            os.makedirs(gdir, exist_ok=True)
            g = np.zeros((grasps_per_obj, 12))
            g[:, 0:3] = rng.uniform(-0.03, 0.03, size=(grasps_per_obj, 3))
            g[:, 3:6] = rng.normal(size=(grasps_per_obj, 3))
            g[:, 6] = 0.085
            g[:, 7] = rng.uniform(-np.pi, np.pi, size=grasps_per_obj)
            g[:, 8] = 0.085
            g[:, 10] = rng.choice([0.4, 0.45, 0.5, 0.8, 1.2, 1.6, 2.0], size=grasps_per_obj)
            g[:, 11] = rng.uniform(0, 1, size=grasps_per_obj)
            np.save(f"{gdir}/{obj}.npy", g)
    with open(f"{root}/PointNetGPD/data/google2cloud.pkl", "wb") as f:
        pickle.dump(transforms, f)
    return root


CASES = [   # (class name, ctor kwargs) recorded by make_golden.py and replayed by the tests
    ("PointGraspOneViewDataset", dict(grasp_points_num=64, grasp_amount_per_file=12, thresh_good=0.6,
                                      thresh_bad=0.6, tag="train")),
    ("PointGraspOneViewMultiClassDataset", dict(grasp_points_num=200, grasp_amount_per_file=12, thresh_good=0.5,
                                                thresh_bad=1.2, tag="test", with_obj=True)),
    ("PointGraspDataset", dict(obj_points_num=4000, grasp_points_num=100, pc_file_used_num=3,
                               grasp_amount_per_file=12, thresh_good=0.6, thresh_bad=0.6, tag="train",
                               with_obj=True)),
    ("PointGraspMultiClassDataset", dict(obj_points_num=4000, grasp_points_num=100, pc_file_used_num=3,
                                         grasp_amount_per_file=12, thresh_good=0.5, thresh_bad=1.2, tag="test")),
    # items the reference returns as None: (a) a score between the two thresholds of the 2-class rule
    # (dataset.py:448-453), (b) fewer than min_point_limit in-box points (dataset.py:71-72; the limit is an instance
    # attribute set in __init__, raised here after construction so that the 3000-point views fall below it)
    ("PointGraspOneViewDataset", dict(grasp_points_num=64, grasp_amount_per_file=12, thresh_good=0.5,
                                      thresh_bad=1.2, tag="train")),
    ("PointGraspOneViewDataset", dict(grasp_points_num=64, grasp_amount_per_file=12, thresh_good=0.6,
                                      thresh_bad=0.6, tag="test"), dict(min_point_limit=175)),
]


def replay(module, root, np_seed=77, indices=range(0, 36, 5)):
    """Instantiate every CASE from ``module`` (the reference's or the mirror's ``dataset``) and return
    [(case, index, item)] under a fixed numpy seed.  Items are looked up BY OBJECT NAME so the result does
    not depend on set-iteration order of ``self.object`` (hash-randomised between processes)."""
    os.environ["PointNetGPD_FOLDER"] = root
    out = []
    for case in CASES:
        name, kw = case[0], case[1]
        ds = getattr(module, name)(**kw)
        for attr, val in (case[2] if len(case) > 2 else {}).items():
            setattr(ds, attr, val)
        assert len(ds) == len(OBJECTS) * kw["grasp_amount_per_file"], (name, len(ds))
        ds.object = sorted(ds.object)
        for key in ds.d_pc:               # glob order is filesystem-dependent (full-view classes do not sort)
            ds.d_pc[key] = sorted(ds.d_pc[key])
        for i in indices:
            np.random.seed(np_seed + i)
            item = ds[i]
            out.append((name, i, item))
    return out
