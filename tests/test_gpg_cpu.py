"""CPU: the GPG-sampler oracle against the goldens recorded from the EXECUTED reference
(oracle/make_golden_gpg.py), and the host half of the product sampler (hand model, rotation sweep)."""
import glob
import os

import numpy as np
import pytest

from oracle import gpg_oracle as go
from tests.helpers import GOLDEN

CASES = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN, "gpg_*.npz")))


def load_case(tag):
    fx = np.load(os.path.join(GOLDEN, f"gpg_{tag}.npz"))
    pts, nrm = go.synth_scene(str(fx["kind"]), int(fx["P"]), int(fx["seed_scene"]))
    assert pts.sum() == fx["pts_sum"] and nrm.sum() == fx["nrm_sum"]       # the scene generator is reproducible
    return fx, pts, pts[pts[:, 2] > 0.010], nrm


def test_golden_inventory():
    assert set(CASES) >= {"cyl", "box", "ell", "cyl_stop", "box_big"}


@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_executed_reference(tag):
    fx, pts, pfs, nrm = load_case(tag)
    got = go.sample_grasps(pts, pfs, nrm, fx["draws"], int(fx["num_grasps"]), int(fx["max_num_samples"]))
    got = np.array(got).reshape(-1, 5, 3)
    assert got.shape == fx["grasps"].shape and got.shape[0] > 0
    np.testing.assert_allclose(got, fx["grasps"], rtol=0, atol=1e-13)


def test_host_half_matches_oracle():
    from pointnetgpd_amd import gpg
    g = gpg._gripper_dict(gpg.ROBOTIQ_85)
    assert g == {k: go.ROBOTIQ_85[k] for k in g}
    rng = np.random.default_rng(0)
    hp = go.hand_points(go.ROBOTIQ_85, np.zeros(3), np.array([1.0, 0, 0]), np.array([0, 1.0, 0]))
    boxes = gpg.hand_boxes(g)
    for i, w in enumerate(go.WAYS):                       # product order: open, left, right, bottom
        assert np.array_equal(boxes[i], go.way_box(hp, w))
    for _ in range(20):
        a = rng.normal(size=3); a /= np.linalg.norm(a)
        b = np.cross(a, rng.normal(size=3)); b /= np.linalg.norm(b)
        c = rng.normal(size=3) * 0.1
        np.testing.assert_allclose(gpg.hand_corners(g, c, a, b), go.hand_points(go.ROBOTIQ_85, c, a, b)[1:], atol=1e-16)
    # the constants block of the device kernels is what numpy computes for the reference's expressions
    prm, R, D, S = gpg.GpgGraspSamplerPcl._params(g)
    assert (R, D, S) == (19, 21, 25)
    np.testing.assert_array_equal(prm[16:16 + R], np.arange(-90, 91, 10).astype(np.float64) / 180 * np.pi)
    fw = g["finger_width"]
    np.testing.assert_array_equal(prm[48:48 + D], np.arange(-10 * fw, 11 * fw, fw))
    assert prm[1] == 0.125 and prm[2] == 0.0625 and prm[7] == -((g["hand_outer_diameter"] - fw * 2) * 0.5)
    # the quirk the device enumeration reproduces: dtheta = 0 is a half turn about the minor axis, not the identity
    minor = rng.normal(size=3); minor /= np.linalg.norm(minor)
    ref = go.rotation_from_quaternion(np.array([0.0, *minor]))
    np.testing.assert_allclose(ref @ minor, minor, atol=1e-15)
    assert abs(np.trace(ref) + 1.0) < 1e-14


def test_sampler_needs_the_gpu():
    from pointnetgpd_amd import gpg
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pts, nrm = go.synth_scene("box", 200, 1)
    with pytest.raises((RuntimeError, AssertionError)):
        gpg.GpgGraspSamplerPcl().sample_grasps(pts, pts, nrm, 5, 5)


def _crop_infer_cases():
    fx = np.load(os.path.join(GOLDEN, "crop_infer.npz"))
    for c in range(int(fx["n_cases"])):
        pts, _ = go.synth_scene(str(fx[f"kind_{c}"]), int(fx[f"P_{c}"]), int(fx[f"seed_{c}"]))
        counts = fx[f"counts_{c}"]
        ind = np.split(fx[f"ind_{c}"].astype(np.int64), np.cumsum(counts)[:-1])
        yield pts.astype(np.float32), fx[f"grasps_{c}"], counts, ind, fx[f"pts_head_{c}"], fx[f"pts_sum_{c}"]


def test_infer_crop_oracle_matches_executed_reference():
    """kinect2grasp.py:178-258, cut out of the reference file and executed by oracle/make_golden_gpg.py — the pin
    of the inference-style crop (oracle and host mirror)."""
    from oracle import crop_oracle as co
    from pointnetgpd_amd import crop
    n = 0
    for pc, grasps, counts, ind, head, psum in _crop_infer_cases():
        ind_o, pts_o = co.collect_pc_infer(grasps, pc)
        frames = crop.frames_from_grasps_infer(grasps)
        for g in range(len(grasps)):
            assert np.array_equal(ind_o[g], ind[g])
            k = min(3, counts[g])
            np.testing.assert_allclose(pts_o[g][:k], head[g][:k], rtol=0, atol=1e-15)
            np.testing.assert_allclose(pts_o[g].sum(0), psum[g], rtol=0, atol=1e-12)
            ih, ph = crop.collect_pc_numpy(frames[g], pc)            # the product's host-side mirror
            assert np.array_equal(ih, ind[g])
            n += 1
    assert n == 102


def test_sampler_alias_for_kinect2grasp_import():
    """``from dexnet.grasping import GpgGraspSamplerPcl`` (kinect2grasp.py:33) resolves to the GPU sampler."""
    import importlib
    import sys
    import pointnetgpd_amd
    from pointnetgpd_amd import gpg
    saved = {k: sys.modules.get(k) for k in ("dexnet", "dexnet.grasping")}
    try:
        cls = pointnetgpd_amd.install_sampler_alias()
        assert cls is gpg.GpgGraspSamplerPcl
        assert importlib.import_module("dexnet.grasping").GpgGraspSamplerPcl is gpg.GpgGraspSamplerPcl
        ags = cls(go.ROBOTIQ_85, {"anything": 1})                       # (gripper, yaml_config) as at kinect2grasp.py:52
        assert gpg._gripper_dict(ags.gripper)["hand_depth"] == 0.125
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
