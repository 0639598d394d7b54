"""CPU: the GPD-row oracle (oracle/gpd_oracle.py) against the records of the EXECUTED reference
(tests/golden/gpd_*.npz, made by oracle/make_golden_gpd.py), and the host mirrors built on it."""
import os

import numpy as np
import torch

from oracle import gpd_oracle as go
from tests import synth_gpd
from tests.conftest import GOLDEN


def _proj():
    return np.load(os.path.join(GOLDEN, "gpd_projection.npz"))


def test_projection_oracle_matches_reference_record():
    fx = _proj()
    for tag in [str(t) for t in fx["tags"]]:
        pc, nrm, ind, w = fx[f"{tag}/pc"], fx[f"{tag}/normals"], fx[f"{tag}/in_ind"], float(fx[f"{tag}/width"])
        for chann in (3, 12):
            got = go.project_pc(pc[ind], nrm[ind], w, chann)
            ref = fx[f"{tag}/out{chann}"]
            assert got.shape == ref.shape == (60, 60, chann)
            np.testing.assert_array_equal(got, ref, err_msg=f"{tag} chann {chann}")   # bit-exact, floats included


def test_cloudgen_oracle_matches_reference_record():
    fx = np.load(os.path.join(GOLDEN, "gpd_cloudgen.npz"))
    for tag in [str(t) for t in fx["tags"]]:
        sc = synth_gpd.cloudgen_scene(tag)
        reg = go.register_depth_map(sc["depth"], sc["rgb"].shape, sc["depthK"], sc["rgbK"], sc["H"])
        assert int((reg > 0).sum()) == int(fx[f"{tag}/reg_nonzero"])
        np.testing.assert_array_equal(reg.reshape(-1)[fx[f"{tag}/reg_pix"]], fx[f"{tag}/reg_val"])
        assert abs(reg.sum() - float(fx[f"{tag}/reg_sum"])) <= 1e-9 * abs(float(fx[f"{tag}/reg_sum"]))
        regm = reg.copy(); regm[sc["mask"]] = 0
        cloud = go.depth_map_to_cloud(regm, sc["rgbK"], sc["refFromRGB"], sc["objFromref"])
        assert len(cloud) == int(fx[f"{tag}/cloud_len"])
        np.testing.assert_array_equal(cloud[fx[f"{tag}/cloud_rows"]], fx[f"{tag}/cloud_val"][:, :3])
        np.testing.assert_allclose(cloud.sum(0), fx[f"{tag}/cloud_colsum"][:3], rtol=1e-12)
        if tag == "small":
            np.testing.assert_array_equal(reg, fx["small/registered"])
            np.testing.assert_array_equal(cloud, fx["small/cloud"][:, :3])
            org = go.depth_map_to_cloud_organized(regm, sc["rgb"], sc["rgbK"], sc["refFromRGB"], sc["objFromref"])
            ref = fx["small/cloud_organized"]            # executed reference, organized=True (ycb_cloud_generate.py:130-155)
            assert org.shape == ref.shape == regm.shape + (6,)
            assert np.isnan(ref[regm <= 0, :3]).all() and (ref[regm <= 0, 3:] == 0).all()
            np.testing.assert_array_equal(org, ref)      # NaNs compare equal position-wise


def test_classifier_oracle_and_mirror_match_reference_record():
    from pointnetgpd_amd.model.gpd import GPDClassifier
    fx = np.load(os.path.join(GOLDEN, "gpd_classifier.npz"))
    for chann in (3, 12):
        torch.manual_seed(100 + chann)
        m = GPDClassifier(chann).eval()                     # same construction order -> same seeded weights
        sd = m.state_dict()
        names = [str(n) for n in fx[f"names{chann}"]]
        assert sorted(sd) == names
        cs = np.array([[sd[k].double().sum().item(), sd[k].double().abs().sum().item()] for k in names])
        np.testing.assert_array_equal(cs, fx[f"checksums{chann}"])
        x = torch.from_numpy(fx[f"x{chann}"])
        with torch.no_grad():
            np.testing.assert_allclose(go.gpd_forward_torch(sd, x).numpy(), fx[f"logp{chann}"], atol=1e-6)
            np.testing.assert_allclose(m(x).numpy(), fx[f"logp{chann}"], atol=1e-6)     # CPU composite of the mirror


def test_dataset_projection_with_supplied_normals():
    """``projection=True`` on the Dataset mirror: collect_pc returns the projection images when a normal estimator
    is supplied (open3d's is not reproducible here — SURVEY.md §8c), and raises a clear error otherwise."""
    import pytest
    from pointnetgpd_amd.model import dataset as mirror
    fx = _proj()
    tag = "dense"
    pc, nrm, ind, w = fx[f"{tag}/pc"], fx[f"{tag}/normals"], fx[f"{tag}/in_ind"], float(fx[f"{tag}/width"])
    ds = mirror.BaseGraspDataset.__new__(mirror.BaseGraspDataset)
    ds.project_chann, ds.project_size, ds.voxel_point_num, ds.projection_margin = 12, 60, 50, 1
    ds.in_ind = ind
    ds.normal_estimator = lambda points: nrm
    np.testing.assert_array_equal(ds.project_pc(pc, w), fx[f"{tag}/out12"])
    ds.normal_estimator = None
    with pytest.raises(RuntimeError, match="open3d"):
        ds.project_pc(pc, w)
