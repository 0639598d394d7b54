"""CPU: crop oracle + host-side crop / Dataset mirror vs the golden records of the reference."""
import os
import tempfile

import numpy as np
import pytest

from oracle import crop_oracle as co
from tests import synth_dataset
from tests.helpers import GOLDEN


@pytest.fixture(scope="module")
def crop_fx():
    return np.load(os.path.join(GOLDEN, "crop_train.npz"))


def test_crop_oracle_matches_reference_collect_pc(crop_fx):
    """oracle.collect_pc_train == reference BaseGraspDataset.collect_pc (recorded): index sets exactly,
    kept points to 1e-12, None decisions identical."""
    fx = crop_fx
    for c in range(len(fx["grasps"])):
        pts, ind, _ = co.collect_pc_train(fx["grasps"][c], fx["pcs"][c], fx["Ts"][c])
        np.testing.assert_array_equal(ind, fx[f"ind_{c}"])
        assert (pts is None) == bool(fx["is_none"][c])
        assert len(ind) == int(fx["counts"][c])
        if pts is not None:
            np.testing.assert_allclose(pts, fx[f"pts_{c}"], rtol=0, atol=1e-12)


def test_product_frames_and_host_crop(crop_fx):
    """pointnetgpd_amd.crop (vectorised frames + numpy crop) vs the oracle and the golden record."""
    from pointnetgpd_amd import crop
    fx = crop_fx
    for c in range(len(fx["grasps"])):
        frame = crop.frames_from_grasps_train(fx["grasps"][c][None], fx["Ts"][c])[0]
        center, M, width = co.grasp_frame_train(fx["grasps"][c], fx["Ts"][c])
        np.testing.assert_allclose(frame[0:3], center, rtol=0, atol=1e-15)
        np.testing.assert_allclose(frame[3:12].reshape(3, 3), M, rtol=0, atol=1e-15)
        np.testing.assert_allclose(frame[15:18], [width / 4, width / 2, width / 4], rtol=0, atol=0)
        ind, pts = crop.collect_pc_numpy(frame, fx["pcs"][c])
        np.testing.assert_array_equal(ind, fx[f"ind_{c}"])
        if not fx["is_none"][c]:
            np.testing.assert_allclose(pts, fx[f"pts_{c}"], rtol=0, atol=1e-12)
    # batched frames == per-grasp frames
    T = fx["Ts"][0]
    all_frames = crop.frames_from_grasps_train(fx["grasps"], T)
    for c in range(len(fx["grasps"])):
        np.testing.assert_allclose(all_frames[c], crop.frames_from_grasps_train(fx["grasps"][c][None], T)[0],
                                   rtol=0, atol=1e-15)   # BLAS picks another kernel for the batched product


def test_infer_frames_and_box():
    """Inference-style crop: product frames vs the oracle restatement of kinect2grasp.py:178-258."""
    from pointnetgpd_amd import crop
    rng = np.random.default_rng(3)
    G, P = 7, 5000
    q = rng.normal(size=(G, 3, 3))
    grasps = np.zeros((G, 5, 3))
    for g in range(G):
        a = q[g, 0] / np.linalg.norm(q[g, 0])
        b = np.cross(a, q[g, 1]); b /= np.linalg.norm(b)
        grasps[g] = [rng.uniform(-0.05, 0.05, 3), a * 1.7, b * 0.4, np.cross(a, b) * 2.0, np.zeros(3)]
    pc = rng.uniform(-0.15, 0.15, size=(P, 3)).astype(np.float32)
    ind_ref, pts_ref = co.collect_pc_infer(grasps, pc)
    frames = crop.frames_from_grasps_infer(grasps)
    lo, hi = co.infer_box()
    for g in range(G):
        np.testing.assert_array_equal(frames[g, 12:15], lo); np.testing.assert_array_equal(frames[g, 15:18], hi)
        ind, pts = crop.collect_pc_numpy(frames[g], pc)
        np.testing.assert_array_equal(ind, ind_ref[g])
        np.testing.assert_allclose(pts, pts_ref[g], rtol=0, atol=1e-12)
    w = 0.218 - 2 * 0.0255
    assert abs(hi[1] - w / 2) < 1e-15 and hi[0] == 0.125 and lo[0] == 0.0


def test_labels_and_resample_rule():
    assert co.label_2class(0.5, 0.0, 0.6, 0.6) == 1 and co.label_2class(0.6, 0.0, 0.6, 0.6) == 0
    assert co.label_2class(0.55, 0.0, 0.5, 0.6) is None
    assert co.label_3class(1.2, 0.0, 0.5, 1.2) == 0 and co.label_3class(0.5, 0.0, 0.5, 1.2) == 2
    assert co.label_3class(0.8, 0.5, 0.5, 1.2) == 1
    assert co.resample_rule(10, 10) == dict(replace_train=True, replace_infer=False)
    assert co.resample_rule(11, 10) == dict(replace_train=False, replace_infer=False)
    assert co.resample_rule(9, 10) == dict(replace_train=True, replace_infer=True)


def test_dataset_mirror_matches_reference_record():
    """The four Dataset classes on the synthetic miniature tree return exactly what the reference's
    classes returned (tests/golden/dataset_items.npz, recorded by oracle/make_golden.py)."""
    from pointnetgpd_amd.model import dataset as mirror
    fx = np.load(os.path.join(GOLDEN, "dataset_items.npz"))
    with tempfile.TemporaryDirectory() as root:
        synth_dataset.build(root)
        items = synth_dataset.replay(mirror, root)
    assert [it[0] for it in items] == [str(s) for s in fx["names"]]
    assert [it[1] for it in items] == [int(i) for i in fx["indices"]]
    for n, (name, i, item) in enumerate(items):
        assert (item is None) == bool(fx["none_mask"][n])
        if item is None:
            continue
        pc, label = item[0], item[1]
        ref = fx[f"pc_{n}"]
        assert pc.shape == ref.shape and pc.dtype == np.float64
        np.testing.assert_allclose(pc, ref, rtol=0, atol=1e-12, err_msg=f"{name}[{i}]")
        assert int(label) == int(fx[f"label_{n}"])
        if len(item) > 2:
            assert str(item[2]) == str(fx[f"obj_{n}"])


def test_dataset_projection_items():
    """``projection=True`` (the GPD baseline's input, dataset.py:73-74,262-270): an item is a (chann,60,60) image stack
    built from the in-box points; without an estimator (open3d is absent here) the error says what to do."""
    from pointnetgpd_amd.model import dataset as mirror
    with tempfile.TemporaryDirectory() as root:
        synth_dataset.build(root)
        os.environ["PointNetGPD_FOLDER"] = root
        ds = mirror.PointGraspOneViewDataset(grasp_points_num=64, grasp_amount_per_file=12, thresh_good=0.6,
                                             thresh_bad=0.6, tag="train", projection=True, project_chann=12)
        with pytest.raises(RuntimeError, match="open3d"):
            ds[0]
        rng = np.random.default_rng(0)

        def fake_normals(points):
            n = rng.normal(size=points.shape)
            return n / np.linalg.norm(n, axis=1, keepdims=True)
        ds.normal_estimator = fake_normals
        np.random.seed(3)
        item = ds[0]
        assert item[0].shape == (12, 60, 60) and item[0].dtype == np.float64 and item[1] in (0, 1)
        occ = item[0][0]
        assert occ.max() == 1.0 and (occ >= 0).all()                       # occupancy normalised by its maximum


def test_device_loader_host_half(tmp_path, monkeypatch):
    """CPU: the host half of device_loader.DeviceGraspLoader — arena layout, the per-DATASET frame / label tables
    (computed once, indexed by item on the device) and an epoch's tables (permutation, view picks, arena spans; rank
    shares under one process per GPU) — against a per-item evaluation with the Dataset mirror's own pieces.  (The
    device half, one foreign call per batch, is covered by tests/test_gpu_device_loader.py.)"""
    from pointnetgpd_amd import crop
    from pointnetgpd_amd.device_loader import DeviceGraspLoader
    from pointnetgpd_amd.model import dataset as ds_mod
    root = synth_dataset.build(str(tmp_path / "tree"))
    monkeypatch.setenv("PointNetGPD_FOLDER", root)
    for ds in (ds_mod.PointGraspOneViewDataset(grasp_points_num=64, grasp_amount_per_file=12, thresh_good=0.45,
                                               thresh_bad=1.2, tag="train"),
               ds_mod.PointGraspMultiClassDataset(obj_points_num=4000, grasp_points_num=100, pc_file_used_num=3,
                                                  grasp_amount_per_file=12, thresh_good=0.5, thresh_bad=1.2, tag="test")):
        ld = object.__new__(DeviceGraspLoader)               # host tables only: no device, no arena upload
        ld.seed, ld.epoch, ld.shuffle, ld.rank, ld.world, ld.B = 7, 0, True, 0, 1, 8
        chunks = ld._index(ds)
        arena = np.concatenate(chunks, 0)
        n_views = 3 if not ld.fullview else 6
        assert arena.shape == (3 * n_views * 3000, 3)
        for path, (s0, n0) in ld.view_range.items():
            assert np.array_equal(arena[s0:s0 + n0], np.load(path))
        assert ld._frames.shape == (len(ds), 18) and ld._labels.shape == (len(ds),)
        for item in range(len(ds)):                          # the per-dataset tables, item by item
            oi, gi = np.unravel_index(item, (len(ds.object), ds.grasp_amount_per_file))
            obj = ds.object[oi]
            grasp = np.load(ds.d_grasp[obj])[gi]
            ref = crop.frames_from_grasps_train(grasp[None, :], ds.transform[obj][1])[0]
            np.testing.assert_allclose(ld._frames[item], ref, rtol=0, atol=1e-15)
            lab = ds._label(grasp[-2] + grasp[-1] * 0.01)
            assert ld._labels[item] == (-1 if lab is None else lab)
        order, obj_of, pick, spans = ld._epoch_tables()
        assert sorted(order.tolist()) == list(range(len(ds))) and order.dtype == np.int32 and spans.dtype == np.int32
        for i, item in enumerate(order):
            oi = item // ds.grasp_amount_per_file
            assert obj_of[i] == oi
            files = ld.files[oi]
            assert sorted(files) == sorted(ds.d_pc[ds.transform[ds.object[oi]][0]])
            if ld.fullview:
                assert spans[i].shape == (3, 2)
                assert [tuple(r) for r in spans[i]] == [ld.view_range[files[j]] for j in pick[i]]
            else:
                assert tuple(spans[i]) == ld.view_range[files[pick[i]]]
        # the same (seed, epoch) reproduces the tables; another epoch reshuffles
        o2 = ld._epoch_tables()[0]
        assert np.array_equal(order, o2)
        ld.epoch = 1
        assert not np.array_equal(order, ld._epoch_tables()[0])
        # one process per GPU: strided shares, padded by wrap-around to equal lengths (DistributedSampler's contract)
        ld.epoch, ld.world = 0, 4
        shares = []
        for r in range(4):
            ld.rank = r
            shares.append(ld._epoch_tables()[0])
        assert all(len(sh) == 9 for sh in shares) and sorted(set(np.concatenate(shares).tolist())) == list(range(36))
        if not ld.fullview:
            assert (ld._labels == -1).any() and (ld._labels >= 0).any()      # thresholds 0.45 / 1.2 leave a None band
