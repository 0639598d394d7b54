"""GPU: the device-resident one-view training loader against the numpy Dataset mirror item by item (in-box counts,
None status, labels, and every output column being one of the item's in-box points), on the synthetic data tree."""
import numpy as np
import pytest
import torch

from tests import synth_dataset

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cls_name,kw", [("PointGraspOneViewDataset", dict(grasp_points_num=64, grasp_amount_per_file=12,
                                                                          thresh_good=0.6, thresh_bad=0.6, tag="train")),
                                         ("PointGraspOneViewDataset", dict(grasp_points_num=64, grasp_amount_per_file=12,
                                                                          thresh_good=0.45, thresh_bad=1.2, tag="train")),   # label None in between
                                         ("PointGraspOneViewMultiClassDataset", dict(grasp_points_num=200, grasp_amount_per_file=12,
                                                                                    thresh_good=0.5, thresh_bad=1.2, tag="test"))])
def test_device_loader_matches_dataset_semantics(cls_name, kw, tmp_path, monkeypatch, cuda_device):
    from pointnetgpd_amd import crop
    from pointnetgpd_amd.device_loader import DeviceGraspLoader
    from pointnetgpd_amd.model import dataset as ds_mod
    root = synth_dataset.build(str(tmp_path / "tree"))
    monkeypatch.setenv("PointNetGPD_FOLDER", root)
    ds = getattr(ds_mod, cls_name)(**kw)
    assert len(ds) == 36
    if cls_name.endswith("MultiClassDataset"):
        ds.min_point_limit = 165                 # in-box counts on this tree are 140..190: exercises the "< limit -> None" drop
    loader = DeviceGraspLoader(ds, batch_size=16, device=cuda_device, shuffle=True, seed=3)
    assert len(loader) == 3
    seen, total_kept = [], 0
    N = kw["grasp_points_num"]
    for data, target in loader:
        meta = loader.last_meta
        keep = meta["keep"].cpu().numpy()
        counts = meta["counts"].cpu().numpy()
        assert data.shape == (int(keep.sum()), 3, N) and data.dtype == torch.float32 and target.dtype == torch.int64
        row = 0
        for i, item in enumerate(meta["items"]):
            oi, gi = np.unravel_index(item, (len(ds.object), ds.grasp_amount_per_file))
            obj = ds.object[oi]
            grasp = np.load(ds.d_grasp[obj])[gi]
            pc = np.load(meta["views"][i])
            assert meta["views"][i] in ds.d_pc[ds.transform[obj][0]]
            frame = crop.frames_from_grasps_train(grasp[None, :], ds.transform[obj][1])[0]
            ind, pts = crop.collect_pc_numpy(frame, pc)                    # the Dataset mirror's own crop
            assert counts[i] == len(ind)
            label = ds._label(grasp[-2] + grasp[-1] * 0.01)
            expect_keep = len(ind) >= ds.min_point_limit and label is not None    # dataset.py:71-72 + label None + my_collate
            assert bool(keep[i]) == expect_keep
            if expect_keep:
                assert int(target[row]) == label
                cols = data[row].cpu().numpy().T                           # (N,3)
                ref32 = pts.astype(np.float32)
                d = np.abs(cols[:, None, :] - ref32[None, :, :]).max(2)
                assert d.min(1).max() <= 1e-7                              # every column is an in-box point of THIS view
                if len(ind) > N:
                    assert len(set(d.argmin(1).tolist())) == N              # without replacement (dataset.py:439)
                row += 1
        assert row == data.shape[0]
        seen += meta["items"].tolist()
        total_kept += row
    assert sorted(seen) == list(range(36)) and 0 < total_kept
    if kw["thresh_good"] == 0.45 or cls_name.endswith("MultiClassDataset"):
        assert total_kept < 36                    # something was dropped
    # a second epoch reshuffles; the same (seed, epoch) reproduces the batches bit for bit
    first = [b[0].clone() for b in DeviceGraspLoader(ds, 16, cuda_device, seed=3)]
    again = [b[0].clone() for b in DeviceGraspLoader(ds, 16, cuda_device, seed=3)]
    assert all(torch.equal(a, b) for a, b in zip(first, again))
    loader.set_epoch(1)
    other = [b[0] for b in loader]
    assert not all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(first, other))


def test_device_loader_fullview_matches_dataset_semantics(tmp_path, monkeypatch, cuda_device):
    """Full-view datasets: the per-sample cloud is the (device-built) gather list over the stacked random views;
    counts / None status / labels / membership checked against the mirror's numpy crop of arena[gather]."""
    from pointnetgpd_amd import crop
    from pointnetgpd_amd.device_loader import DeviceGraspLoader
    from pointnetgpd_amd.model import dataset as ds_mod
    root = synth_dataset.build(str(tmp_path / "tree"))
    monkeypatch.setenv("PointNetGPD_FOLDER", root)
    ds = ds_mod.PointGraspMultiClassDataset(obj_points_num=4000, grasp_points_num=100, pc_file_used_num=3,
                                            grasp_amount_per_file=12, thresh_good=0.5, thresh_bad=1.2, tag="train")
    ds.min_point_limit = 225                        # in-box counts here are ~185..250: exercises the drop
    loader = DeviceGraspLoader(ds, batch_size=16, device=cuda_device, shuffle=True, seed=5)
    arena = loader.arena.cpu().numpy()
    assert arena.shape == (3 * 6 * 3000, 3)          # every view (NP1 + NP3) of every object, once
    kept, dropped = 0, 0
    for data, target in loader:
        meta = loader.last_meta
        gather = meta["gather"].cpu().numpy()
        keep, counts = meta["keep"].cpu().numpy(), meta["counts"].cpu().numpy()
        assert gather.shape == (len(meta["items"]), 4000)
        row = 0
        for i, item in enumerate(meta["items"]):
            oi, gi = np.unravel_index(item, (len(ds.object), ds.grasp_amount_per_file))
            obj = ds.object[oi]
            assert len(meta["views"][i]) == 3 and all(v in ds.d_pc[ds.transform[obj][0]] for v in meta["views"][i])
            spans = [loader.view_range[v] for v in meta["views"][i]]
            inside = np.zeros(4000, bool)
            for s0, n0 in spans:
                inside |= (gather[i] >= s0) & (gather[i] < s0 + n0)
            assert inside.all()                                            # rows come from the chosen views only
            grasp = np.load(ds.d_grasp[obj])[gi]
            frame = crop.frames_from_grasps_train(grasp[None, :], ds.transform[obj][1])[0]
            ind, pts = crop.collect_pc_numpy(frame, arena[gather[i]])      # the mirror's crop of the same sample cloud
            assert counts[i] == len(ind)
            label = ds._label(grasp[-2] + grasp[-1] * 0.01)
            assert bool(keep[i]) == (len(ind) >= ds.min_point_limit)
            if keep[i]:
                assert int(target[row]) == label
                cols = data[row].cpu().numpy().T
                d = np.abs(cols[:, None, :] - pts.astype(np.float32)[None, :, :]).max(2)
                assert d.min(1).max() <= 1e-7
                row += 1
                kept += 1
            else:
                dropped += 1
        assert row == data.shape[0]
    assert kept > 0 and dropped > 0
    # view slots are drawn in proportion to their length and rows uniformly: with equal-length views every arena row
    # of the chosen views is equally likely -> mean index of a sample sits near the mean of its spans' centres
    g0 = gather[0].astype(np.float64)
    centres = np.mean([s0 + n0 / 2 for s0, n0 in [loader.view_range[v] for v in meta["views"][0]]])
    assert abs(g0.mean() - centres) < 4 * 3000 / np.sqrt(12) * np.sqrt(3) / np.sqrt(4000) * 6 + 1500


def test_device_loader_rejects_cpu_and_projection(tmp_path, monkeypatch, cuda_device):
    from pointnetgpd_amd.device_loader import DeviceGraspLoader
    from pointnetgpd_amd.model import dataset as ds_mod
    root = synth_dataset.build(str(tmp_path / "tree"))
    monkeypatch.setenv("PointNetGPD_FOLDER", root)
    one = ds_mod.PointGraspOneViewDataset(grasp_points_num=64, grasp_amount_per_file=12, thresh_good=0.6, thresh_bad=0.6, tag="train")
    with pytest.raises(RuntimeError, match="CUDA"):
        DeviceGraspLoader(one, 8, "cpu")
    one.projection = True
    with pytest.raises(NotImplementedError):
        DeviceGraspLoader(one, 8, cuda_device)


def test_cli_with_device_data(tmp_path, monkeypatch, cuda_device):
    """main_1v-style training on the (synthetic) on-disk tree with the HBM-resident loader feeding the HIP step."""
    from pointnetgpd_amd import mains
    root = synth_dataset.build(str(tmp_path / "tree"), grasps_per_obj=6500)   # the CLI indexes 6500 / 500 grasps per file
    monkeypatch.setenv("PointNetGPD_FOLDER", root)
    monkeypatch.setitem(mains.VARIANTS["1v"], "num_points", 64)         # the tree's crops hold ~150 points
    args = ["--mode", "train", "--epoch", "2", "--cuda", "--gpu", "0", "--batch-size", "8", "--num-workers", "0",
            "--max-batches", "3",
            "--model-path", str(tmp_path / "m"), "--log-dir", str(tmp_path / "l"), "--seed", "4", "--tag", "dd",
            "--device-data"]
    r = mains.run("1v", args)
    assert np.isfinite(r["test_loss"]) and 0.0 <= r["train_acc"] <= 1.0
    assert (tmp_path / "m" / "dd_1.model").exists()


@pytest.mark.parametrize("fullview", [False, True])
def test_overlapped_and_serial_loaders_yield_identical_batches(fullview, tmp_path, monkeypatch, cuda_device):
    """The side-stream schedule (batches produced ``prefetch`` ahead, under a busy consumer stream) and the serial one
    run the same launches with the same keys: every batch is bit-identical, whatever the prefetch depth, and the draws
    of a sample depend on its position in the epoch, not on the batch size."""
    from pointnetgpd_amd.device_loader import DeviceGraspLoader
    from pointnetgpd_amd.model import dataset as ds_mod
    root = synth_dataset.build(str(tmp_path / "tree"), grasps_per_obj=40)
    monkeypatch.setenv("PointNetGPD_FOLDER", root)
    if fullview:
        ds = ds_mod.PointGraspMultiClassDataset(obj_points_num=4000, grasp_points_num=100, pc_file_used_num=3,
                                                grasp_amount_per_file=40, thresh_good=0.5, thresh_bad=1.2, tag="train")
        ds.min_point_limit = 225                   # drops some samples -> ragged batches
    else:
        ds = ds_mod.PointGraspOneViewDataset(grasp_points_num=64, grasp_amount_per_file=40, thresh_good=0.45,
                                             thresh_bad=1.2, tag="train")     # label None between the thresholds
    serial = [(d.clone(), t.clone()) for d, t in DeviceGraspLoader(ds, 16, cuda_device, seed=7, prefetch=0)]
    assert len(serial) == 8 and any(d.shape[0] < 16 for d, _ in serial) and sum(d.shape[0] for d, _ in serial) > 0
    busy = torch.randn(2048, 2048, device=cuda_device)
    for depth in (1, 2, 4):
        got = []
        for d, t in DeviceGraspLoader(ds, 16, cuda_device, seed=7, prefetch=depth):
            for _ in range(3):
                busy = torch.tanh(busy @ busy * 1e-3)          # the consumer's stream is busy while the next batches are made
            got.append((d.clone(), t.clone()))
        assert len(got) == len(serial)
        for (d0, t0), (d1, t1) in zip(serial, got):
            assert d0.shape == d1.shape and torch.equal(d0, d1) and torch.equal(t0, t1)
    # batch size 8: the same samples, two batches for one
    half = [(d.clone(), t.clone()) for d, t in DeviceGraspLoader(ds, 8, cuda_device, seed=7, prefetch=2)]
    for i, (d0, t0) in enumerate(serial):
        pair = half[2 * i:2 * i + 2]                            # 120 items: the last 16-batch holds 8 = one 8-batch
        d1 = torch.cat([p[0] for p in pair]); t1 = torch.cat([p[1] for p in pair])
        assert torch.equal(d0, d1) and torch.equal(t0, t1)


def test_segmented_scan_of_long_sample_clouds_is_the_single_pass_scan(tmp_path, monkeypatch, cuda_device):
    """Full-view sample clouds of >= 16384 rows are scanned by 4 workgroups per sample in two launches (segment counts,
    then ordered writes): counts, in-box lists and therefore the batches are those of the one-workgroup scan, and the
    counts are the mirror's numpy crop of the same gathered rows."""
    from pointnetgpd_amd import crop
    from pointnetgpd_amd.device_loader import DeviceGraspLoader
    from pointnetgpd_amd.model import dataset as ds_mod
    root = synth_dataset.build(str(tmp_path / "tree"), grasps_per_obj=20, points=9000)
    monkeypatch.setenv("PointNetGPD_FOLDER", root)
    ds = ds_mod.PointGraspMultiClassDataset(obj_points_num=20001, grasp_points_num=300, pc_file_used_num=3,
                                            grasp_amount_per_file=20, thresh_good=0.5, thresh_bad=1.2, tag="train")
    out = {}
    for seg in (True, False):
        ld = DeviceGraspLoader(ds, 16, cuda_device, seed=11, prefetch=0, max_keep=16384)
        ld.segmented_scan = seg
        got = []
        for d, t in ld:
            m = ld.last_meta
            got.append((d.clone(), t.clone(), m["counts"].clone(), m["gather"].clone(), list(m["items"])))
        out[seg] = got
    arena = ld.arena.cpu().numpy()
    assert len(out[True]) == len(out[False]) and sum(d.shape[0] for d, *_ in out[True]) > 0
    for (d0, t0, c0, g0, it0), (d1, t1, c1, g1, it1) in zip(out[True], out[False]):
        assert torch.equal(g0, g1) and torch.equal(c0, c1) and it0 == it1
        assert d0.shape == d1.shape and torch.equal(d0, d1) and torch.equal(t0, t1)
    d0, t0, c0, g0, it0 = out[True][0]
    assert g0.shape[1] == 20001 and int(c0.max()) > 0
    for i, item in enumerate(it0[:6]):
        oi, gi = np.unravel_index(item, (len(ds.object), ds.grasp_amount_per_file))
        obj = ds.object[oi]
        frame = crop.frames_from_grasps_train(np.load(ds.d_grasp[obj])[gi][None, :], ds.transform[obj][1])[0]
        ind, _ = crop.collect_pc_numpy(frame, arena[g0[i].cpu().numpy()])
        assert int(c0[i]) == len(ind)


def test_device_loader_rank_shards_cover_the_epoch(tmp_path, monkeypatch, cuda_device):
    """One process per GPU: the ranks' strided shares of one epoch's permutation are disjoint up to the wrap-around
    padding and cover every item (DistributedSampler's contract, main_1v.py:120-128 under torchrun)."""
    from pointnetgpd_amd.device_loader import DeviceGraspLoader
    from pointnetgpd_amd.model import dataset as ds_mod
    root = synth_dataset.build(str(tmp_path / "tree"), grasps_per_obj=13)
    monkeypatch.setenv("PointNetGPD_FOLDER", root)
    ds = ds_mod.PointGraspOneViewDataset(grasp_points_num=64, grasp_amount_per_file=13, thresh_good=0.6, thresh_bad=0.6,
                                         tag="train")
    seen = []
    for rank in range(4):
        loader = DeviceGraspLoader(ds, 4, cuda_device, seed=2, rank=rank, world=4)
        assert len(loader) == 3                     # 39 items -> 10 per rank (1 wrapped) -> 3 batches
        mine = []
        for _ in loader:
            mine += loader.last_meta["items"].tolist()
        assert len(mine) == 10
        seen += mine
    assert sorted(set(seen)) == list(range(39)) and len(seen) == 40
