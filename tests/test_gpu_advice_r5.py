"""Regression tests for the round-5 advisor findings: short grasp files, non-finite scene points in the
indexed crop, non-unit frames in the indexed crop's broad phase, the `rows` / `out` bound of crop_resample, and a
FlatAdam slice that leaves the fused regime."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import synth_dataset
from tests.helpers import build_model, synth_cloud
from tests.test_gpu_crop_scoring import _scene

pytestmark = pytest.mark.gpu


def test_device_loader_raises_on_a_short_grasp_file(tmp_path, monkeypatch, cuda_device):
    """dataset.py:421-430 indexes np.load(...)[grasp_ind] with grasp_ind < grasp_amount_per_file: a short file is an
    IndexError in the reference; the HBM-resident loader raises the same up front instead of training on garbage."""
    from pointnetgpd_amd.device_loader import DeviceGraspLoader
    from pointnetgpd_amd.model import dataset as ds_mod
    root = synth_dataset.build(str(tmp_path / "tree"), grasps_per_obj=12)
    monkeypatch.setenv("PointNetGPD_FOLDER", root)
    one = ds_mod.PointGraspOneViewDataset(grasp_points_num=64, grasp_amount_per_file=20, thresh_good=0.6,
                                          thresh_bad=0.6, tag="train")
    with pytest.raises(IndexError, match="grasp_amount_per_file"):
        DeviceGraspLoader(one, 8, cuda_device)


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_indexed_crop_drops_only_the_non_finite_point(bad, cuda_device):
    """A NaN / Inf point fails every strict face test (kinect2grasp.py:218-229 drops exactly that point); it must not
    take its 64-point chunk — or the whole Morton order — with it."""
    from pointnetgpd_amd import crop
    from pointnetgpd_amd.gpg import CloudIndex
    pc, grasps = _scene(40, 20000, 11)
    pc = pc.astype(np.float32)
    frames = torch.from_numpy(crop.frames_from_grasps_infer(grasps)).to(cuda_device)
    clean = torch.from_numpy(pc).to(cuda_device)
    c_clean, _ = crop.crop_count_compact(clean, frames, max_keep=20000)
    hit = torch.nonzero(c_clean > 64)[0].item()
    _, i_clean = crop.crop_count_compact(clean, frames[hit:hit + 1], max_keep=20000)
    victims = i_clean[0, :3].long()                       # three in-box points of one hand become non-finite
    dirty = clean.clone()
    dirty[victims[0], 0] = bad
    dirty[victims[1], 1] = -bad if bad == bad else bad
    dirty[victims[2]] = bad
    c_brute, i_brute = crop.crop_count_compact(dirty, frames, max_keep=20000)
    index = CloudIndex(dirty)
    c_idx, i_idx = crop.crop_count_compact_indexed(index, frames, max_keep=20000)
    assert torch.equal(c_brute, c_idx)
    assert int(c_clean[hit] - c_idx[hit]) == 3            # exactly the three points, not their chunks
    order = index.order.long()
    for g in range(frames.shape[0]):
        n = int(c_idx[g])
        assert torch.equal(torch.sort(order[i_idx[g, :n].long()]).values, i_brute[g, :n].long())
    assert torch.equal(CloudIndex(dirty).order, index.order)          # stable sort: the same order every time


def test_indexed_crop_broad_phase_with_non_unit_frames(cuda_device):
    """pngpd_crop_count_compact_indexed is a public entry taking arbitrary (G,18) frames: with rows of M scaled (a
    mesh->cloud transform that includes scale) the sphere cull must stay conservative — counts equal the un-indexed
    kernel's."""
    from pointnetgpd_amd import crop
    from pointnetgpd_amd.gpg import CloudIndex
    pc, grasps = _scene(60, 30000, 13)
    cloud = torch.from_numpy(pc.astype(np.float32)).to(cuda_device)
    f = crop.frames_from_grasps_infer(grasps)
    scale = np.random.default_rng(3).uniform(1.5, 6.0, size=(f.shape[0], 3))
    f[:, 3:12] = (f[:, 3:12].reshape(-1, 3, 3) * scale[:, :, None]).reshape(-1, 9)     # rows of M scaled
    f[:, 12:15] *= scale; f[:, 15:18] *= scale                                          # the same box in frame units
    frames = torch.from_numpy(f).to(cuda_device)
    c0, _ = crop.crop_count_compact(cloud, frames, max_keep=30000)
    c1, _ = crop.crop_count_compact_indexed(CloudIndex(cloud), frames, max_keep=30000)
    assert int(c0.max()) > 0 and torch.equal(c0, c1)


def test_crop_resample_rows_need_a_full_size_out(cuda_device):
    from pointnetgpd_amd import crop
    pc, grasps = _scene(16, 5000, 17)
    cloud = torch.from_numpy(pc.astype(np.float32)).to(cuda_device)
    frames = torch.from_numpy(crop.frames_from_grasps_infer(grasps)).to(cuda_device)
    counts, idx = crop.crop_count_compact(cloud, frames, max_keep=1024)
    rows = torch.arange(16, device=cuda_device, dtype=torch.int32)
    small = torch.empty(8, 3, 32, device=cuda_device)
    with pytest.raises(RuntimeError, match=">= G"):
        crop.crop_resample(cloud, frames, counts, idx, 32, crop.MODE_INFER, 1, rows=rows, out=small)
    full = torch.empty(16, 3, 32, device=cuda_device)
    out, _ = crop.crop_resample(cloud, frames, counts, idx, 32, crop.MODE_INFER, 1, rows=rows, out=full)
    assert out.data_ptr() == full.data_ptr()


def test_flat_adam_slice_leaving_the_fused_regime(cuda_device):
    """A piece whose gradients used to be OVERWRITTEN by the fused backward (exempt from zero_grad) and then goes
    through the pass-by-pass path (autograd accumulation) must not accumulate onto the stale gradient."""
    from pointnetgpd_amd.optim import FlatAdam
    dev = cuda_device
    x = synth_cloud(8, 128, 3, "gauss").to(dev)
    y = (torch.arange(8, device=dev) % 2).long()
    m = build_model(128, 2, 5, 6).to(dev).train()
    opt = FlatAdam(m.parameters(), lr=0.0)                 # lr 0: the weights stay put, only gradients matter

    def step():
        opt.zero_grad()
        lp, _ = m(x)
        F.nll_loss(lp, y).backward()
        g = opt.flat_g.clone()
        opt.step()
        return g
    g_fused = step()
    assert any(opt._grouped)
    m.set_precision(sequencing="passes")
    g_passes = step()                                      # must equal the fused gradient, not twice it
    assert not any(opt._grouped)
    assert torch.equal(g_passes, g_fused)
    g_again = step()
    assert torch.equal(g_again, g_fused)
    m.set_precision(sequencing="fused")
    assert torch.equal(step(), g_fused)
