"""GPU parity of the batched crop / resample kernels and the grasp-scoring service against the numpy
oracle (bit-exact index sets, fp64 transforms rounded to fp32) and the reference's golden crop record."""
import os

import numpy as np
import pytest
import torch

from oracle import crop_oracle as co
from oracle import pointnet_oracle as po
from tests.helpers import GOLDEN, build_model, state_dict_cpu

pytestmark = pytest.mark.gpu


def _scene(G, P, seed):
    """SURVEY.md §8d config-5 style scene: cloud U(-0.15,0.15)^2 x U(0,0.2), random grasp frames whose
    bottom centre sits 5 cm behind a cloud point."""
    rng = np.random.default_rng(seed)
    pc = np.stack([rng.uniform(-0.15, 0.15, P), rng.uniform(-0.15, 0.15, P), rng.uniform(0, 0.2, P)], 1)
    q = rng.normal(size=(G, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], 1),
                  np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], 1),
                  np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)], 1)
    approach, binormal, minor = R[:, :, 0], R[:, :, 1], R[:, :, 2]
    bottom = pc[rng.integers(0, P, G)] - 0.05 * approach
    grasps = np.stack([bottom, approach, binormal, minor, bottom], 1)
    return pc, grasps


def test_crop_kernel_vs_reference_record(cuda_device):
    """Training-style crop on the reference's recorded cases: counts and index sets exactly."""
    from pointnetgpd_amd import crop
    fx = np.load(os.path.join(GOLDEN, "crop_train.npz"))
    for c in range(len(fx["grasps"])):
        frames = crop.frames_from_grasps_train(fx["grasps"][c][None], fx["Ts"][c])
        cloud = torch.from_numpy(fx["pcs"][c]).to(cuda_device)            # fp64 cloud, like the .npy files
        counts, idx = crop.crop_count_compact(cloud, torch.from_numpy(frames).to(cuda_device), max_keep=1024)
        n = int(counts[0])
        assert n == int(fx["counts"][c])
        np.testing.assert_array_equal(idx[0, :n].cpu().numpy(), fx[f"ind_{c}"])
        # resample with an injected draw: out[:, j] == pts[sel[j]] rounded to fp32; None <-> invalid
        N = 64
        sel = torch.from_numpy(np.random.default_rng(c).integers(0, max(n, 1), size=(1, N)).astype(np.int32)).to(cuda_device)
        out, valid = crop.crop_resample(cloud, torch.from_numpy(frames).to(cuda_device), counts, idx, N,
                                        crop.MODE_TRAIN, crop.MIN_POINT_LIMIT, sel=sel)
        assert bool(valid[0]) == (not bool(fx["is_none"][c]))
        if valid[0]:
            ref = fx[f"pts_{c}"][sel[0].cpu().numpy()].T.astype(np.float32)
            np.testing.assert_allclose(out[0].cpu().numpy(), ref, rtol=0, atol=1e-7)
        else:
            assert float(out.abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_infer_crop_vs_oracle(dtype, cuda_device):
    from pointnetgpd_amd import crop
    pc, grasps = _scene(64, 20000, 7)
    pc = pc.astype(np.float32 if dtype == torch.float32 else np.float64)
    ind_ref, pts_ref = co.collect_pc_infer(grasps, pc)
    frames = torch.from_numpy(crop.frames_from_grasps_infer(grasps)).to(cuda_device)
    cloud = torch.from_numpy(pc).to(cuda_device)
    counts, idx = crop.crop_count_compact(cloud, frames, max_keep=2048)
    assert counts.max().item() <= 2048
    for g in range(len(grasps)):
        n = int(counts[g])
        assert n == len(ind_ref[g])
        np.testing.assert_array_equal(idx[g, :n].cpu().numpy(), ind_ref[g])
    # device RNG resampling: every output column is one of the grasp's in-box points (fp32-rounded);
    # without replacement (m >= N) -> all distinct point indices
    N = 32
    out, valid = crop.crop_resample(cloud, frames, counts, idx, N, crop.MODE_INFER, crop.MIN_POINTS_TO_NET, seed=11)
    out2, _ = crop.crop_resample(cloud, frames, counts, idx, N, crop.MODE_INFER, crop.MIN_POINTS_TO_NET, seed=11)
    assert torch.equal(out, out2)                                   # reproducible per seed
    out3, _ = crop.crop_resample(cloud, frames, counts, idx, N, crop.MODE_INFER, crop.MIN_POINTS_TO_NET, seed=12)
    assert not torch.equal(out, out3)
    for g in range(len(grasps)):
        m = len(ind_ref[g])
        assert bool(valid[g]) == (m >= crop.MIN_POINTS_TO_NET)
        if not valid[g]:
            continue
        ref32 = pts_ref[g].astype(np.float32)
        cols = out[g].cpu().numpy().T                                # (N,3)
        # index sets are exact; coordinates agree with numpy's BLAS product to 1 fp32 ulp (the 3-term dot
        # products are summed in a different order there): 1e-8 absolute at |y| <= 0.125
        match = [np.where(np.abs(ref32 - c).max(1) <= 1e-8)[0] for c in cols]
        assert all(len(mm) >= 1 for mm in match)
        if m >= N:
            assert len({int(mm[0]) for mm in match}) == N


def test_max_keep_truncation(cuda_device):
    from pointnetgpd_amd import crop
    pc, grasps = _scene(4, 60000, 9)
    frames = torch.from_numpy(crop.frames_from_grasps_infer(grasps)).to(cuda_device)
    cloud = torch.from_numpy(pc.astype(np.float32)).to(cuda_device)
    counts, idx = crop.crop_count_compact(cloud, frames, max_keep=16)
    ind_ref, _ = co.collect_pc_infer(grasps, pc.astype(np.float32))
    for g in range(4):
        assert int(counts[g]) == len(ind_ref[g])                    # count is the true count
        np.testing.assert_array_equal(idx[g, :min(16, len(ind_ref[g]))].cpu().numpy(), ind_ref[g][:16])


def test_scorer_end_to_end(cuda_device):
    """GraspScorer == the reference's per-grasp loop semantics, evaluated by the oracle on the very points
    the crop kernel produced (kinect2grasp.py:454-497 with repeat=1)."""
    from pointnetgpd_amd import crop
    from pointnetgpd_amd.scoring import GraspScorer, test_network
    N, k = 64, 3
    m = build_model(N, k, 33, 4700).eval()
    sd = state_dict_cpu(m)
    pc, grasps = _scene(40, 30000, 21)
    pc32 = pc.astype(np.float32)
    mg = m.to(cuda_device)
    scorer = GraspScorer(mg, num_points=N, repeat=1, batch=16, seed=5)
    res = scorer.score(pc32, grasps)
    ind_ref, _ = co.collect_pc_infer(grasps, pc32)
    counts_ref = np.array([len(i) for i in ind_ref])
    np.testing.assert_array_equal(res["counts"].cpu().numpy(), counts_ref)
    np.testing.assert_array_equal(res["valid"].cpu().numpy(), counts_ref >= 20)
    # re-run the crop with the scorer's keys to get the exact clouds it scored (the scorer crops over the scene's
    # spatial index: lists in the order of the Morton-sorted cloud, resample against that cloud)
    from pointnetgpd_amd.gpg import CloudIndex
    frames = torch.from_numpy(crop.frames_from_grasps_infer(grasps)).to(cuda_device)
    cloud = torch.from_numpy(pc32).to(cuda_device)
    index = CloudIndex(cloud)
    counts, idx = crop.crop_count_compact_indexed(index, frames, 4096)
    probs = res["probs"][0].cpu().numpy()
    for s in range(0, 40, 16):
        e = min(40, s + 16)
        pts, v = crop.crop_resample(index.cloud, frames[s:e], counts[s:e], idx[s:e], N, crop.MODE_INFER, 20,
                                    seed=5 * 1000003, g_base=s)
        with torch.no_grad():
            lp_ref, _ = po.forward_torch(sd, pts.cpu())
        np.testing.assert_allclose(probs[s:e], lp_ref.softmax(1).numpy(), atol=1e-3)
    pred = res["pred"].cpu().numpy(); score = res["score"].cpu().numpy(); valid = res["valid"].cpu().numpy()
    assert (pred[~valid] == 0).all() and (score[~valid] == 0).all()
    np.testing.assert_array_equal(pred[valid], probs[valid].argmax(1))
    np.testing.assert_allclose(score[valid], probs[valid][:, 2], atol=1e-6)      # best class of a 3-class model
    good = res["good"].cpu().numpy()
    np.testing.assert_array_equal(good, valid & (pred == 2))
    order = res["order"].cpu().numpy()
    assert set(order.tolist()) == set(np.nonzero(good)[0].tolist())
    assert (np.diff(score[order]) <= 1e-7).all()
    # test_network (main_test.py:59-69) on one cloud agrees with the batched path
    g0 = int(np.nonzero(valid)[0][0])
    # the draw of candidate g0 depends on (seed, global index) only: resample it ALONE, as the reference's loop does
    pts, _ = crop.crop_resample(index.cloud, frames[g0:g0 + 1], counts[g0:g0 + 1], idx[g0:g0 + 1], N,
                                crop.MODE_INFER, 20, seed=5 * 1000003, g_base=g0)
    p1, pr1 = test_network(mg, pts[0].cpu().numpy().T)
    assert int(p1) == int(pred[g0])
    np.testing.assert_allclose(pr1[0], probs[g0], atol=1e-5)


def test_scorer_vote_repeat(cuda_device):
    from pointnetgpd_amd.scoring import GraspScorer
    m = build_model(32, 2, 34, 4701).eval().to(cuda_device)
    pc, grasps = _scene(24, 30000, 22)
    res = GraspScorer(m, num_points=32, repeat=5, batch=8, seed=1).score(pc.astype(np.float32), grasps)
    probs = res["probs"].cpu().numpy()                               # (5, G, 2)
    votes = probs.argmax(2)
    valid = res["valid"].cpu().numpy()
    for g in np.nonzero(valid)[0]:
        vals, cnts = np.unique(votes[:, g], return_counts=True)
        maj = vals[np.argmax(cnts)]                                  # scipy.stats.mode: smallest of the most frequent
        assert int(res["pred"][g]) == int(maj)
        agree = votes[:, g] == maj
        np.testing.assert_allclose(float(res["score"][g]), probs[agree, g, 1].mean(), atol=1e-6)


def test_graphed_forward_matches_eager(cuda_device):
    from pointnetgpd_amd.scoring import GraphedForward
    from tests.helpers import synth_cloud
    m = build_model(500, 3, 35, 4702).eval().to(cuda_device)
    gf = GraphedForward(m, batch=8, num_points=500)
    for seed in (1, 2):
        x = synth_cloud(8, 500, 2000 + seed, "box").to(cuda_device)
        with torch.no_grad():
            lp_e, tr_e = m(x)
        lp_g, tr_g = gf(x)
        assert torch.equal(lp_e, lp_g) and torch.equal(tr_e, tr_g)
    with pytest.raises(RuntimeError, match="captured for"):
        gf(torch.zeros(4, 3, 500, device=cuda_device))


def test_score_scene_distributed_single_rank(cuda_device):
    """world_size 1 (no process group): the sharded entry returns exactly what GraspScorer.score returns."""
    from pointnetgpd_amd.scoring import GraspScorer, score_scene_distributed
    m = build_model(48, 2, 36, 4703).eval().to(cuda_device)
    pc, grasps = _scene(33, 30000, 23)
    scorer = GraspScorer(m, num_points=48, repeat=1, batch=16, seed=3)
    ref = scorer.score(pc.astype(np.float32), grasps)
    res = score_scene_distributed(scorer.score, pc.astype(np.float32), grasps)
    assert torch.equal(res["pred"], ref["pred"]) and torch.equal(res["valid"], ref["valid"])
    assert torch.equal(res["counts"], ref["counts"]) and torch.equal(res["score"], ref["score"])
    sc = res["score"][res["order"]]
    assert (sc[:-1] >= sc[1:]).all()


def test_infer_crop_kernel_vs_executed_reference(cuda_device):
    """Crop kernel vs the record of kinect2grasp.py:178-258 executed by oracle/make_golden_gpg.py: index sets exact."""
    from pointnetgpd_amd import crop
    from tests.test_gpg_cpu import _crop_infer_cases
    for pc, grasps, counts_ref, ind_ref, head, _ in _crop_infer_cases():
        frames = torch.from_numpy(crop.frames_from_grasps_infer(grasps)).to(cuda_device)
        cloud = torch.from_numpy(pc).to(cuda_device)
        counts, idx = crop.crop_count_compact(cloud, frames, max_keep=2048)
        np.testing.assert_array_equal(counts.cpu().numpy(), counts_ref)
        idx = idx.cpu().numpy()
        for g in range(len(grasps)):
            np.testing.assert_array_equal(idx[g, :counts_ref[g]], ind_ref[g])
        # hand-frame coordinates of the first in-box points (fp32-rounded by the resample kernel)
        N = 3
        sel = torch.arange(N, dtype=torch.int32, device=cuda_device).repeat(len(grasps), 1)
        out, valid = crop.crop_resample(cloud, frames, counts, torch.from_numpy(idx).to(cuda_device), N,
                                        crop.MODE_INFER, 3, sel=sel)
        for g in np.nonzero(counts_ref >= 3)[0]:
            np.testing.assert_allclose(out[g].cpu().numpy().T, head[g].astype(np.float32), rtol=0, atol=1e-8)


def test_resample_without_replacement_is_a_uniform_subset(cuda_device):
    """4,000 identical grasps (the draw is keyed by the grasp index): every draw is N DISTINCT in-box points in
    ascending index order, and each in-box point is chosen with frequency N/m (5 sigma)."""
    from pointnetgpd_amd import crop
    pc, grasps = _scene(1, 3000, 51)
    pc32 = pc.astype(np.float32)
    G, N = 4000, 16
    frames1 = crop.frames_from_grasps_infer(grasps)
    frames = torch.from_numpy(np.repeat(frames1, G, 0)).to(cuda_device)
    cloud = torch.from_numpy(pc32).to(cuda_device)
    counts, idx = crop.crop_count_compact(cloud, frames, max_keep=512)
    m = int(counts[0])
    assert N < m <= 512 and bool((counts == m).all())
    out, valid = crop.crop_resample(cloud, frames, counts, idx, N, crop.MODE_INFER, 1, seed=123)
    assert bool(valid.all())
    ind_ref, pts_ref = co.collect_pc_infer(grasps, pc32)
    ref32 = torch.from_numpy(pts_ref[0].astype(np.float32)).to(cuda_device)          # (m,3)
    # map every output column back to its in-box rank (exact fp32 match up to 1e-8, see test_infer_crop_vs_oracle)
    d = (out.permute(0, 2, 1).unsqueeze(2) - ref32.view(1, 1, m, 3)).abs().amax(3)  # (G,N,m)
    rank = d.argmin(2)
    assert float(d.amin(2).max()) <= 1e-8
    assert bool((rank[:, 1:] > rank[:, :-1]).all())                                  # distinct + ascending
    freq = torch.bincount(rank.reshape(-1), minlength=m).double() / G
    p = N / m
    assert float((freq - p).abs().max()) < 5 * np.sqrt(p * (1 - p) / G)


@pytest.mark.parametrize("max_keep,N,mode_name", [(128, 16, "infer"), (64, 32, "train"), (48, 500, "train")])
def test_resample_uniform_over_all_points_when_count_exceeds_max_keep(max_keep, N, mode_name, cuda_device):
    """A hand holding MORE in-box points than ``max_keep``: the draw is still uniform over ALL of them (reference:
    np.random.choice over the whole in-box set, kinect2grasp.py:473-478 / dataset.py:438-444), not over the first
    max_keep in index order.  Third case: count <= N with max_keep < count -> the with-replacement branch, also over
    every in-box point."""
    from pointnetgpd_amd import crop
    pc, grasps = _scene(1, 40000, 77)
    pc32 = pc.astype(np.float32)
    G = 3000
    mode = crop.MODE_INFER if mode_name == "infer" else crop.MODE_TRAIN
    frames1 = crop.frames_from_grasps_infer(grasps)
    frames = torch.from_numpy(np.repeat(frames1, G, 0)).to(cuda_device)
    cloud = torch.from_numpy(pc32).to(cuda_device)
    counts, idx = crop.crop_count_compact(cloud, frames, max_keep=max_keep)
    m = int(counts[0])
    assert m > max_keep, (m, max_keep)
    out, valid = crop.crop_resample(cloud, frames, counts, idx, N, mode, 1, seed=4321)
    assert bool(valid.all())
    ind_ref, pts_ref = co.collect_pc_infer(grasps, pc32)
    assert len(ind_ref[0]) == m
    ref32 = torch.from_numpy(pts_ref[0].astype(np.float32)).to(cuda_device)          # (m,3)
    rank = torch.empty(G, N, dtype=torch.long, device=cuda_device)
    for s in range(0, G, 250):                                                       # (250,N,m) distance blocks
        d = (out[s:s + 250].permute(0, 2, 1).unsqueeze(2) - ref32.view(1, 1, m, 3)).abs().amax(3)
        assert float(d.amin(2).max()) <= 1e-8                                        # every column IS an in-box point
        rank[s:s + 250] = d.argmin(2)
    without = m > N if mode == crop.MODE_TRAIN else m >= N
    if without:
        assert bool((rank[:, 1:] > rank[:, :-1]).all())                              # distinct, ascending index order
    freq = torch.bincount(rank.reshape(-1), minlength=m).double() / G
    p = N / m
    sd = np.sqrt(p * (1 - p) / G) if without else np.sqrt(N * (1 / m) * (1 - 1 / m) / G)
    assert float((freq - p).abs().max()) < 5.5 * sd, (float((freq - p).abs().max()), sd)
    # the points beyond the truncated list are drawn as often as the ones inside it
    assert abs(float(freq[max_keep:].mean()) / float(freq[:max_keep].mean()) - 1.0) < 0.05


@pytest.mark.parametrize("dtype,P,G,max_keep,N", [(np.float32, 30000, 50, 4096, 64), (np.float64, 5000, 33, 256, 100),
                                                  (np.float32, 130, 7, 64, 16), (np.float32, 50000, 24, 8192, 1024)])
def test_indexed_crop_equals_plain_crop(dtype, P, G, max_keep, N, cuda_device):
    """The crop over the scene's spatial index (``pngpd_crop_count_compact_indexed``: chunk spheres against the hand's
    box, then the same fp64 per-point test) against the whole-cloud kernel: identical counts; the index lists hold the
    same POINTS (sorted positions mapped back through ``index.order``), in ascending sorted position; truncation at
    ``max_keep`` included.  And the one-launch form (``pngpd_crop_indexed``: the list never leaves LDS) is bit-identical to
    indexed count + resample against the sorted cloud — both resample rules, overflowing hands, invalid hands."""
    from pointnetgpd_amd import crop
    from pointnetgpd_amd.gpg import CloudIndex
    pc, grasps = _scene(G, P, 31)
    pc = pc.astype(dtype)
    frames = torch.from_numpy(crop.frames_from_grasps_infer(grasps)).to(cuda_device)
    cloud = torch.from_numpy(pc).to(cuda_device)
    index = CloudIndex(cloud)
    c0, i0 = crop.crop_count_compact(cloud, frames, max_keep=P)                      # untruncated reference lists
    c1, i1 = crop.crop_count_compact_indexed(index, frames, max_keep=max_keep)
    assert torch.equal(c0, c1) and int(c0.max()) > 0
    order = index.order.long()
    for g in range(G):
        n = int(c0[g])
        m = min(n, max_keep)
        pos = i1[g, :m].long()
        assert bool((pos[1:] > pos[:-1]).all())                                      # ascending sorted position
        if n <= max_keep:
            assert torch.equal(torch.sort(order[pos]).values, i0[g, :n].long())      # the same points
        else:
            full = torch.sort(torch.nonzero(torch.isin(order, i0[g, :n].long())).squeeze(1)).values
            assert torch.equal(pos, full[:m])                                        # the first max_keep in sorted order
    for mode in (crop.MODE_INFER, crop.MODE_TRAIN):
        ref, vref = crop.crop_resample(index.cloud, frames, c1, i1, N, mode, 20, seed=77, g_base=1000)
        out, cnt, v = crop.crop_indexed(index, frames, N, mode, 20, seed=77, g_base=1000, max_keep=max_keep)
        assert torch.equal(cnt, c1) and torch.equal(v, vref) and torch.equal(out, ref)
    # every output column is an in-box point of its hand (hand frame): inside the box, strictly
    f = frames.cpu().numpy()
    o = out.cpu().numpy()
    for g in np.nonzero(v.cpu().numpy())[0][:10]:
        lo, hi = f[g, 12:15], f[g, 15:18]
        assert (o[g] > lo[:, None] - 1e-6).all() and (o[g] < hi[:, None] + 1e-6).all()


def _dense_scene(G, P, seed):
    """A small object inside the hand's reach (as a table-top object is): most of the cloud lies in every hand's box."""
    pc, grasps = _scene(G, P, seed)
    pc = pc * np.array([0.2, 0.2, 0.3])                      # 6 x 6 x 6 cm blob
    grasps = grasps.copy()
    grasps[:, 0] = pc[np.random.default_rng(seed + 1).integers(0, P, G)] - 0.05 * grasps[:, 1]
    grasps[:, 4] = grasps[:, 0]
    return pc, grasps


@pytest.mark.parametrize("dtype,P,N,mode_name", [(np.float32, 50000, 1024, "infer"), (np.float64, 20001, 750, "train"),
                                                 (np.float32, 9000, 64, "infer")])
def test_one_scan_overflow_path_equals_the_radix_path(dtype, P, N, mode_name, cuda_device):
    """Hands that hold more in-box points than ``max_keep`` (every hand on a dense table-top scene): the one-scan path
    (in-box bitmask in LDS + ~N + 8 sqrt(N) candidate keys, round 6) must pick the very subset the radix path picks —
    forced here by a ``max_keep`` whose scratch cannot hold the bitmask — in the same column order, bit for bit; with
    ``ranges`` (a grasp's own sub-cloud) as well."""
    from pointnetgpd_amd import crop
    G = 96
    pc, grasps = _dense_scene(G, P, 91)
    mode = crop.MODE_INFER if mode_name == "infer" else crop.MODE_TRAIN
    cloud = torch.from_numpy(pc.astype(dtype)).to(cuda_device)
    frames = torch.from_numpy(crop.frames_from_grasps_infer(grasps)).to(cuda_device)
    big, small = 8192, max(N, 256)
    cb, ib = crop.crop_count_compact(cloud, frames, max_keep=big)
    cs, is_ = crop.crop_count_compact(cloud, frames, max_keep=small)
    assert torch.equal(cb, cs)
    over = cb > big
    assert int(over.sum()) >= (G // 2 if P > 10000 else 0) and int((cb > small).sum()) >= G // 2
    ob, vb = crop.crop_resample(cloud, frames, cb, ib, N, mode, 20, seed=99, g_base=5000)
    os_, vs = crop.crop_resample(cloud, frames, cs, is_, N, mode, 20, seed=99, g_base=5000)
    assert torch.equal(vb, vs)
    # identical wherever both runs re-scan the cloud (count > both list sizes); elsewhere the big list is used as is,
    # which the radix re-scan of the small run must reproduce too (same keys, same ranks)
    assert torch.equal(ob, os_)
    # a grasp's own sub-cloud through `ranges`
    half = P // 2
    ranges = torch.tensor([[0, half] if g % 2 == 0 else [half, P - half] for g in range(G)], dtype=torch.int32,
                          device=cuda_device)
    c2, i2 = crop.crop_count_compact_ranges(cloud, frames, ranges, max_keep=big) if hasattr(crop, "crop_count_compact_ranges") else (None, None)
    if c2 is not None:
        c3, i3 = crop.crop_count_compact_ranges(cloud, frames, ranges, max_keep=small)
        o2, _ = crop.crop_resample(cloud, frames, c2, i2, N, mode, 20, seed=7, ranges=ranges)
        o3, _ = crop.crop_resample(cloud, frames, c3, i3, N, mode, 20, seed=7, ranges=ranges)
        assert torch.equal(o2, o3)


def test_one_scan_overflow_path_is_a_uniform_subset(cuda_device):
    """Statistics of the one-scan path itself: 3,000 identical hands holding m > max_keep points, N = 64 drawn without
    replacement: distinct, ascending, every in-box point with frequency N / m (5.5 sigma), beyond the list as within."""
    from pointnetgpd_amd import crop
    pc, grasps = _dense_scene(1, 12000, 17)
    pc32 = pc.astype(np.float32)
    G, N, max_keep = 3000, 64, 4096
    frames = torch.from_numpy(np.repeat(crop.frames_from_grasps_infer(grasps), G, 0)).to(cuda_device)
    cloud = torch.from_numpy(pc32).to(cuda_device)
    counts, idx = crop.crop_count_compact(cloud, frames, max_keep=max_keep)
    m = int(counts[0])
    assert m > max_keep, m
    out, valid = crop.crop_resample(cloud, frames, counts, idx, N, crop.MODE_INFER, 1, seed=4321)
    assert bool(valid.all())
    ind_ref, pts_ref = co.collect_pc_infer(grasps, pc32)
    assert len(ind_ref[0]) == m
    ref32 = torch.from_numpy(pts_ref[0].astype(np.float32)).to(cuda_device)
    rank = torch.empty(G, N, dtype=torch.long, device=cuda_device)
    for s in range(0, G, 100):
        d = (out[s:s + 100].permute(0, 2, 1).unsqueeze(2) - ref32.view(1, 1, m, 3)).abs().amax(3)
        assert float(d.amin(2).max()) <= 1e-8
        rank[s:s + 100] = d.argmin(2)
    assert bool((rank[:, 1:] > rank[:, :-1]).all())
    freq = torch.bincount(rank.reshape(-1), minlength=m).double() / G
    p = N / m
    assert float((freq - p).abs().max()) < 5.5 * np.sqrt(p * (1 - p) / G)
    assert abs(float(freq[max_keep:].mean()) / float(freq[:max_keep].mean()) - 1.0) < 0.05
