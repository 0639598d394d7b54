#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference.

Run in the build container only (needs /root/reference):

    python oracle/make_golden.py

It imports /root/reference/PointNetGPD/model/pointnet.py (PointNetCls) and
/root/reference/PointNetGPD/model/dataset.py (BaseGraspDataset.collect_pc, with
``open3d`` stubbed — it is imported at dataset.py:7 but only used by the GPD
projection path) and records inputs + outputs as small .npz fixtures.

Model weights are NOT stored (6.4 MB per model).  A fixture records the recipe
``torch.manual_seed(seed_w); PointNetCls(num_points, 3, k)`` followed by
``oracle.pointnet_oracle.randomize_bn_(state_dict, seed_bn)`` together with a
per-tensor fp64 checksum (sum, abs-sum) of the resulting reference state_dict, so a
consumer that rebuilds the weights (with its own mirror of the module) can prove it
holds bit-identical values before comparing outputs.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "PointNetGPD"))
os.environ["PointNetGPD_FOLDER"] = REF
sys.modules.setdefault("open3d", types.ModuleType("open3d"))

from model.pointnet import PointNetCls  # noqa: E402  (the reference)
from model import dataset as ref_dataset  # noqa: E402

from oracle.pointnet_oracle import randomize_bn_  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def checksums(sd):
    names = sorted(k for k in sd.keys() if not k.endswith("num_batches_tracked"))
    cs = np.array([[sd[k].double().sum().item(), sd[k].double().abs().sum().item()] for k in names])
    return np.array(names), cs


def synth_cloud(b, n, seed, kind):
    """SURVEY.md §8d synthetic in-gripper clouds, layout (B,3,N) fp32."""
    g = torch.Generator().manual_seed(seed)
    if kind == "box":       # U over the training crop box, w = 0.085
        w = 0.085
        u = torch.rand(b, 3, n, generator=g) - 0.5
        scale = torch.tensor([w / 2, w, w / 2]).view(1, 3, 1)
        return (u * scale).float().contiguous()
    if kind == "gauss":
        return (torch.randn(b, 3, n, generator=g) * 0.02).float().contiguous()
    if kind == "randn":
        return torch.randn(b, 3, n, generator=g).float().contiguous()
    raise ValueError(kind)


def build_ref(num_points, k, seed_w, seed_bn):
    torch.manual_seed(seed_w)
    m = PointNetCls(num_points=num_points, input_chann=3, k=k)
    if seed_bn is not None:
        randomize_bn_(m.state_dict(), seed_bn)
    return m


def eval_case(tag, num_points, k, b, seed_w, seed_bn, seed_x, kind):
    m = build_ref(num_points, k, seed_w, seed_bn).eval()
    x = synth_cloud(b, num_points, seed_x, kind)
    cap = {}
    h1 = m.feat.stn.mp1.register_forward_hook(lambda mod, i, o: cap.__setitem__("stn_pool", o.detach().view(-1, 1024)))
    h2 = m.feat.mp1.register_forward_hook(lambda mod, i, o: cap.__setitem__("feat_pool", o.detach().view(-1, 1024)))
    with torch.no_grad():
        logp, trans = m(x)
    h1.remove(); h2.remove()
    names, cs = checksums(m.state_dict())
    np.savez_compressed(os.path.join(OUT, f"pointnet_eval_{tag}.npz"),
                        num_points=num_points, k=k, b=b, seed_w=seed_w,
                        seed_bn=-1 if seed_bn is None else seed_bn, seed_x=seed_x, kind=kind,
                        x=x.numpy(), logp=logp.numpy(), trans=trans.numpy(),
                        stn_pool=cap["stn_pool"].numpy(), feat_pool=cap["feat_pool"].numpy(),
                        names=names, checksums=cs)
    print("eval", tag, logp[0].tolist())


def sample_idx(numel, count=256):
    return np.unique(np.linspace(0, numel - 1, min(count, numel)).astype(np.int64))


def train_case(tag, num_points, k, b, seed_w, seed_bn, seed_x, kind):
    m = build_ref(num_points, k, seed_w, seed_bn).train()
    x = synth_cloud(b, num_points, seed_x, kind)
    y = (torch.arange(b) % k).long()
    logp, trans = m(x)
    loss = torch.nn.functional.nll_loss(logp, y)   # main_1v.py:74
    loss.backward()
    rec = dict(num_points=num_points, k=k, b=b, seed_w=seed_w,
               seed_bn=-1 if seed_bn is None else seed_bn, seed_x=seed_x, kind=kind,
               x=x.numpy(), y=y.numpy(), loss=loss.item(), logp=logp.detach().numpy(),
               trans=trans.detach().numpy())
    gnames, gnorm, gsum = [], [], []
    for name, p in m.named_parameters():
        g = p.grad.detach()
        gnames.append(name); gnorm.append(g.double().norm().item()); gsum.append(g.double().sum().item())
        flat = g.flatten().numpy()
        if flat.size <= 16384:
            rec["grad/" + name] = flat.reshape(g.shape)
        else:
            idx = sample_idx(flat.size, 2048)
            rec["gradidx/" + name] = idx
            rec["gradsample/" + name] = flat[idx]
    rec["grad_names"] = np.array(gnames); rec["grad_norm"] = np.array(gnorm); rec["grad_sum"] = np.array(gsum)
    for name, buf in m.named_buffers():
        if "running_" in name:
            rec["stat/" + name] = buf.detach().numpy()
    # re-create the *initial* state for the checksum (m was mutated: running stats moved)
    m0 = build_ref(num_points, k, seed_w, seed_bn)
    names, cs = checksums(m0.state_dict())
    rec["names"] = names; rec["checksums"] = cs
    np.savez_compressed(os.path.join(OUT, f"pointnet_train_{tag}.npz"), **rec)
    print("train", tag, loss.item())


def kat_cases():
    """SURVEY.md §8c survey KATs, regenerated (not trusted blindly)."""
    torch.manual_seed(0)
    m = PointNetCls(750, 3, 2).eval()
    x = torch.randn(64, 3, 750)
    with torch.no_grad():
        logp, trans = m(x)
    torch.manual_seed(1)
    m2 = PointNetCls(1024, 3, 3).train()
    x2 = torch.randn(16, 3, 1024)
    y2 = torch.arange(16) % 3
    lp2, _ = m2(x2)
    loss = torch.nn.functional.nll_loss(lp2, y2)
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "pointnet_kat.npz"),
                        a_logp0=logp[0].numpy(), a_logp_sum=logp.double().sum().item(), a_trans0=trans[0].numpy(),
                        a_logp=logp.numpy(),
                        b_loss=loss.item(),
                        b_gn_fc3=m2.fc3.weight.grad.double().norm().item(),
                        b_gn_conv3=m2.feat.conv3.weight.grad.double().norm().item(),
                        b_gn_stn_conv1=m2.feat.stn.conv1.weight.grad.double().norm().item(),
                        b_bn3_rm3=m2.feat.bn3.running_mean[:3].numpy())
    print("KAT-A", logp[0].tolist(), logp.sum().item())
    print("KAT-B", loss.item())


class _RefCrop(ref_dataset.BaseGraspDataset):
    def __init__(self, min_point_limit=50):
        super().__init__()
        self.min_point_limit = min_point_limit
        self.projection = False


def crop_cases():
    ds = _RefCrop()
    keys = sorted(ds.transform.keys())
    rng = np.random.default_rng(2024)
    recs = {}
    n_case = 12
    P = 4000
    grasps = np.zeros((n_case, 12)); pcs = np.zeros((n_case, P, 3)); Ts = np.zeros((n_case, 4, 4))
    counts = np.zeros(n_case, dtype=np.int64); is_none = np.zeros(n_case, dtype=bool)
    for c in range(n_case):
        T = np.asarray(ds.transform[keys[(c * 5) % len(keys)]][1], dtype=np.float64)
        axis = rng.normal(size=3)
        if c == 3:
            axis = np.array([0.0, 0.0, 1.0])      # degenerate axis_x branch, dataset.py:29
        if c == 4:
            axis = np.array([0.0, 0.0, -2.5])
        center = rng.uniform(-0.05, 0.05, size=3)
        width = 0.085 if c % 3 else rng.uniform(0.03, 0.085)
        angle = rng.uniform(-np.pi, np.pi)
        g = np.concatenate([center, axis, [width, angle, width, 0.0, rng.choice([0.4, 0.5, 0.8, 1.2, 2.0]),
                                            rng.uniform(0, 1)]])
        # cloud in the *cloud* frame: points around T @ center so that the box is populated
        ctr_t = (T @ np.r_[center, 1.0])[:3]
        spread = 0.06 if c != 7 else 0.5       # case 7: sparse -> fewer than 50 points -> None
        pc = ctr_t + rng.uniform(-spread, spread, size=(P, 3))
        out = ds.collect_pc(g, pc, T)
        grasps[c], pcs[c], Ts[c] = g, pc, T
        counts[c] = len(ds.in_ind)
        is_none[c] = out is None
        recs[f"ind_{c}"] = np.asarray(ds.in_ind, dtype=np.int64)
        # full transformed cloud, via the kept rows (reference returns only the kept rows)
        recs[f"pts_{c}"] = out if out is not None else np.zeros((0, 3))
    np.savez_compressed(os.path.join(OUT, "crop_train.npz"), grasps=grasps, pcs=pcs, Ts=Ts,
                        counts=counts, is_none=is_none, **recs)
    print("crop counts", counts.tolist(), "none", is_none.tolist())


def dataset_cases():
    """What the reference's four Dataset classes return on the synthetic miniature tree of
    tests/synth_dataset.py (indices, numpy seed and ctor kwargs recorded there)."""
    import tempfile
    from tests import synth_dataset
    rec = {}
    with tempfile.TemporaryDirectory() as root:
        synth_dataset.build(root)
        items = synth_dataset.replay(ref_dataset, root)
    none_mask = []
    for n, (name, i, item) in enumerate(items):
        none_mask.append(item is None)
        if item is None:
            continue
        rec[f"pc_{n}"] = np.asarray(item[0])
        rec[f"label_{n}"] = int(item[1])
        if len(item) > 2:
            rec[f"obj_{n}"] = str(item[2])
    rec["none_mask"] = np.array(none_mask)
    rec["names"] = np.array([it[0] for it in items]); rec["indices"] = np.array([it[1] for it in items])
    np.savez_compressed(os.path.join(OUT, "dataset_items.npz"), **rec)
    print("dataset items", len(items), "none:", int(np.sum(none_mask)))


if __name__ == "__main__":
    eval_case("n64_k2", 64, 2, 3, 11, 4321, 101, "box")
    eval_case("n750_k2", 750, 2, 2, 12, 4322, 102, "box")
    eval_case("n1024_k3", 1024, 3, 2, 13, 4323, 103, "gauss")
    eval_case("n100_k3_b1", 100, 3, 1, 14, 4324, 104, "randn")
    eval_case("n500_k3_defaultbn", 500, 3, 2, 15, None, 105, "box")
    train_case("n64_k2", 64, 2, 4, 21, 4331, 201, "randn")
    train_case("n200_k3", 200, 3, 16, 22, 4332, 202, "box")
    kat_cases()
    crop_cases()
    dataset_cases()   # last: it re-points PointNetGPD_FOLDER at a temporary tree
