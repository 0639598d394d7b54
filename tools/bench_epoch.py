#!/usr/bin/env python3
"""End-to-end training throughput: the HBM-resident data layer (device_loader.DeviceGraspLoader, batches prefetched on
a side stream) feeding the HIP training step — what ``main_1v.py --cuda --device-data`` delivers per second, next to the
step alone on a resident batch and the loader alone.  Reference: main_1v.py:59-84 (the step) fed by :120-128 (32
DataLoader workers running dataset.py:420-458 / :244-282 per sample).

    python tools/bench_epoch.py [--steps 150] [--cases one,full,big]

One JSON line per case.  ``measure()`` is also what bench.py's ``train.epoch`` block calls."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {
    # name: (dataset class, ctor kwargs, batch, max_keep, points per view file, label)
    "one": ("PointGraspOneViewDataset", dict(grasp_points_num=750), 64, 8192, 20000,
            "one-view, B=64, N=750 (main_1v.py's own recipe)"),
    "full": ("PointGraspDataset", dict(grasp_points_num=1000, obj_points_num=50000, pc_file_used_num=6), 64, 16384, 20000,
             "full-view, B=64, N=1000, 50000-point sample clouds over 6 views (main_fullv.py's recipe)"),
    "big": ("PointGraspOneViewDataset", dict(grasp_points_num=1024), 1024, 8192, 20000,
            "one-view, B=1024, N=1024 (BASELINE configs[1] shape)"),
}


def _tree(points):
    from tests import synth_dataset
    root = synth_dataset.build(tempfile.mkdtemp(prefix="pngpd_epoch_"), grasps_per_obj=6500, points=points)
    os.environ["PointNetGPD_FOLDER"] = root
    return root


def measure(case, steps=150, warmup=10, dev=None, prefetch=2, root=None):
    import torch
    import torch.nn.functional as F
    from pointnetgpd_amd.device_loader import DeviceGraspLoader
    from pointnetgpd_amd.model import dataset as ds_mod
    from pointnetgpd_amd.model.pointnet import PointNetCls
    from pointnetgpd_amd.optim import FlatAdam
    from pointnetgpd_amd.train import loss_backward
    dev = dev or torch.device("cuda:0")
    cls, kw, B, max_keep, points, label = CASES[case]
    if root is None:
        _tree(points)
    else:
        os.environ["PointNetGPD_FOLDER"] = root
    ds = getattr(ds_mod, cls)(grasp_amount_per_file=6500, thresh_good=0.6, thresh_bad=0.6, tag="train", **kw)
    N = kw["grasp_points_num"]
    torch.manual_seed(0)
    model = PointNetCls(num_points=N, input_chann=3, k=2).to(dev).train()
    opt = FlatAdam(model.parameters(), lr=0.005)

    def step(x, y):
        opt.zero_grad()
        loss, _, _ = model.forward_loss(x, y)                  # mains.py's step (main_1v.py:72-76)
        loss_backward(loss)
        opt.step()

    def batches(loader):
        ep = 0
        while True:
            loader.set_epoch(ep)
            for b in loader:
                yield b
            ep += 1

    def timed(fn, it, n):
        for _ in range(2):
            fn(*next(it))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kept = 0
        for _ in range(n):
            x, y = next(it)
            kept += x.shape[0]
            fn(x, y)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0), kept

    out = {"case": label, "batch": B, "num_points": N, "steps": steps, "prefetch": prefetch}
    loader = DeviceGraspLoader(ds, B, dev, seed=1, max_keep=max_keep, prefetch=prefetch)
    serial = DeviceGraspLoader(ds, B, dev, seed=1, max_keep=max_keep, prefetch=0)
    x0, y0 = next(iter(loader))
    x0, y0 = x0.clone(), y0.clone()
    out["kept_frac"] = round(x0.shape[0] / B, 3)

    def fixed():
        while True:
            yield x0, y0
    # The eager step at B = 64 is host-bound, so these numbers move with the host's clocks and whatever ran before: the
    # four legs are timed in ALTERNATING blocks (each leg `blocks` times) and the medians are reported.
    legs = {"step_only": (step, fixed()), "loader_only": (lambda x, y: None, batches(loader)),
            "end_to_end": (step, batches(loader)), "serial_loader_end_to_end": (step, batches(serial))}
    blocks = 5
    per = max(10, steps // blocks)
    res = {k: [] for k in legs}
    for k, (fn, it) in legs.items():                        # warm every leg once
        timed(fn, it, warmup)
    for _ in range(blocks):
        for k, (fn, it) in legs.items():
            t, kept = timed(fn, it, per)
            res[k].append((t / per, kept / t))
    import statistics
    for k, v in res.items():
        out[k + "_ms"] = round(statistics.median(a for a, _ in v) * 1e3, 4)
        out[k + "_samples_s"] = round(statistics.median(b_ for _, b_ in v), 1)
    out["blocks"] = blocks
    out["end_to_end_over_step_only"] = round(out["end_to_end_samples_s"] / out["step_only_samples_s"], 4)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--cases", type=str, default="one,full,big")
    ap.add_argument("--prefetch", type=int, default=2)
    a = ap.parse_args()
    root = _tree(20000)
    for c in a.cases.split(","):
        print(json.dumps(measure(c, steps=a.steps if c != "big" else max(20, a.steps // 4), prefetch=a.prefetch, root=root)),
              flush=True)
