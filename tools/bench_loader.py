#!/usr/bin/env python3
"""Training-data feed rate: HBM-resident DeviceGraspLoader vs the numpy Dataset mirror (one host process, i.e. what
ONE of the reference's 32 DataLoader workers delivers), on a synthetic on-disk tree with YCB-like sizes."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import synth_dataset
from pointnetgpd_amd.device_loader import DeviceGraspLoader
from pointnetgpd_amd.model import dataset as ds_mod

dev = torch.device("cuda:0")
root = synth_dataset.build(tempfile.mkdtemp(), grasps_per_obj=6500, points=20000)
os.environ["PointNetGPD_FOLDER"] = root
cases = [("one-view N=750 B=64", ds_mod.PointGraspOneViewDataset(grasp_points_num=750, grasp_amount_per_file=6500, thresh_good=0.6, thresh_bad=0.6, tag="train"), 64, 8192),
         ("full-view 50000-pt sample clouds, N=1000 B=64", ds_mod.PointGraspDataset(obj_points_num=50000, grasp_points_num=1000, pc_file_used_num=6, grasp_amount_per_file=6500, thresh_good=0.6, thresh_bad=0.6, tag="train"), 64, 16384)]
for name, ds, B, mk in cases:
    loader = DeviceGraspLoader(ds, B, dev, seed=1, max_keep=mk)
    it = iter(loader)
    for _ in range(3): next(it)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for _ in range(30):
        d, t = next(it); n += d.shape[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    np.random.seed(0)
    t0 = time.perf_counter(); m = 0
    for i in np.random.default_rng(0).integers(0, len(ds), 40):
        m += ds[int(i)] is not None
    host = (time.perf_counter() - t0) / 40
    print(json.dumps({"case": name, "device_loader_samples_per_s": round(n / dt, 1), "ms_per_batch": round(dt / 30 * 1e3, 3),
                      "host_numpy_ms_per_sample_one_process": round(host * 1e3, 3),
                      "host_samples_per_s_one_process": round(1 / host, 1), "kept_frac": round(n / (30 * B), 3)}))
