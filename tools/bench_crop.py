#!/usr/bin/env python3
"""The inference crop of BASELINE configs[4] alone (100,000 hands x one 50,000-point scene -> (G,3,1024) clouds):
whole-cloud count + resample (two kernels, index lists through HBM) against the one-launch crop over the scene's spatial
index (lists in LDS).  HIP events on the stream."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pointnetgpd_amd import crop
from pointnetgpd_amd.gpg import CloudIndex
dev = torch.device("cuda:0")
G, P, N, MK = 100000, 50000, 1024, 8192
pc, grasps = bench.synth_scene(G, P)
cloud = torch.from_numpy(pc).to(dev)
frames = torch.from_numpy(crop.frames_from_grasps_infer(grasps)).to(dev)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def plain():
    counts, idx = crop.crop_count_compact(cloud, frames, MK)
    for s in range(0, G, 1024):
        crop.crop_resample(cloud, frames[s:s + 1024], counts[s:s + 1024], idx[s:s + 1024], N, crop.MODE_INFER, 20, seed=1, g_base=s)
index = CloudIndex(cloud)
def indexed_two():
    counts, idx = crop.crop_count_compact_indexed(index, frames, MK)
    for s in range(0, G, 1024):
        crop.crop_resample(index.cloud, frames[s:s + 1024], counts[s:s + 1024], idx[s:s + 1024], N, crop.MODE_INFER, 20, seed=1, g_base=s)
def fused():
    for s in range(0, G, 1024):
        crop.crop_indexed(index, frames[s:s + 1024], N, crop.MODE_INFER, 20, seed=1, g_base=s, max_keep=MK)
out = {"workload": f"{G} hands x {P}-point scene -> N={N}", "plain_count_plus_resample_ms": round(timed(plain), 2),
       "indexed_count_plus_resample_ms": round(timed(indexed_two), 2), "indexed_one_launch_ms": round(timed(fused), 2),
       "index_build_ms": round(timed(lambda: CloudIndex(cloud), 5), 3)}
c, _ = crop.crop_count_compact(cloud, frames[:2048], MK)
out["mean_in_box_points"] = round(float(c.float().mean()), 1)
print(json.dumps(out))
