#!/usr/bin/env python3
"""In-box point counts of the config-5 SAMPLED candidates (can scene) and the time of the two crop kernels per batch of
1024 hands, by count class: which path of crop_resample_kernel the bench's 680 us average comes from."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from pointnetgpd_amd import gpg, crop
dev = torch.device("cuda:0")
pts, nrm = bench.synth_can_scene(50000)
pfs = pts[pts[:, 2] > 0.010]
draws = np.random.default_rng(5).integers(0, len(pfs), 20000)
cloud = torch.from_numpy(pts).to(dev)
s = gpg.GpgGraspSamplerPcl(device=dev)
grasps = s.sample_grasps(cloud, pfs, nrm, 10 ** 9, len(draws), sample_indices=draws, as_array=True)
index = gpg.CloudIndex(cloud)
frames = torch.from_numpy(crop.frames_from_grasps_infer(grasps, crop.ROBOTIQ_85)).to(dev)
G = frames.shape[0]
out = {"candidates": int(G)}
for max_keep in (8192, 16384):
    counts, idx = crop.crop_count_compact_indexed(index, frames[:1024], max_keep)
    c = counts.cpu().numpy()
    out[f"max_keep {max_keep}"] = {"count_quantiles_0_10_50_90_100": [int(np.quantile(c, q)) for q in (0, .1, .5, .9, 1)],
                                   "frac_over_max_keep": float((c > max_keep).mean()), "frac_ge_1024": float((c >= 1024).mean())}
    def t(fn, reps=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    out[f"max_keep {max_keep}"]["count_compact_ms"] = round(t(lambda: crop.crop_count_compact_indexed(index, frames[:1024], max_keep)), 4)
    out[f"max_keep {max_keep}"]["resample_ms"] = round(t(lambda: crop.crop_resample(index.cloud, frames[:1024], counts, idx, 1024, crop.MODE_INFER, 20, seed=1)), 4)
    # by count class: hands sorted by count, timed in groups of 256
    order = np.argsort(c)
    cls = []
    for q in range(4):
        sel = torch.from_numpy(order[q * 256:(q + 1) * 256].copy()).to(dev)
        f2 = frames[:1024][sel].contiguous()
        c2, i2 = crop.crop_count_compact_indexed(index, f2, max_keep)
        cls.append({"counts": [int(c[order[q * 256]]), int(c[order[(q + 1) * 256 - 1]])],
                    "resample_ms_256_hands": round(t(lambda: crop.crop_resample(index.cloud, f2, c2, i2, 1024, crop.MODE_INFER, 20, seed=1)), 4)})
    out[f"max_keep {max_keep}"]["by_count_quartile"] = cls
print(json.dumps(out))
