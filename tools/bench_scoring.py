#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU: G sampled grasp candidates -> in-gripper crop (50k-point scene) ->
resample to N=1024 -> 3-class PointNet scoring -> vote/sort.  Prints one JSON line with the stage split."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pointnetgpd_amd import crop
from pointnetgpd_amd.scoring import GraspScorer
from tests.test_gpu_crop_scoring import _scene

G = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
P, N, k = 50000, 1024, 3
dev = torch.device("cuda:0")
model = bench.build_model(N, k, dev)
pc, grasps = _scene(G, P, 78)
pc32 = pc.astype(np.float32)
scorer = GraspScorer(model, num_points=N, repeat=1, batch=4096, seed=1, max_keep=8192)
scorer.score(pc32, grasps[:8192]); torch.cuda.synchronize()
t0 = time.perf_counter(); res = scorer.score(pc32, grasps); torch.cuda.synchronize(); total = time.perf_counter() - t0
# stage split
cloud = torch.from_numpy(pc32).to(dev)
frames = torch.from_numpy(crop.frames_from_grasps_infer(grasps)).to(dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
counts, idx = crop.crop_count_compact(cloud, frames, 8192); torch.cuda.synchronize(); t_crop = time.perf_counter() - t0
t0 = time.perf_counter()
for s in range(0, G, 4096):
    crop.crop_resample(cloud, frames[s:s+4096], counts[s:s+4096], idx[s:s+4096], N, crop.MODE_INFER, 20, seed=s)
torch.cuda.synchronize(); t_res = time.perf_counter() - t0
# the same scene with the opt-in bf16x3 trunk
from pointnetgpd_amd.model import pointnet as pn
pn.set_inference_precision("bf16x3")
scorer.score(pc32, grasps[:8192]); torch.cuda.synchronize()
t0 = time.perf_counter(); res_f = scorer.score(pc32, grasps); torch.cuda.synchronize(); total_f = time.perf_counter() - t0
pn.set_inference_precision("fp32")
agree = float((res_f["pred"] == res["pred"]).float().mean())
print(json.dumps({"workload": f"config5: {G} candidates x {P}-point scene, N={N}, k={k}, 1 GPU", "grasps_per_s": round(G / total, 1),
                  "total_s": round(total, 4), "crop_count_compact_s": round(t_crop, 4), "resample_s": round(t_res, 4),
                  "valid_frac": round(float(res["valid"].float().mean()), 3), "good": int(res["good"].sum()),
                  "bf16x3": {"total_s": round(total_f, 4), "grasps_per_s": round(G / total_f, 1),
                             "pred_agreement_with_fp32": agree,
                             "max_abs_dscore": float((res_f["score"] - res["score"]).abs().max())}}))
