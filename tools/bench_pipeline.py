#!/usr/bin/env python3
"""Scene -> GPG sampler -> in-gripper crop -> PointNet scoring, end to end on one GPU (scoring.detect_grasps),
at the robot's settings (kinect2grasp.py:42-47: 40 grasps / 150 sample points) and at candidate-generation scale."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from oracle import gpg_oracle as go          # synthetic scene generator only
from pointnetgpd_amd.scoring import GraspScorer, detect_grasps

dev = torch.device("cuda:0")
N, k = 500, 3                                 # model_type "3class" (kinect2grasp.py:58-65)
model = bench.build_model(N, k, dev)
scorer = GraspScorer(model, num_points=N, repeat=1, batch=4096, seed=1, max_keep=8192)
for P, num_grasps, max_samples in [(3000, 40, 150), (20000, 40, 150), (50000, 10 ** 9, 20000)]:
    pts, nrm = go.synth_scene("cylinder", P, 41)
    pts32 = pts.astype(np.float32)
    detect_grasps(pts32, nrm, scorer, num_grasps=num_grasps, max_num_samples=max_samples, seed=0)   # same sizes: pools warm
    torch.cuda.synchronize()
    reps = 5 if max_samples <= 150 else 3
    t0 = time.perf_counter()
    for r in range(reps):
        res = detect_grasps(pts32, nrm, scorer, num_grasps=num_grasps, max_num_samples=max_samples, seed=r)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"P": P, "num_grasps": num_grasps, "max_num_samples": max_samples, "candidates": int(len(res["grasps"])),
                      "good": int(res["good"].sum()), "seconds_per_scene": round(dt, 4),
                      "candidates_per_s": round(len(res["grasps"]) / dt, 1)}))
