#!/usr/bin/env python3
"""Eval-mode forward at a few (B, N) shapes: ms, grasps/s and fraction of the fp32-MFMA peak (events on the stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
shapes = [(64, 750, 2), (64, 1024, 2), (128, 750, 2), (40, 500, 3), (8, 500, 3), (1, 500, 3), (256, 1024, 2), (512, 1024, 3), (1024, 1024, 2)]
for B, N, k in shapes:
    m = bench.build_model(N, k, dev).eval()
    x = bench.synth_clouds(B, N, 1, dev)
    with torch.no_grad():
        for _ in range(5): m(x)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): m(x)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
    fl = bench.flops_per_grasp(N, k) * B
    print(f"B={B:5d} N={N:5d} k={k}: {best:.3f} ms  {B / best * 1e3:10.0f} grasps/s  {fl / best / 1e9:6.1f} TFLOP/s = {fl / best / 1e9 / 157.3 * 100:.1f}% of peak")
