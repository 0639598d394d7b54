#!/usr/bin/env python3
"""Eval-mode forward in the two reduced-precision modes at B = N = 1024 and B = 512 (events on the stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointnetgpd_amd.model import pointnet as pn
dev = torch.device("cuda:0")
for B, N, k in [(1024, 1024, 2), (512, 1024, 3)]:
    m = bench.build_model(N, k, dev).eval()
    x = bench.synth_clouds(B, N, 1, dev)
    for prec in ("bf16x3", "bf16"):
        pn.set_inference_precision(prec)
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
        pn.set_inference_precision("fp32")
        print(f"B={B} N={N} {prec}: {best:.4f} ms {B / best * 1e3:.0f} grasps/s")
