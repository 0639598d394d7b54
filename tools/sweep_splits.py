#!/usr/bin/env python3
"""fp32 train step (graph replay) at N = 1024 for every workgroups-per-cloud choice S of the training passes, per batch
size — checks the library's cost model (pngpd_trunk_splits).  usage: python tools/sweep_splits.py [B ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from pointnetgpd_amd import ops, train
from pointnetgpd_amd.train import GraphedTrainStep

dev = torch.device("cuda:0")


def t_ms(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return round(sorted(out)[2], 4)


N, k = 1024, 2
for B in [int(a) for a in sys.argv[1:]] or [32, 64, 128, 256, 512]:
    x = bench.synth_clouds(B, N, 1, dev); y = (torch.arange(B, device=dev) % k).long()
    row = {"B": B, "auto_S": ops.train_splits(B, N)}
    for S in (1, 2, 4, 8, 16):
        ops.TRAIN_TARGET_BLOCKS = B * S
        train._SIZE_CACHE.clear()
        assert ops.train_splits(B, N) == S
        gs = GraphedTrainStep(bench.build_model(N, k, dev), B, N, lr=0.005)
        row[f"S{S}"] = t_ms(lambda: gs(x, y), 20 if B >= 512 else 50)
        del gs
    ops.TRAIN_TARGET_BLOCKS = 0
    print(json.dumps(row), flush=True)
