#!/usr/bin/env python3
"""fp32 / bf16x3 / bf16 training step (graph replay) at B = N = 1024 for ONE library (PNGPD_LIB selects it)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointnetgpd_amd import train
from pointnetgpd_amd.train import GraphedTrainStep
dev = torch.device("cuda:0")
B, N, k = 1024, 1024, 2
x = bench.synth_clouds(B, N, 1, dev); y = (torch.arange(B, device=dev) % k).long()
row = {"lib": os.path.basename(os.environ.get("PNGPD_LIB", "product"))}
for prec in ("fp32", "bf16x3", "bf16"):
    train.set_train_precision(prec)
    gs = GraphedTrainStep(bench.build_model(N, k, dev), B, N, lr=0.005)
    for _ in range(5): gs(x, y)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): gs(x, y)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
    row[prec + "_ms"] = round(sorted(ts)[2], 4)
    del gs
train.set_train_precision("fp32")
print(json.dumps(row), flush=True)
