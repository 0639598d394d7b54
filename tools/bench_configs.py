#!/usr/bin/env python3
"""Every BASELINE.json config shape on ONE MI355X (per-GPU share where the config names 8 GPUs):
eval forward and full training step (fwd + nll_loss + bwd + Adam), events on the current stream."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import bench
from pointnetgpd_amd import train as pt
from pointnetgpd_amd.optim import FlatAdam
from pointnetgpd_amd.model import pointnet as pn

dev = torch.device("cuda:0")
CONFIGS = [("C1 shape on GPU: B=64 N=750 k=2", 64, 750, 2),
           ("C2: B=1024 N=1024 k=2", 1024, 1024, 2),
           ("C3 per-GPU share: B=512 N=1024 k=3", 512, 1024, 3),
           ("C4: B=512 N=4096 k=2", 512, 4096, 2)]


def timeit(fn, reps, blocks=5):
    """Median over ``blocks`` event-timed blocks of ``reps`` calls: a one-off host hiccup early in the process's life
    (tens of ms, seen once per run at a random leg) lands in one block and does not reach the reported figure."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(blocks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return sorted(out)[len(out) // 2]


def main():
    rows = []
    for name, B, N, k in CONFIGS:
        torch.manual_seed(0)
        m = pn.PointNetCls(N, 3, k).to(dev)
        x = bench.synth_clouds(B, N, 1, dev)
        y = torch.randint(0, k, (B,), device=dev)
        flops = B * (N * 557842 + 2627072)
        row = dict(config=name)
        xb = x.to(torch.bfloat16)
        for prec in ("fp32", "bf16x3", "bf16", "bf16_storage"):
            pn.set_inference_precision("bf16" if prec == "bf16_storage" else prec)
            m.eval()
            xin = xb if prec == "bf16_storage" else x
            with torch.no_grad():
                ms = timeit(lambda: m(xin), 20)
            row[f"infer_{prec}_ms"] = round(ms, 3)
            row[f"infer_{prec}_grasps_s"] = round(B / ms * 1e3)
            if prec == "fp32":
                row["infer_fp32_tflops"] = round(flops / ms / 1e9, 1)
        pn.set_inference_precision("fp32")
        for prec in ("fp32", "bf16x3", "bf16"):
            pt.set_train_precision(prec)
            m.train()
            opt = FlatAdam(m.parameters(), lr=0.005)        # the CLI's optimizer on --cuda (mains.py)

            def step():
                opt.zero_grad()
                loss, _, _ = m.forward_loss(x, y)          # mains.py's step: loss inside the head's calls
                pt.loss_backward(loss)
                opt.step()
            ms = timeit(step, 8)
            row[f"train_{prec}_ms"] = round(ms, 3)
            row[f"train_{prec}_grasps_s"] = round(B / ms * 1e3)
        pt.set_train_precision("fp32")
        torch.manual_seed(0)
        mg = pn.PointNetCls(N, 3, k).to(dev)
        gstep = pt.GraphedTrainStep(mg, B, N, lr=0.005)
        ms = timeit(lambda: gstep(x, y), 8)
        row["train_fp32_hipgraph_ms"] = round(ms, 3)
        row["train_fp32_hipgraph_grasps_s"] = round(B / ms * 1e3)
        del gstep, mg
        rows.append(row)
        print(json.dumps(row))
    return rows


if __name__ == "__main__":
    main()
