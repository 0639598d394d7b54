#!/usr/bin/env python3
"""Timings of the FC-stack kernels alone and of the whole fp32 train step for ONE library (PNGPD_LIB selects it);
tools/ab_fc.sh alternates two libraries across processes.  usage: python tools/ab_fc.py [B ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import bench
from pointnetgpd_amd import ops
from pointnetgpd_amd.optim import FlatAdam
from pointnetgpd_amd.train import GraphedTrainStep

dev = torch.device("cuda:0")


def t_us(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / reps)
    return round(sorted(best)[2], 2)


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [64, 128, 1024]
    N, k = 1024, 2
    for B in Bs:
        row = {"lib": os.path.basename(os.environ.get("PNGPD_LIB", "product")), "B": B}
        g = torch.Generator().manual_seed(B)
        for name, (K, Nout) in {"fc1": (1024, 512), "fc2": (512, 256), "fc3": (256, k)}.items():
            x = torch.randn(B, K, generator=g).to(dev); W = torch.randn(Nout, K, generator=g).to(dev)
            b = torch.randn(Nout, generator=g).to(dev); gg = torch.randn(B, Nout, generator=g).to(dev)
            # back-to-back launches of the same kernel on one stream: per-launch time incl. the boundary
            row[name + "_fwd_us"] = t_us(lambda: ops.fc_fwd(x, W, b, ops.EPI_NONE))
            row[name + "_bwd_us"] = t_us(lambda: ops.fc_bwd(gg, x, W))
        m = bench.build_model(N, k, dev).train()
        opt = FlatAdam(m.parameters(), lr=0.005)
        x = bench.synth_clouds(B, N, 1, dev); y = (torch.arange(B, device=dev) % k).long()

        def plain():
            opt.zero_grad(); F.nll_loss(m(x)[0], y).backward(); opt.step()
        row["step_eager_ms"] = round(t_us(plain, 20 if B >= 512 else 60) / 1e3, 4)
        gs = GraphedTrainStep(bench.build_model(N, k, dev), B, N, lr=0.005)
        row["step_graph_ms"] = round(t_us(lambda: gs(x, y), 20 if B >= 512 else 60) / 1e3, 4)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
