#!/usr/bin/env python3
"""A/B of the FC stacks as one launch per direction (pngpd_head_train_t.tickets) vs launch-per-op, same process,
alternating blocks: fp32 train step (fwd + nll_loss + bwd + FlatAdam) at N = 1024, eager and as a replayed graph.
usage: python tools/ab_head_tickets.py [B ...]"""
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


def block(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    N, k = 1024, 2
    Bs = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024]
    for B in Bs:
        reps = 20 if B >= 512 else 60
        m = bench.build_model(N, k, dev).train()
        opt = FlatAdam(m.parameters(), lr=0.005)
        x = bench.synth_clouds(B, N, 1, dev); y = (torch.arange(B, device=dev) % k).long()

        def plain():
            opt.zero_grad(); F.nll_loss(m(x)[0], y).backward(); opt.step()
        graphs = {}
        for tk in (True, False):
            ops.HEAD_TICKETS = tk
            graphs[tk] = GraphedTrainStep(bench.build_model(N, k, dev), B, N, lr=0.005)
        t = {(tk, kind): [] for tk in (True, False) for kind in ("eager", "graph")}
        for rnd in range(6):
            for tk in (True, False):
                ops.HEAD_TICKETS = tk
                for _ in range(3):
                    plain()
                t[(tk, "eager")].append(block(plain, reps))
                g = graphs[tk]
                g(x, y)
                t[(tk, "graph")].append(block(lambda: g(x, y), reps))
        ops.HEAD_TICKETS = True
        med = lambda v: round(sorted(v)[len(v) // 2], 4)
        print(json.dumps({"B": B, "N": N,
                          "eager_one_launch_ms": med(t[(True, "eager")]), "eager_per_op_ms": med(t[(False, "eager")]),
                          "graph_one_launch_ms": med(t[(True, "graph")]), "graph_per_op_ms": med(t[(False, "graph")])}),
              flush=True)


if __name__ == "__main__":
    main()
