#!/usr/bin/env python3
"""The reference's own recipe (main_1v.py: batch 64, N = 750) in the three arithmetic modes: eager step (forward_loss +
backward + FlatAdam) vs the same step replayed from a HIP graph (train.GraphedTrainStep) — VERDICT r5 weak #8 asked for the
graph figure of the reduced-precision modes.  One JSON line."""
import json, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pointnetgpd_amd import train as _train
from pointnetgpd_amd.optim import FlatAdam
dev = torch.device("cuda:0")
B, N = 64, 750
out = {"B": B, "N": N, "unit": "ms per step (median of 5 blocks of 50)"}


def timeit(fn, reps=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / reps)
    return round(statistics.median(ms), 4)


x = bench.synth_clouds(B, N, 5, dev)
y = (torch.arange(B, device=dev) % 2).long()
for prec in ("fp32", "bf16x3", "bf16"):
    m = bench.build_model(N, 2, dev).set_precision(prec).train()
    opt = FlatAdam(m.parameters(), lr=0.005)

    def step():
        opt.zero_grad()
        loss, _, _ = m.forward_loss(x, y)
        _train.loss_backward(loss)
        opt.step()
    eager = timeit(step)
    m2 = bench.build_model(N, 2, dev).set_precision(prec).train()
    g = _train.GraphedTrainStep(m2, B, N, lr=0.005)
    graph = timeit(lambda: g(x, y))
    out[prec] = {"eager_ms": eager, "hip_graph_ms": graph, "samples_per_s_graph": round(B / graph * 1e3, 1)}
print(json.dumps(out))
