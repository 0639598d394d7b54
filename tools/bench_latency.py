#!/usr/bin/env python3
"""Small-batch scoring latency (the robot loop of kinect2grasp.py scores <= 40 grasps per scene, B=1 each):
median wall time of one eval forward at B in {1, 8, 40}, N=500, k=3 — eager launches vs one HIP-graph replay."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
dev = torch.device("cuda:0")
N, k = 500, 3
model = bench.build_model(N, k, dev)
out = {}
for B in (1, 8, 40):
    x = bench.synth_clouds(B, N, 5, dev)
    with torch.no_grad():
        for _ in range(5): model(x)
        torch.cuda.synchronize()
        ts = []
        for _ in range(200):
            t0 = time.perf_counter(); lp, _ = model(x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        eager = float(np.median(ts)) * 1e6
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            lpg, _ = model(x)
        g.replay(); torch.cuda.synchronize()
        assert torch.equal(lp, lpg)
        ts = []
        for _ in range(200):
            t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        graph = float(np.median(ts)) * 1e6
    out[f"B={B}"] = {"eager_us": round(eager, 1), "graph_us": round(graph, 1)}
print(json.dumps({"workload": f"eval forward N={N} k={k}", **out}))
