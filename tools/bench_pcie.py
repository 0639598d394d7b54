#!/usr/bin/env python3
"""The PCIe-inclusive rate of the headline workload (B = N = 1024 fp32 clouds handed over in pinned HOST memory instead of
HBM): the boundary itself takes device pointers, so this is a caller-side figure — DESIGN.md §6's PCIe note, measured.

  serial     : copy_(non_blocking) then forward on one stream
  overlapped : batch t+1's copy on a side stream under batch t's forward (two device buffers, one event per batch)

One JSON line.  `resident` is bench.py's `value` (inputs already in HBM)."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
B = N = 1024
model = bench.build_model(N, 2, dev)
host = [bench.synth_clouds(B, N, 5 + i, dev).cpu().pin_memory() for i in range(4)]
dbuf = [torch.empty(B, 3, N, device=dev) for _ in range(2)]
steps = 60


def timed(fn):
    for _ in range(3):
        fn(6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def resident(n):
    with torch.no_grad():
        for _ in range(n):
            model(dbuf[0])


def serial(n):
    with torch.no_grad():
        for i in range(n):
            dbuf[0].copy_(host[i % 4], non_blocking=True)
            model(dbuf[0])


side = torch.cuda.Stream(dev, priority=-1)        # the copy must not queue behind a forward that fills every CU


def overlapped(n):
    main = torch.cuda.current_stream(dev)
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    freed = [torch.cuda.Event(), torch.cuda.Event()]
    for e in freed:
        e.record(main)
    with torch.no_grad():
        with torch.cuda.stream(side):
            side.wait_event(freed[0])
            dbuf[0].copy_(host[0], non_blocking=True)
            ready[0].record(side)
        for i in range(n):
            cur, nxt = i & 1, (i + 1) & 1
            if i + 1 < n:
                with torch.cuda.stream(side):
                    side.wait_event(freed[nxt])
                    dbuf[nxt].copy_(host[(i + 1) % 4], non_blocking=True)
                    ready[nxt].record(side)
            main.wait_event(ready[cur])
            model(dbuf[cur])
            freed[cur].record(main)


copy_ms = timed(lambda n: [dbuf[0].copy_(host[i % 4], non_blocking=True) for i in range(n)])
r, s, o = timed(resident), timed(serial), timed(overlapped)
print(json.dumps({"workload": "eval forward B=1024 N=1024 fp32, clouds in pinned host memory (12.6 MB per batch)",
                  "h2d_copy_ms": round(copy_ms, 4), "h2d_GBps": round(B * 3 * N * 4 / copy_ms / 1e6, 1),
                  "resident_ms": round(r, 4), "resident_grasps_s": round(B / r * 1e3, 1),
                  "serial_copy_then_forward_ms": round(s, 4), "serial_grasps_s": round(B / s * 1e3, 1),
                  "overlapped_ms": round(o, 4), "overlapped_grasps_s": round(B / o * 1e3, 1)}))
