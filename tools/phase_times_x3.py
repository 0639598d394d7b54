#!/usr/bin/env python3
"""Per-phase wave cycles of trunk_infer_x3_kernel from a -DPNGPD_TIMING variant library (kernel experiments only)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pointnetgpd_amd import _lib, ops
from pointnetgpd_amd.model import pointnet as pn
lib = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
dev = torch.device("cuda:0")
B = N = 1024
m = bench.build_model(N, 2, dev).eval(); x = bench.synth_clouds(B, N, 1, dev)
names = ["(loop)", "stage points", "barrier a", "layer 1", "barrier b", "layer 2 + h2", "barrier c", "layer 3 (4 blocks)"]
for nt in (1, 3):
    wts = pn._trunk_infer_weights_x3(m.feat.stn, dev)
    ops.trunk_fwd_infer_bf(x, None, *wts, relu_last=True, nterms=nt)
    lib.pngpd_tm_read_x3(buf, 1)
    for _ in range(3): ops.trunk_fwd_infer_bf(x, None, *wts, relu_last=True, nterms=nt)
    lib.pngpd_tm_read_x3(buf, 0)
    waves, tiles = buf[15], 8
    tot = sum(buf[i] for i in range(10))
    print(f"== trunk_infer_x3_kernel<{nt}>: {waves} waves, cycles per wave per 128-point tile (total {tot / waves / tiles:.0f})")
    for i, n in enumerate(names): print(f"  {n:22s} {buf[i] / waves / tiles:9.0f}  {100.0 * buf[i] / tot:5.1f}%")
