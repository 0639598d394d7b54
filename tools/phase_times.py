#!/usr/bin/env python3
"""Per-phase wave-cycle accounting of passes D / E from a -DPNGPD_TIMING variant library (kernel experiments only):
   PNGPD_LIB=pointnetgpd_amd/csrc/build/variants/lib_tm.so python tools/phase_times.py"""
import ctypes, os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PNGPD_PASSES"] = "none"
import torch
from pointnetgpd_amd import _lib, ops
lib = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_pass.py"))
names = {"fwd_train": ["(loop)", "barrier 1", "h2 tile build", "barrier 2", "8 x 128 mfma (issue)", "8 x epilogue"],
         "bwd_d": ["loads+census", "barrier1", "compact+h2 write", "barrier2", "A.h2 (128 mfma)", "sparse", "gram (80 mfma)", "epilogue+store"],
         "bwd_e": ["next tile's point loads issued", "barrier A", "layer1", "dz tile (first use of the prefetched hand-off)", "barrier B",
                   "W2^T dz (64 mfma) + prefetch issue", "g1 epilogue", "dW2 (64 mfma)"]}
x, T, w1, b1, s1c, t1c, w2p, s2c, t2c, is2, nm2, Ap, cvec, w3, idx, coef, S, z2t, g2t = [g[k] for k in
    "x T w1 b1 s1c t1c w2p s2c t2c is2 nm2 Ap cvec w3 idx coef S z2t g2t".split()]
is1, nm1, ev, w2tp = g["is1"], g["nm1"], g["ev"], g["w2tp"]
def run(which):
    lib.pngpd_tm_read(buf, 1)
    for _ in range(3):
        if which == "fwd_train":
            ops.trunk_fwd_train(x, T, w1, b1, s1c, t1c, w2p, s2c, t2c, g["w3p"], S, z2t)
        elif which == "bwd_d":
            ops.trunk_bwd_d(x, T, w1, b1, s1c, t1c, w2p, s2c, t2c, is2, nm2, Ap, cvec, w3, idx, coef, S, z2t)
        else:
            ops.trunk_bwd_e(x, T, w1, b1, s1c, t1c, w2p, is1, nm1, is2, nm2, ev[0], ev[1], ev[2], w2tp, g2t, S, z2t)
    lib.pngpd_tm_read(buf, 0)
    waves = buf[15]; tiles = 16
    tot = sum(buf[i] for i in range(10))
    print(f"== {which}: {waves} waves, avg cycles per wave per tile by phase (total {tot / waves / tiles:.0f})")
    for i, n in enumerate(names[which]):
        print(f"  {n:22s} {buf[i] / waves / tiles:9.0f}  {100.0 * buf[i] / tot:5.1f}%")
run("fwd_train"); run("bwd_d"); run("bwd_e")
