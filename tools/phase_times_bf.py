#!/usr/bin/env python3
"""Per-phase wave-cycle accounting of the bf16-native passes D / E (pngpd_bwd_bf.h) from a -DPNGPD_TIMING build:
   PNGPD_LIB=build_probe/lib_tm.so python tools/phase_times_bf.py [nterms]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointnetgpd_amd import _lib, ops
lib = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
NAMES = {"D": ["z unpack (+ prefetch issue)", "census", "barrier 1", "compaction + h2 build + hT write", "barrier 2",
               "h2 A (tr reads + mfma)", "sparse term", "gram", "epilogue + g2 store"],
         "E": ["top", "layer 1 -> h1T", "dz build + dzT write", "barrier", "W2^T dz (tr reads + mfma)", "g1 sums (mfma)",
               "dW2", "-", "-"]}
for nt in ([int(a) for a in sys.argv[1:]] or [1, 3]):
    for which in ("D", "E"):
        orig = getattr(ops, "trunk_bwd_d_bf" if which == "D" else "trunk_bwd_e_bf")
        calls = []
        def spy(*a, **k):
            if not calls:
                lib.pngpd_tm_read(buf, 1)
            calls.append(1)
            return orig(*a, **k)
        setattr(ops, orig.__name__, spy)
        try:
            bench.train_pass_rooflines_bf(1024, 1024, torch.device("cuda:0"), nt, reps=3)
        finally:
            setattr(ops, orig.__name__, orig)
        # the spy reset the counters at the pass's first launch; later passes do not stamp this array's slots again
        lib.pngpd_tm_read(buf, 0)
        waves, tiles = buf[15], 16
        if not waves:
            print(f"== pass {which} nterms {nt}: no stamps (library not built with -DPNGPD_TIMING?)"); continue
        tot = sum(buf[i] for i in range(10))
        print(f"== pass {which} nterms {nt}: {waves} waves, cycles per wave and 64-point tile by phase (total {tot / waves / tiles:.0f})")
        for i, n in enumerate(NAMES[which]):
            print(f"  {n:36s} {buf[i] / waves / tiles:9.0f}  {100.0 * buf[i] / max(tot, 1):5.1f}%")
