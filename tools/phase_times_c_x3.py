#!/usr/bin/env python3
"""Per-phase wave cycles of pass C of the bf16 modes (trunk_fwd_train_x3_kernel<NT, LOADZ>) from a -DPNGPD_TIMING build:
   PNGPD_LIB=build_probe/lib_tm.so python tools/phase_times_c_x3.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pointnetgpd_amd import _lib, ops
lib = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
names = ["loop top", "barrier (previous tile's readers)", "z unpack + h2 build + tile write", "barrier (tile complete)",
         "4 x weight fetch (exposed: vmcnt(0) in this build)", "4 x 96/32 matrix instructions + A reads", "4 x epilogue"]
for nt in (1, 3):
    orig = ops.trunk_fwd_train_bf
    state = {"n": 0}
    def spy(*a, **k):
        if state["n"] == 0:
            lib.pngpd_tm_read_x3(buf, 1)
        state["n"] += 1
        return orig(*a, **k)
    ops.trunk_fwd_train_bf = spy
    try:
        bench.train_pass_rooflines_bf(1024, 1024, torch.device("cuda:0"), nt, reps=3)
    finally:
        ops.trunk_fwd_train_bf = orig
    lib.pngpd_tm_read_x3(buf, 0)
    waves, tiles = buf[15], 8
    tot = sum(buf[i] for i in range(10))
    print(f"== trunk_fwd_train_x3_kernel<{nt}, true>: {waves} waves, cycles per wave and 128-point tile (total {tot / max(waves,1) / tiles:.0f})")
    for i, n in enumerate(names):
        print(f"  {n:52s} {buf[i] / max(waves,1) / tiles:9.0f}  {100.0 * buf[i] / max(tot, 1):5.1f}%")
