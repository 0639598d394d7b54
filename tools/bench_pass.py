#!/usr/bin/env python3
"""Time the individual training passes (events on the launch stream) on synthetic but valid operands."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointnetgpd_amd import ops
import bench
dev = torch.device("cuda:0")
B, N = int(os.environ.get("PNGPD_BENCH_B", "1024")), 1024
g = torch.Generator(device="cpu").manual_seed(0)
x = bench.synth_clouds(B, N, 1, dev)
T = (torch.eye(3)[None] + 0.1 * torch.randn(B, 3, 3, generator=g)).to(dev).contiguous()
r = lambda *s: torch.randn(*s, generator=g).to(dev)
w1, b1 = r(64, 3), r(64) * 0.1
s1c, t1c = (torch.rand(64, generator=g) + 0.5).to(dev) * 30, r(64) * 0.1
w2 = r(128, 64) / 8; w3 = r(1024, 128) / 11
w2p = ops.pack_mfma_b(w2); w3p = ops.pack_mfma_b(w3); w2tp = ops.pack_mfma_b(w2.t().contiguous())
s2c, t2c = (torch.rand(128, generator=g) + 0.5).to(dev), r(128) * 0.1
is2, nm2, is1, nm1 = s2c.clone(), t2c.clone(), s1c.clone(), t1c.clone()
idx = torch.randint(0, N, (B, 1024), generator=g, dtype=torch.int32).to(dev)
coef = r(B, 1024) * 1e-3
A = r(128, 128) * 1e-3; Ap = ops.pack_mfma_b(((A + A.t()) / 2).contiguous()); cvec = r(128) * 1e-3
ev = r(3, 128)

ONLY = [t for t in os.environ.get("PNGPD_PASSES", "").split(",") if t]


def timeit(name, fn, reps=10 if B >= 512 else 60):
    if ONLY and not any(name.startswith(t) for t in ONLY):
        return fn()
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:14s} {e0.elapsed_time(e1) / reps:8.3f} ms")
    return out

S = int(os.environ.get("PNGPD_TRAIN_SPLITS", "0")) or ops.train_splits(B, N)
print(f"B {B} N {N} S {S}")
timeit("infer", lambda: ops.trunk_fwd_infer(x, T, w1, b1, w2p, t2c, w3p, r(1024), relu_last=False))
_, z2t = timeit("bn2_stats", lambda: ops.trunk_bn2_stats(x, T, w1, b1, s1c, t1c, w2p, S))
if os.environ.get("PNGPD_RECOMPUTE") == "1":
    z2t = None          # A/B: passes C / D / E recompute layers 1-2 instead of reading pass B's z2
timeit("fwd_train", lambda: ops.trunk_fwd_train(x, T, w1, b1, s1c, t1c, w2p, s2c, t2c, w3p, S, z2t))
timeit("gather", lambda: ops.trunk_bwd_gather(x, T, w1, b1, s1c, t1c, w2p, s2c, t2c, idx, coef))
g2t, pa, ps2 = timeit("bwd_d", lambda: ops.trunk_bwd_d(x, T, w1, b1, s1c, t1c, w2p, s2c, t2c, is2, nm2, Ap, cvec, w3, idx, coef, S, z2t))
pc, pR, pW2 = timeit("bwd_e", lambda: ops.trunk_bwd_e(x, T, w1, b1, s1c, t1c, w2p, is1, nm1, is2, nm2, ev[0], ev[1], ev[2], w2tp, g2t, S, z2t))
Gp = ops.trunk_bwd_gather(x, T, w1, b1, s1c, t1c, w2p, s2c, t2c, idx, coef)
timeit("reduce_D(3seg)", lambda: ops.reduce4((Gp, 1, Gp.shape[0], 1024 * 128), (pa, 1, B * S, 256), (ps2, 1, B * S, 12 * 1024)))
timeit("reduce_E(3seg)", lambda: ops.reduce4((pW2, 1, B * S, 128 * 64), (pc, 1, B * S, 128), (pR, B, S, 192)))
