#!/usr/bin/env python3
"""Per-pass timing of the reduced-precision training passes (bench.train_pass_rooflines_bf) with output checksums:
`python tools/bench_pass_bf.py [B] [N]` prints one JSON line per mode.  Two library builds give identical checksums
iff their pass outputs are bit-identical in sum and absolute sum (the A/B gate of bit-preserving kernel edits)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
for nt in (1, 3):
    r = bench.train_pass_rooflines_bf(B, N, dev, nt, checksums=True)
    print(json.dumps({"nterms": nt, "B": B, "N": N, "ms": {k: v["avg_ms"] for k, v in r["passes"].items()},
                      "hbm_frac": {k: v["frac_of_hbm_peak"] for k, v in r["passes"].items()},
                      "mfma_frac": {k: v["frac_of_bf16_mfma_peak"] for k, v in r["passes"].items()},
                      "x2_ms": r["trunk_passes_ms_x2"], "pool_refine": r.get("pool_refine"), "checksums": r["checksums"]}))
