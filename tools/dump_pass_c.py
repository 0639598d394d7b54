#!/usr/bin/env python3
"""Debug helper: run pass B + C on the bench_pass operands with the library in PNGPD_LIB and save the partials."""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PNGPD_PASSES"] = "none"
import torch
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_pass.py"))
from pointnetgpd_amd import ops
out = ops.trunk_fwd_train(g["x"], g["T"], g["w1"], g["b1"], g["s1c"], g["t1c"], g["w2p"], g["s2c"], g["t2c"], g["w3p"], g["S"], g["z2t"])
torch.save([o.cpu() for o in out], sys.argv[1])
