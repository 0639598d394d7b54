#!/usr/bin/env python3
"""How many DISTINCT arg-max points does a cloud have over its 1,024 pooled channels?  (The pool refinement re-evaluates
layers 1-2 in fp32 at B x 1,024 points; only the distinct ones are needed.)  Headline iid clouds and diverse clouds,
both trunks of a random-init PointNetCls."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pointnetgpd_amd import ops
from pointnetgpd_amd.model import pointnet as pn
dev = torch.device("cuda:0")
out = {}
for name, x in (("iid_box", bench.synth_clouds(256, 1024, 1, dev)), ("diverse", bench.synth_clouds_diverse(256, 1024, 2, dev))):
    m = bench.build_model(1024, 2, dev)
    with torch.no_grad():
        _, trans = m(x)
    for tname, mod, tr, relu in (("stn", m.feat.stn, None, 1), ("feat", m.feat, trans, 0)):
        w = pn._trunk_infer_weights_x3(mod, dev)
        w1, b1, w2x, b2, w3x, b3 = w[0], w[1], w[2], w[3], w[4], w[5]
        pooled, arg = ops.trunk_fwd_infer_bf(x, tr, w1, b1, w2x, b2, w3x, b3, relu, nterms=3, want_arg=True)
        u = torch.tensor([arg[b].unique().numel() for b in range(arg.shape[0])], dtype=torch.float32)
        out[f"{name}.{tname}"] = {"unique_argmax_points_per_cloud_min_mean_max": [int(u.min()), round(float(u.mean()), 1), int(u.max())]}
print(json.dumps(out))
