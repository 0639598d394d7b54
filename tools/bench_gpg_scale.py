#!/usr/bin/env python3
"""The GPG sampler at candidate-generation scale (BASELINE configs[4]'s upstream; grasp_sampler.py:1389-1656): a
50,000-point scene, 20,000 sample points -> candidates/s, with a per-stage breakdown (synchronising run) next to the
free-running time, for the fused per-(sample point, rotation) sweep and the per-pose sweep it replaces."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import gpg_oracle as go          # synthetic scene generator only
from pointnetgpd_amd import gpg

dev = torch.device("cuda:0")
P, SAMPLES = int(os.environ.get("P", 50000)), int(os.environ.get("SAMPLES", 20000))
pts, nrm = go.synth_scene(os.environ.get("SCENE", "cylinder"), P, 41)
pts32 = pts.astype(np.float32)
pfs = pts32[pts32[:, 2] > 0.01]
cloud_d = torch.from_numpy(pts32).to(dev)
ref = None
for fused in ([True, False] if os.environ.get("BOTH", "1") == "1" else [True]):
    s = gpg.GpgGraspSamplerPcl(device=dev, fused_sweep=fused, **({"batch_samples": int(os.environ["BATCH"])} if "BATCH" in os.environ else {}))
    s.sample_grasps(cloud_d, pfs, nrm, 10 ** 9, SAMPLES, seed=0, as_array=True)   # same sizes: allocator + pinned pools warm
    torch.cuda.synchronize()
    dts = []
    for _ in range(3):
        t0 = time.perf_counter()
        res = s.sample_grasps(cloud_d, pfs, nrm, 10 ** 9, SAMPLES, seed=1, as_array=True)
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[1]
    s.profile = {}
    s.sweep_stats = torch.zeros(4, dtype=torch.int64, device=dev)
    s.pushin_stats = torch.zeros(4, dtype=torch.int64, device=dev)
    res2 = s.sample_grasps(cloud_d, pfs, nrm, 10 ** 9, SAMPLES, seed=1, as_array=True)
    prof = {k: round(v * 1e3, 2) for k, v in s.profile.items()}
    assert np.array_equal(res, res2)
    if ref is None:
        ref = res
    print(json.dumps({"P": P, "samples": SAMPLES, "fused_sweep": fused, "batch_samples": s.batch_samples, "candidates": int(len(res)),
                      "seconds": round(dt, 4), "candidates_per_s": round(len(res) / dt, 1),
                      "identical_to_first_variant": bool(np.array_equal(res, ref)), "stage_ms_synchronised": prof,
                      "sweep_units_chunks_passed_evaluated_exact": s.sweep_stats.tolist(),
                      "pushin_units_chunks_passed_evaluated_exact": s.pushin_stats.tolist(),
                      "potential": s.last_stats["potential"]}), flush=True)
