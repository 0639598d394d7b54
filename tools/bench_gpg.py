"""GPG sampler: GPU (pointnetgpd_amd.gpg) vs the numpy oracle on the host cores, per sample point.
Usage: python tools/bench_gpg.py [--P 3000 20000] [--samples 150] [--cpu-draws 4]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gpg_oracle as go  # noqa: E402  (CPU baseline leg only)
from pointnetgpd_amd import gpg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, nargs="+", default=[3000, 20000])
    ap.add_argument("--samples", type=int, default=150)
    ap.add_argument("--cpu-draws", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    out = []
    for P in a.P:
        pts, nrm = go.synth_scene("cylinder", P, 41)
        pts32 = pts.astype(np.float32)
        pfs = pts32[pts32[:, 2] > 0.01]
        draws = np.random.default_rng(5).integers(0, len(pfs), a.samples)
        s = gpg.GpgGraspSamplerPcl(device=dev)
        cloud_d = torch.from_numpy(pts32).to(dev)
        for _ in range(2):
            res = s.sample_grasps(cloud_d, pfs, nrm, 10 ** 9, a.samples, sample_indices=draws, as_array=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            res = s.sample_grasps(cloud_d, pfs, nrm, 10 ** 9, a.samples, sample_indices=draws, as_array=True)
        torch.cuda.synchronize()
        gpu_s = (time.perf_counter() - t0) / a.reps
        # device-only time of the two kernels on the sweep-sized problem
        g = gpg._gripper_dict(gpg.ROBOTIQ_85)
        boxes = torch.from_numpy(gpg.hand_boxes(g)).to(dev)
        Q = a.samples * 19 * 21
        poses = torch.zeros(Q, 12, dtype=torch.float64, device=dev)
        poses[:, 3] = 1; poses[:, 7] = 1; poses[:, 11] = 1
        poses[:, :3] = torch.from_numpy(pts[np.random.default_rng(1).integers(0, P, Q)]).to(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gpg.hand_box_counts(cloud_d, poses, boxes)
        ev0.record()
        for _ in range(5):
            gpg.hand_box_counts(cloud_d, poses, boxes)
        ev1.record(); torch.cuda.synchronize()
        sweep_ms = ev0.elapsed_time(ev1) / 5
        index = gpg.CloudIndex(cloud_d)
        ev0.record()
        for _ in range(5):
            gpg.CloudIndex(cloud_d)
        ev1.record(); torch.cuda.synchronize()
        index_ms = ev0.elapsed_time(ev1) / 5
        gpg.hand_box_counts(cloud_d, poses, boxes, index=index)
        ev0.record()
        for _ in range(5):
            gpg.hand_box_counts(cloud_d, poses, boxes, index=index)
        ev1.record(); torch.cuda.synchronize()
        sweep_idx_ms = ev0.elapsed_time(ev1) / 5
        q_d = torch.from_numpy(pfs[draws].astype(np.float64)).to(dev)
        n_d = torch.from_numpy(nrm).to(dev)
        gpg.normal_moments(cloud_d, n_d, q_d, 0.1925)
        ev0.record()
        for _ in range(5):
            gpg.normal_moments(cloud_d, n_d, q_d, 0.1925)
        ev1.record(); torch.cuda.synchronize()
        mom_ms = ev0.elapsed_time(ev1) / 5
        t0 = time.perf_counter()
        ref = go.sample_grasps(pts32.astype(np.float64), pfs.astype(np.float64), nrm, draws[:a.cpu_draws], 10 ** 9, a.cpu_draws)
        cpu_per_draw = (time.perf_counter() - t0) / a.cpu_draws
        out.append(dict(P=P, samples=a.samples, grasps=int(len(res)), gpu_s_per_scene=gpu_s,
                        gpu_ms_per_draw=gpu_s / a.samples * 1e3, cpu_oracle_s_per_draw=cpu_per_draw,
                        speedup_per_draw=cpu_per_draw / (gpu_s / a.samples),
                        sweep_kernel_brute_ms=sweep_ms, sweep_kernel_indexed_ms=sweep_idx_ms, index_build_ms=index_ms,
                        sweep_pairs_per_s_brute=Q * P / (sweep_ms * 1e-3),
                        moments_kernel_ms=mom_ms, potential=s.last_stats["potential"]))
        print(json.dumps(out[-1]))
    return out


if __name__ == "__main__":
    main()
