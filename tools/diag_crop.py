import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import crop_oracle as co
from pointnetgpd_amd import crop
from tests.test_gpu_crop_scoring import _scene
dev = torch.device("cuda:0")
pc, grasps = _scene(64, 20000, 7)
pc = pc.astype(np.float64)
ind_ref, pts_ref = co.collect_pc_infer(grasps, pc)
frames = torch.from_numpy(crop.frames_from_grasps_infer(grasps)).to(dev)
cloud = torch.from_numpy(pc).to(dev)
counts, idx = crop.crop_count_compact(cloud, frames, max_keep=2048)
N = 32
out, valid = crop.crop_resample(cloud, frames, counts, idx, N, crop.MODE_INFER, 20, seed=11)
bad = 0; worst = 0
for g in range(64):
    if not valid[g]: continue
    ref = pts_ref[g]
    cols = out[g].cpu().numpy().T.astype(np.float64)
    for c in cols:
        d = np.abs(ref - c).max(1).min()
        rel = d
        worst = max(worst, d)
        if not ((ref.astype(np.float32) == c.astype(np.float32)).all(1)).any(): bad += 1
print("bad", bad, "worst abs diff to nearest ref point", worst)
# direct fp64 comparison of transformed coordinates using sel injection
sel = torch.zeros(64, N, dtype=torch.int32, device=dev)
out0, _ = crop.crop_resample(cloud, frames, counts, idx, N, crop.MODE_INFER, 20, sel=sel)
for g in range(3):
    if valid[g]:
        print(g, out0[g, :, 0].cpu().numpy(), pts_ref[g][0].astype(np.float32), pts_ref[g][0])
