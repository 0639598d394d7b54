import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from pointnetgpd_amd.model import pointnet as pn
dev = torch.device("cuda:0")
B, N, k = int(os.environ.get("B", 64)), int(os.environ.get("N", 750)), 2
m = bench.build_model(N, k, dev).eval()
x = bench.synth_clouds(B, N, 1, dev)
with torch.no_grad():
    for i in range(30):
        m(x)
torch.cuda.synchronize()
