import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn.functional as F
import bench
from pointnetgpd_amd import train, ops
from pointnetgpd_amd.optim import FlatAdam
dev = torch.device("cuda:0")
def run(B, N, k, target, reps=100):
    ops.TRAIN_TARGET_BLOCKS = target
    m = bench.build_model(N, k, dev).train()
    opt = FlatAdam(m.parameters(), lr=0.005)
    x = bench.synth_clouds(B, N, 1, dev); y = (torch.arange(B, device=dev) % k).long()
    def step():
        opt.zero_grad(); lp, _ = m(x); F.nll_loss(lp, y).backward(); opt.step()
    for _ in range(5): step()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): step()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3, ops.train_splits(B, N)
for B, N, k in [(16, 750, 2), (64, 750, 2), (128, 750, 2), (512, 1024, 3), (1024, 1024, 2), (512, 4096, 2), (100, 1000, 2)]:
    a = run(B, N, k, 1024, 40 if B >= 512 else 150); b = run(B, N, k, 0, 40 if B >= 512 else 150)
    print(f"B={B} N={N}: old rule S={a[1]} {a[0]:.3f} ms  |  cost model S={b[1]} {b[0]:.3f} ms")
