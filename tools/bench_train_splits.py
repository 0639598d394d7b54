import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn.functional as F
import bench
from pointnetgpd_amd import train, ops
from pointnetgpd_amd.optim import FlatAdam
dev = torch.device("cuda:0")
def run(B, N, k, target, reps=300):
    ops.TRAIN_TARGET_BLOCKS = target
    m = bench.build_model(N, k, dev).train()
    opt = FlatAdam(m.parameters(), lr=0.005)
    x = bench.synth_clouds(B, N, 1, dev); y = (torch.arange(B, device=dev) % k).long()
    def step():
        opt.zero_grad(); lp, _ = m(x); F.nll_loss(lp, y).backward(); opt.step()
    for _ in range(10): step()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): step()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3, ops.train_splits(B, N)
for B, N in [(64, 750), (64, 1024), (128, 750), (32, 750), (256, 1024)]:
    for target in (1024, 768, 512, 384, 256):
        dt, S = run(B, N, 2, target, 200)
        print(f"B={B} N={N} target={target} S={S}: {dt:.3f} ms/step")
