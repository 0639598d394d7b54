import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn.functional as F
import bench
from pointnetgpd_amd import train
from pointnetgpd_amd.optim import FlatAdam
dev = torch.device("cuda:0")
def run(B, N, k, flat, seq="fused", reps=200):
    train.set_sequencing(seq)
    m = bench.build_model(N, k, dev).train()
    opt = FlatAdam(m.parameters(), lr=0.005) if flat else torch.optim.Adam(m.parameters(), lr=0.005, fused=True)
    x = bench.synth_clouds(B, N, 1, dev); y = (torch.arange(B, device=dev) % k).long()
    def step():
        opt.zero_grad(); loss, _, _ = m.forward_loss(x, y); train.loss_backward(loss); opt.step()   # mains.py's step
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    # host-only time: how long to ENQUEUE
    t0 = time.perf_counter()
    for _ in range(50): step()
    th = (time.perf_counter() - t0) / 50
    torch.cuda.synchronize()
    train.set_sequencing("fused")
    return dt * 1e3, th * 1e3
for B, N, k, reps in [(64, 750, 2, 300), (1024, 1024, 2, 30)]:
    for flat, seq in [(False, "passes"), (False, "fused"), (True, "fused")]:
        dt, th = run(B, N, k, flat, seq, reps)
        print(f"B={B} N={N} seq={seq} flatadam={flat}: {dt:.3f} ms/step (host enqueue {th:.3f} ms)")
