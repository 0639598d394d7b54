import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import bench
dev = torch.device("cuda:0")
B, N, k = 64, 256, 2
x = bench.synth_clouds(B, N, 1, dev); y = (torch.arange(B, device=dev) % k).long()
stage = sys.argv[1]
m = bench.build_model(N, k, dev)
if stage == "infer":
    with torch.no_grad():
        m(x); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out, _ = m(x)
        g.replay(); torch.cuda.synchronize(); print("infer graph ok", out.sum().item())
else:
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=0.005, capturable=True)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            lp, _ = m(x); l = F.nll_loss(lp, y); l.backward(); opt.step()
        del lp, l
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    opt.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    print("capturing", stage, flush=True)
    with torch.cuda.graph(g):
        lp, _ = m(x)
        loss = F.nll_loss(lp, y)
        if stage in ("bwd", "full"):
            loss.backward()
        if stage == "full":
            opt.step()
    print("captured", flush=True)
    g.replay(); torch.cuda.synchronize(); print(stage, "graph ok", loss.item())
