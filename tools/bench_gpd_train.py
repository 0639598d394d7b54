#!/usr/bin/env python3
"""The GPD comparator's training step on a resident batch (main_1v_gpd.py:97-106: forward, nll_loss, backward): the
libpngpd autograd node (gpd_ops.GPDNetFn) next to the ATen / MIOpen composite the reference would run on this GPU.

    python tools/bench_gpd_train.py [--batch 64] [--chann 12] [--steps 50]

One JSON line.  Not the hot path: reported so the comparator's HIP backward has a number."""
import argparse
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.nn.functional as F
    from pointnetgpd_amd.model.gpd import GPDClassifier
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--chann", type=int, default=12)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--hip-only", action="store_true", help="skip the ATen / MIOpen leg (kernel traces)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = GPDClassifier(a.chann).to(dev).train()
    x = torch.rand(a.batch, a.chann, 60, 60, device=dev) * (torch.rand(a.batch, 1, 60, 60, device=dev) < 0.3)
    t = torch.randint(0, 2, (a.batch,), device=dev)

    def hip():
        m.zero_grad(set_to_none=True)
        F.nll_loss(m(x), t).backward()

    def aten():
        m.zero_grad(set_to_none=True)
        h = F.max_pool2d(F.conv2d(x, m.conv1.weight, m.conv1.bias), 2, 2)
        h = F.max_pool2d(F.conv2d(h, m.conv2.weight, m.conv2.bias), 2, 2)
        h = F.relu(F.linear(h.view(-1, 7200), m.fc1.weight, m.fc1.bias))
        F.nll_loss(F.log_softmax(F.linear(h, m.fc2.weight, m.fc2.bias), -1), t).backward()

    out = {"batch": a.batch, "chann": a.chann, "steps": a.steps}
    for name, fn in (("hip", hip), ("aten_miopen", aten))[:1 if a.hip_only else 2]:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[f"{name}_ms"] = round(e0.elapsed_time(e1) / a.steps, 4)
    if not a.hip_only:
        hip()
        gh = [p.grad.clone() for p in m.parameters()]
        aten()
        out["max_rel_grad_diff_vs_aten"] = max(((g - p.grad).abs().max() / p.grad.abs().max()).item()
                                               for g, p in zip(gh, m.parameters()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
