"""Per-model arithmetic (arith.py; SURVEY.md §8(b) "no global mutable state"): two models with different precisions
interleaved in one process — and driven from two threads, as nn.DataParallel would (main_1v.py:158-165) — produce
exactly what each produces alone; the setting travels in the whole-module pickle (main_1v.py:177-178)."""
import io
import threading

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import build_model, synth_cloud

pytestmark = pytest.mark.gpu

MODES = ("fp32", "bf16x3", "bf16")


def _train_once(m, x, y):
    m.train()
    m.zero_grad(set_to_none=True)
    lp, _ = m(x)
    loss = F.nll_loss(lp, y)
    loss.backward()
    return lp.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}


def _alone(mode, dev, x, y):
    m = build_model(256, 3, 21, 22).to(dev).set_precision(mode)
    m.eval()
    with torch.no_grad():
        ev = m(x)[0].clone()
    tr, grads = _train_once(m, x, y)
    return ev, tr, grads


def test_two_models_interleaved(cuda_device):
    dev = cuda_device
    x = synth_cloud(16, 256, 5, "gauss").to(dev)
    y = (torch.arange(16, device=dev) % 3).long()
    alone = {mode: _alone(mode, dev, x, y) for mode in MODES}
    assert not torch.equal(alone["fp32"][0], alone["bf16"][0])          # the modes really differ
    models = {mode: build_model(256, 3, 21, 22).to(dev).set_precision(mode) for mode in MODES}
    got = {}
    for mode in MODES:                       # eval forwards interleaved
        models[mode].eval()
        with torch.no_grad():
            got[mode] = [models[mode](x)[0].clone()]
    for mode in reversed(MODES):             # training steps interleaved, opposite order
        got[mode] += list(_train_once(models[mode], x, y))
    for mode in MODES:
        ev, tr, grads = alone[mode]
        assert torch.equal(got[mode][0], ev), mode
        assert torch.equal(got[mode][1], tr), mode
        for n, g in grads.items():
            assert torch.equal(got[mode][2][n], g), (mode, n)


def test_two_threads_two_precisions(cuda_device):
    dev = cuda_device
    x = synth_cloud(8, 256, 6, "box").to(dev)
    ref = {}
    for mode in ("fp32", "bf16"):
        m = build_model(256, 2, 31, 32).to(dev).eval().set_precision(mode)
        with torch.no_grad():
            ref[mode] = m(x)[0].clone()
    out, errs = {}, []
    # models are built on the main thread: build_model seeds the process-global RNG
    built = {mode: build_model(256, 2, 31, 32).to(dev).eval().set_precision(mode) for mode in ("fp32", "bf16")}

    def work(mode):
        try:
            torch.cuda.set_device(dev)
            m = built[mode]
            res = []
            with torch.no_grad():
                for _ in range(20):
                    res.append(m(x)[0].clone())
            torch.cuda.synchronize()
            out[mode] = res
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(mode,)) for mode in ("fp32", "bf16")]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for mode in ("fp32", "bf16"):
        for r in out[mode]:
            assert torch.equal(r, ref[mode]), mode


def test_pickle_carries_the_arithmetic(cuda_device, tmp_path):
    from pointnetgpd_amd import install_reference_aliases
    from pointnetgpd_amd.mains import save_model
    dev = cuda_device
    x = synth_cloud(4, 256, 7, "box").to(dev)
    m = build_model(256, 2, 41, 42).to(dev).eval().set_precision("bf16x3", refine=True)
    with torch.no_grad():
        a = m(x)[0]
    path = str(tmp_path / "m.model")
    save_model(m, path)
    install_reference_aliases()
    m2 = torch.load(path, map_location=dev, weights_only=False)
    assert m2.get_precision().infer == "bf16x3" and m2.get_precision().infer_refine
    assert m2.feat.stn.get_precision().train == "bf16x3"
    with torch.no_grad():
        b = m2(x)[0]
    assert torch.equal(a, b)
