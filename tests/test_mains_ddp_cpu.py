"""CPU: the train/eval CLI end to end (BASELINE config 1 plumbing: CPU tensors -> ATen composite),
whole-module checkpoints, and the data-parallel gradient averaging over gloo with world_size 2."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_train_then_test_cpu(tmp_path):
    from pointnetgpd_amd import mains
    common = ["--batch-size", "16", "--num-workers", "0", "--synthetic", "64", "--max-batches", "2",
              "--model-path", str(tmp_path / "models"), "--log-dir", str(tmp_path / "log"), "--seed", "3",
              "--tag", "t"]
    r = mains.run("1v", ["--mode", "train", "--epoch", "1"] + common)
    assert 0.0 <= r["train_acc"] <= 1.0 and np.isfinite(r["test_loss"])
    ckpt = tmp_path / "models" / "t_0.model"
    assert ckpt.exists()
    # eval from the whole-module pickle (main_1v.py:152-155, weights_only=False)
    r2 = mains.run("1v", ["--mode", "test", "--load-model", str(ckpt)] + common)
    assert np.isfinite(r2["test_loss"])
    # the pickled module is OUR class and round-trips its state_dict keys (44 params + 30 buffers)
    m = torch.load(ckpt, weights_only=False)
    assert type(m).__module__ == "pointnetgpd_amd.model.pointnet"
    assert len(list(m.named_parameters())) == 44
    assert len([k for k in m.state_dict() if "running_" in k or "num_batches" in k]) == 30
    # the state_dict sidecar: tensors only (weights_only=True), the keys / values of the pickled module
    sd = torch.load(str(ckpt) + ".state_dict", weights_only=True)
    assert list(sd) == list(m.state_dict()) and all(torch.equal(sd[k], v) for k, v in m.state_dict().items())
    # the grasps/s scalar of the epoch next to the reference's four scalars (JSONL sink when tensorboard is absent)
    import glob, json
    files = glob.glob(str(tmp_path / "log" / "t" / "scalars.jsonl"))
    if files:
        tags = {json.loads(l)["tag"] for l in open(files[0])}
        assert {"train_loss", "train_acc", "test_acc", "test_loss", "train_grasps_per_s"} <= tags
    with pytest.raises(SystemExit, match="--device-data"):
        mains.run("1v", ["--mode", "train", "--epoch", "1", "--device-data", "--cuda"] + common)


@pytest.mark.parametrize("variant,k,n", [("1v_mc", 3, 750), ("fullv", 2, 1000), ("fullv_mc", 3, 1000)])
def test_cli_variants_cpu(variant, k, n, tmp_path):
    from pointnetgpd_amd import mains
    assert mains.VARIANTS[variant]["k"] == k and mains.VARIANTS[variant]["num_points"] == n
    r = mains.run(variant, ["--mode", "train", "--epoch", "1", "--batch-size", "8", "--num-workers", "0",
                            "--synthetic", "16", "--max-batches", "1", "--model-path", str(tmp_path / "m"),
                            "--log-dir", str(tmp_path / "l"), "--seed", "1"])
    assert np.isfinite(r["test_loss"])


def test_reference_flag_surface():
    """Every flag of the reference parser (main_1v.py:18-33) exists with the same default."""
    from pointnetgpd_amd import mains
    p = mains.build_parser()
    d = vars(p.parse_args(["--mode", "train"]))
    ref = dict(tag="default", epoch=200, batch_size=1, cuda=False, gpu=0, lr=0.005, load_model="", load_epoch=-1,
               model_path="./assets/learned_models", log_interval=10, save_interval=1)
    for k_, v in ref.items():
        assert d[k_] == v, k_


def test_reference_pickle_alias_roundtrip(tmp_path):
    """A module pickled under the reference's module path ``model.pointnet`` loads onto the mirror."""
    import pointnetgpd_amd
    from pointnetgpd_amd.model import pointnet as pn
    pointnetgpd_amd.install_reference_aliases()
    import importlib
    import pickle
    ref_mod = importlib.import_module("model.pointnet")
    assert ref_mod.PointNetCls is pn.PointNetCls
    assert importlib.import_module("model.dataset").PointGraspOneViewDataset is not None
    # a pickle that names the class by the reference's path resolves to the mirror
    blob = pickle.dumps(("model.pointnet", "PointNetCls"))
    mod_name, cls_name = pickle.loads(blob)
    cls = getattr(importlib.import_module(mod_name), cls_name)
    m = cls(64, 3, 2)
    assert isinstance(m, pn.PointNetCls)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _ddp_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    import torch.nn.functional as F
    from pointnetgpd_amd import ddp
    from pointnetgpd_amd.model.pointnet import PointNetCls
    r, w, _ = ddp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                 # different initial weights per rank on purpose
    m = PointNetCls(64, 3, 2).train()
    avg = ddp.GradAverager(m)                     # -> rank 0's weights everywhere
    g = torch.Generator().manual_seed(5)
    x_all = torch.randn(8, 3, 64, generator=g); y_all = torch.arange(8) % 2
    xs, ys = x_all[rank * 4:(rank + 1) * 4], y_all[rank * 4:(rank + 1) * 4]
    with torch.no_grad():
        m.feat.bn1.running_mean.add_(rank)        # rank-local drift that sync_buffers must undo
    avg.sync_buffers()
    logp, _ = m(xs)
    F.nll_loss(logp, ys).backward()
    local = {n: p.grad.clone() for n, p in m.named_parameters()}
    avg.average_gradients()
    torch.save(dict(local=local, avg={n: p.grad.clone() for n, p in m.named_parameters()},
                    w0=m.fc3.weight.detach().clone(), rm=m.feat.bn1.running_mean.clone()),
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_averager_gloo_world2(tmp_path):
    world, port = 2, _free_port()
    mp.start_processes(_ddp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "r0.pt"); r1 = torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["w0"], r1["w0"])                       # parameters were broadcast from rank 0
    for n in r0["local"]:
        mean = (r0["local"][n] + r1["local"][n]) / 2
        assert torch.allclose(r0["avg"][n], mean, atol=1e-7), n  # all-reduce(sum)/world
        assert torch.equal(r0["avg"][n], r1["avg"][n]), n        # identical on both ranks
    # per-rank BN statistics differ (as under nn.DataParallel) -> local grads differ
    assert not torch.allclose(r0["local"]["fc1.weight"], r1["local"]["fc1.weight"])
    # running stats: rank 1's drift was overwritten by rank 0's buffers before the forward, then each
    # rank applied its own batch statistics
    assert abs(r1["rm"].mean().item() - r0["rm"].mean().item()) < 0.5


def _ragged_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    import torch.nn.functional as F
    from pointnetgpd_amd import ddp
    from pointnetgpd_amd.model.pointnet import PointNetCls
    ddp.init_from_env("gloo")
    torch.manual_seed(7)
    m = PointNetCls(64, 3, 2).train()
    avg = ddp.GradAverager(m)
    g = torch.Generator().manual_seed(11)
    x_all = torch.randn(8, 3, 64, generator=g); y_all = torch.arange(8) % 2
    out = {}
    # step A: ragged per-rank batches (my_collate dropped samples): 5 on rank 0, 3 on rank 1
    lo, hi = (0, 5) if rank == 0 else (5, 8)
    m.zero_grad()
    lp, _ = m(x_all[lo:hi])
    assert avg.backward(F.nll_loss(lp, y_all[lo:hi], reduction="sum"), hi - lo) is None
    out["ragged"] = {n: p.grad.clone() for n, p in m.named_parameters()}
    # what this rank alone contributes (sum of its per-sample gradients)
    m.zero_grad()
    lp, _ = m(x_all[lo:hi])
    F.nll_loss(lp, y_all[lo:hi], reduction="sum").backward()
    out["local_sum"] = {n: p.grad.clone() for n, p in m.named_parameters()}
    # step B: rank 1 is left with ONE sample (train-mode BatchNorm cannot run): it sits the step out, nobody hangs
    m.zero_grad()
    if rank == 0:
        lp, _ = m(x_all[0:5])
        avg.backward(F.nll_loss(lp, y_all[0:5], reduction="sum"), 5)
    else:
        avg.backward(None, 1)
    out["sit_out"] = {n: p.grad.clone() for n, p in m.named_parameters()}
    # step C: every sample of rank 1's batch was dropped (my_collate -> None)
    m.zero_grad()
    if rank == 0:
        lp, _ = m(x_all[0:5])
        avg.backward(F.nll_loss(lp, y_all[0:5], reduction="sum"), 5)
    else:
        avg.backward(None, 0)
    out["empty"] = {n: p.grad.clone() for n, p in m.named_parameters()}
    # running statistics drifted apart (per-replica BatchNorm): sync_buffers before eval makes rank 0's everyone's
    avg.sync_buffers()
    m.eval()
    with torch.no_grad():
        out["eval_logp"] = m(x_all)[0].clone()
    out["rm"] = m.feat.bn3.running_mean.clone()
    torch.save(out, os.path.join(out_dir, f"rag{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_batches_weighted_mean_and_sit_out(tmp_path):
    """per-rank kept-sample counts differ under my_collate.  The combined gradient is the per-sample mean
    over the GLOBAL batch (what DataParallel's gathered nll_loss gives), a rank with < 2 samples contributes zeros and
    still joins the collectives, and eval after sync_buffers() uses rank 0's running statistics on every rank."""
    from pointnetgpd_amd import mains
    assert mains.my_collate([None, None]) is None              # an all-dropped batch no longer raises in default_collate
    world, port = 2, _free_port()
    mp.start_processes(_ragged_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "rag0.pt"); r1 = torch.load(tmp_path / "rag1.pt")
    for n in r0["ragged"]:
        want = (r0["local_sum"][n] + r1["local_sum"][n]) / 8.0
        assert torch.allclose(r0["ragged"][n], want, atol=1e-6), n
        assert torch.equal(r0["ragged"][n], r1["ragged"][n]), n
        for key in ("sit_out", "empty"):
            assert torch.allclose(r0[key][n], r0["local_sum"][n] / 5.0, atol=1e-6), (key, n)
            assert torch.equal(r0[key][n], r1[key][n]), (key, n)
    assert torch.equal(r0["rm"], r1["rm"]) and torch.equal(r0["eval_logp"], r1["eval_logp"])


def test_shard_grasps():
    from pointnetgpd_amd.scoring import shard_grasps
    spans = [shard_grasps(100000, r, 8) for r in range(8)]
    assert spans[0][0] == 0 and spans[-1][1] == 100000
    assert all(spans[i][1] == spans[i + 1][0] for i in range(7))
    assert sum(e - s for s, e in spans) == 100000
    assert shard_grasps(5, 7, 8) == (5, 5)


def _score_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from pointnetgpd_amd import ddp, scoring
    ddp.init_from_env("gloo")
    G = 11
    grasps = np.arange(G * 15, dtype=np.float64).reshape(G, 5, 3)

    def fake_score(cloud, gs):       # stands in for GraspScorer.score (needs a GPU): deterministic per candidate
        gid = torch.from_numpy(gs[:, 0, 0] / 15.0).float()
        return dict(pred=(gid.long() % 3), score=torch.sin(gid) * 0.5 + 0.5, counts=(gid * 7).int(), valid=gid.long() % 4 != 1)

    res = scoring.score_scene_distributed(fake_score, None, grasps)
    torch.save({k: v.cpu() for k, v in res.items()}, os.path.join(out_dir, f"s{rank}.pt"))
    dist.barrier(); dist.destroy_process_group()


def test_score_scene_distributed_gloo(tmp_path):
    world, port = 3, _free_port()       # 11 candidates over 3 ranks: slices 4,4,3
    mp.start_processes(_score_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    res = [torch.load(tmp_path / f"s{r}.pt") for r in range(world)]
    gid = torch.arange(11).float()
    for r in res:
        assert torch.equal(r["pred"], gid.long() % 3)
        assert torch.allclose(r["score"], torch.sin(gid) * 0.5 + 0.5)
        assert torch.equal(r["counts"], (gid * 7).int())
        assert torch.equal(r["valid"], gid.long() % 4 != 1)
        sc = r["score"][r["order"]]
        assert (sc[:-1] >= sc[1:]).all() and set(r["order"].tolist()) == set(torch.nonzero(r["valid"]).squeeze(1).tolist())


def _cli_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from pointnetgpd_amd import mains
    r = mains.run("1v_mc", ["--mode", "train", "--epoch", "1", "--batch-size", "8", "--num-workers", "0",
                            "--synthetic", "32", "--max-batches", "2", "--model-path", os.path.join(out_dir, "m"),
                            "--log-dir", os.path.join(out_dir, "l"), "--seed", "2", "--tag", "ddp"])
    torch.save(r, os.path.join(out_dir, f"cli{rank}.pt"))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_cli_two_ranks_gloo(tmp_path):
    """The train loop under torchrun-style env vars (DistributedSampler + gradient all-reduce) on CPU/gloo."""
    world, port = 2, _free_port()
    mp.start_processes(_cli_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "cli0.pt"); r1 = torch.load(tmp_path / "cli1.pt")
    assert np.isfinite(r0["test_loss"]) and np.isfinite(r1["test_loss"])
    assert (tmp_path / "m" / "ddp_0.model").exists()


def test_per_rank_batch_keeps_the_reference_global_batch():
    """--batch-size is the global batch (the reference's nn.DataParallel scatters ONE batch of --batch-size,
    main_1v.py:158-165): each of `world` ranks loads batch_size / world."""
    from pointnetgpd_amd import mains
    assert mains.per_rank_batch(512, 1) == 512 and mains.per_rank_batch(512, 8) == 64
    with pytest.raises(ValueError, match="divisible"):
        mains.per_rank_batch(100, 8)


REF = "/root/reference/PointNetGPD"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "model", "pointnet.py")),
                    reason="needs the reference checkout (build container only)")
def test_saved_checkpoint_loads_in_the_unmodified_reference(tmp_path):
    """Two-way pickle compatibility: a checkpoint written by mains.save_model unpickles in a process that only has
    the REFERENCE's model.pointnet on its path (as kinect2grasp.py / main_test.py do) and computes the same output;
    the same file also loads back onto this implementation."""
    import subprocess
    from pointnetgpd_amd import mains
    from pointnetgpd_amd.model.pointnet import PointNetCls
    torch.manual_seed(3)
    m = PointNetCls(64, 3, 3).eval()
    x = torch.randn(4, 3, 64)
    with torch.no_grad():
        m(x)                                         # populate the fold cache: it must not be pickled
        lp, _ = m(x)
    path = str(tmp_path / "ckpt.model")
    mains.save_model(m, path)
    assert type(m).__module__ == "pointnetgpd_amd.model.pointnet"          # restored after saving
    torch.save(dict(x=x, lp=lp), str(tmp_path / "io.pt"))
    code = (f"import sys, torch; sys.path.insert(0, {REF!r})\n"
            f"m = torch.load({path!r}, weights_only=False)\n"
            "assert type(m).__module__ == 'model.pointnet' and 'reference' in sys.modules['model.pointnet'].__file__\n"
            f"io = torch.load({str(tmp_path / 'io.pt')!r})\n"
            "m.eval()\n"
            "with torch.no_grad(): lp, _ = m(io['x'])\n"
            "assert torch.allclose(lp, io['lp'], atol=1e-6), (lp - io['lp']).abs().max()\n"
            "print('REFERENCE-LOADED-OK')\n")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), env=env)
    assert "REFERENCE-LOADED-OK" in out.stdout, out.stderr[-1500:]
    from pointnetgpd_amd import install_reference_aliases
    install_reference_aliases()
    back = torch.load(path, weights_only=False)
    assert isinstance(back, PointNetCls)
    with torch.no_grad():
        assert torch.equal(back.eval()(x)[0], lp)


def test_save_model_failure_raises_single_process_and_none_type_attribute_pickles(tmp_path):
    """ADVICE r3 (low x2): a failed checkpoint raises when there is one process (the reference's ``torch.save`` would;
    silently losing every checkpoint of a run is worse than stopping), is reported + returned under world > 1 (run()
    broadcasts the flag); and the pickler needs no ``dispatch`` override — a module attribute holding ``type(None)``
    pickles through pickle's own ``save_type``."""
    from pointnetgpd_amd import mains
    from pointnetgpd_amd.model.pointnet import PointNetCls
    m = PointNetCls(64, 3, 2)
    bad = str(tmp_path / "no_such_dir" / "x.model")
    with pytest.raises(Exception):
        mains.save_model(m, bad)
    assert mains.save_model(m, bad, world=2) is False
    m.some_type = type(None)
    m.other = (type(NotImplemented), type(...))
    ok = str(tmp_path / "ok.model")
    assert mains.save_model(m, ok) is True
    from pointnetgpd_amd import install_reference_aliases
    install_reference_aliases()
    back = torch.load(ok, weights_only=False)
    assert back.some_type is type(None) and back.other == (type(NotImplemented), type(...))
