"""Train / eval loops of the PointNet variants, with the reference's CLI.

One implementation behind four entry points that mirror the reference scripts
(same flags, same hyper-parameters, same printed lines, same checkpoint naming):

    main_1v.py       PointGraspOneViewDataset            N=750  k=2  good=bad=0.6   reference main_1v.py
    main_1v_mc.py    PointGraspOneViewMultiClassDataset  N=750  k=3  good=0.5 bad=1.2   main_1v_mc.py
    main_fullv.py    PointGraspDataset (50k pts, 20 views) N=1000 k=2  good=bad=0.6     main_fullv.py
    main_fullv_mc.py PointGraspMultiClassDataset         N=1000 k=3  good=0.5 bad=1.2   main_fullv_mc.py

Reference behaviour kept: Adam(lr 0.005) + StepLR(30, 0.5) with
``scheduler.step()`` at the start of every epoch; ``main_1v`` re-creates optimizer and scheduler every
epoch (main_1v.py:60-62) while the others keep one (main_fullv.py:112-116); ``F.nll_loss`` (mean) for
training, summed for eval; whole-module pickles ``<model-path>/<tag>_<epoch>.model``.

What changes: the reference's ``nn.DataParallel`` over ids [0,1,2,3] (``--gpu -1``) becomes one
process per GPU under ``torchrun`` with an RCCL gradient all-reduce (``pointnetgpd_amd.ddp``);
``torch.load`` passes ``weights_only=False`` (whole-module pickles fail otherwise on torch >= 2.6).
On ``--cuda`` the optimizer is ``optim.FlatAdam`` — the same Adam (main_1v.py:61) over one flat HBM buffer, one
launch per step, gradients written in place by the fused backward.  Running accuracy is accumulated on the device
and read at ``--log-interval`` / epoch end instead of the reference's per-batch ``.cpu()`` (main_1v.py:78).
Additive flags: ``--seed --num-workers --synthetic --max-batches --persistent-optimizer --precision --hip-graph
--device-data``.
"""
import argparse
import json
import os
import pickle
import time

import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim
import torch.utils.data
from torch.optim.lr_scheduler import StepLR

from . import ddp
from .model.pointnet import PointNetCls

VARIANTS = {
    "1v": dict(dataset="PointGraspOneViewDataset", fullview=False, k=2, num_points=750, good=0.6, bad=0.6,
               recreate_optimizer=True),
    "1v_mc": dict(dataset="PointGraspOneViewMultiClassDataset", fullview=False, k=3, num_points=750, good=0.5,
                  bad=1.2, recreate_optimizer=False),
    "fullv": dict(dataset="PointGraspDataset", fullview=True, k=2, num_points=1000, good=0.6, bad=0.6,
                  recreate_optimizer=False),
    "fullv_mc": dict(dataset="PointGraspMultiClassDataset", fullview=True, k=3, num_points=1000, good=0.5,
                     bad=1.2, recreate_optimizer=False),
}


def build_parser():
    p = argparse.ArgumentParser(description="pointnetGPD")
    p.add_argument("--tag", type=str, default="default")
    p.add_argument("--epoch", type=int, default=200)
    p.add_argument("--mode", choices=["train", "test"], required=True)
    p.add_argument("--batch-size", type=int, default=1)
    p.add_argument("--cuda", action="store_true")
    p.add_argument("--gpu", type=int, default=0)
    p.add_argument("--lr", type=float, default=0.005)
    p.add_argument("--load-model", type=str, default="")
    p.add_argument("--load-epoch", type=int, default=-1)
    p.add_argument("--model-path", type=str, default="./assets/learned_models", help="pre-trained model path")
    p.add_argument("--log-interval", type=int, default=10)
    p.add_argument("--save-interval", type=int, default=1)
    # additive (defaults reproduce the reference)
    p.add_argument("--seed", type=int, default=None, help="fix numpy/torch seeds (reference: time-based)")
    p.add_argument("--num-workers", type=int, default=32,
                   help="DataLoader workers of the NODE (the reference's single process had 32 for all its GPUs, "
                        "main_1v.py:124); under torchrun every rank starts num_workers / ranks-on-the-node")
    p.add_argument("--synthetic", type=int, default=0, metavar="G",
                   help="train/eval on G synthetic in-gripper clouds instead of the YCB files")
    p.add_argument("--max-batches", type=int, default=0, help="stop every epoch after this many batches")
    p.add_argument("--persistent-optimizer", action="store_true",
                   help="main_1v only: keep one optimizer/scheduler (fixes main_1v.py:60-62)")
    p.add_argument("--hip-graph", action="store_true",
                   help="single-GPU --cuda training: replay full-size batches from a captured HIP graph "
                        "(train.GraphedTrainStep); ragged batches (my_collate dropped samples) run eagerly")
    p.add_argument("--device-data", action="store_true",
                   help="--cuda: keep every cloud resident in HBM and crop/resample training batches on the GPU, "
                        "prefetched on a side stream under the step (device_loader.DeviceGraspLoader) instead of "
                        "DataLoader workers; under torchrun every rank walks its share of the epoch")
    p.add_argument("--precision", choices=["fp32", "bf16x3", "bf16"], default="fp32",
                   help="--cuda: arithmetic of the trunk contractions. fp32 = exact (default); bf16x3 = 3-term split "
                        "bf16 products on the bf16 matrix cores (meets the fp32 parity bars); bf16 = plain bf16 "
                        "operands and bf16 z2/g2 tiles (BASELINE configs[2]; does NOT meet 1e-3 in training)")
    p.add_argument("--log-dir", type=str, default="./assets/log/")
    return p


class _ScalarLog:
    """``logger.add_scalar`` sink: tensorboardX / torch.utils.tensorboard when importable, else JSONL."""

    def __init__(self, path):
        os.makedirs(path, exist_ok=True)
        self.w = None
        try:
            from tensorboardX import SummaryWriter
            self.w = SummaryWriter(path)
        except Exception:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.w = SummaryWriter(path)
            except Exception:
                self.f = open(os.path.join(path, "scalars.jsonl"), "a")

    def add_scalar(self, name, value, step):
        if self.w is not None:
            self.w.add_scalar(name, value, step)
        else:
            self.f.write(json.dumps({"tag": name, "value": float(value), "step": int(step)}) + "\n")
            self.f.flush()


def worker_init_fn(pid):
    np.random.seed(torch.initial_seed() % (2 ** 31 - 1))          # main_1v.py:44-45


def my_collate(batch):
    batch = list(filter(lambda x: x is not None, batch))            # main_1v.py:48-50
    if not batch:
        # every sample of this (per-rank) batch was dropped: the reference's default_collate raises on the empty
        # list; under one-process-per-GPU the peers would then wait in their collectives forever — hand the loop an
        # empty batch instead (it sits the step out)
        return None
    return torch.utils.data.dataloader.default_collate(batch)


class SyntheticGraspDataset(torch.utils.data.Dataset):
    """G random in-gripper clouds with the reference's item layout ((3,N) float64, label[, name]);
    a deterministic stand-in for the YCB files, which are not redistributable."""

    def __init__(self, amount, num_points, k, with_obj=False, seed=0):
        self.amount, self.num_points, self.k, self.with_obj, self.seed = amount, num_points, k, with_obj, seed

    def __len__(self):
        return self.amount

    def __getitem__(self, i):
        rng = np.random.default_rng(self.seed * 1000003 + i)
        label = int(rng.integers(0, self.k))
        w = 0.085
        pc = (rng.random((3, self.num_points)) - 0.5) * np.array([[w / 2], [w], [w / 2]])
        pc[0] += 0.004 * label          # a learnable signal: class shifts the cloud along the approach axis
        if self.with_obj:
            return pc, label, f"synthetic_{i % 7}"
        return pc, label


def per_rank_batch(batch_size, world):
    """--batch-size is the GLOBAL batch, as under the reference's ``nn.DataParallel`` (main_1v.py:158-165 scatters one
    batch of --batch-size over the GPUs): each rank loads batch_size / world samples, lr is unchanged."""
    if world <= 1:
        return batch_size
    if batch_size % world:
        raise ValueError(f"--batch-size {batch_size} must be divisible by the number of ranks ({world})")
    return batch_size // world


def _make_loaders(cfg, args, world=1, rank=0, local_rank=0):
    from .model import dataset as ds
    args.rank_batch = per_rank_batch(args.batch_size, world)
    from .hostbudget import workers_per_rank
    args.rank_workers = workers_per_rank(args.num_workers) if world > 1 else args.num_workers
    common = dict(batch_size=args.rank_batch, num_workers=args.rank_workers, pin_memory=True, shuffle=True,
                  worker_init_fn=worker_init_fn, collate_fn=my_collate)
    if args.synthetic:
        tr = SyntheticGraspDataset(args.synthetic, cfg["num_points"], cfg["k"], seed=1)
        te = SyntheticGraspDataset(max(args.synthetic // 4, args.batch_size), cfg["num_points"], cfg["k"],
                                   with_obj=True, seed=2)
    else:
        cls = getattr(ds, cfg["dataset"])
        kw = dict(grasp_points_num=cfg["num_points"], thresh_good=cfg["good"], thresh_bad=cfg["bad"])
        if cfg["fullview"]:
            kw.update(obj_points_num=50000, pc_file_used_num=20)
        tr = cls(tag="train", grasp_amount_per_file=6500, **kw)
        te = cls(tag="test", grasp_amount_per_file=500, with_obj=True, **kw)
    sampler = None
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        sampler = torch.utils.data.distributed.DistributedSampler(tr, shuffle=True)
        common_tr = dict(common, shuffle=False, sampler=sampler)
    else:
        common_tr = common
    if getattr(args, "device_data", False) and args.synthetic:
        raise SystemExit("--device-data keeps the YCB files' clouds resident in HBM and crops on the GPU; --synthetic "
                         "clouds are generated already cropped, per item, on the host: there is nothing to keep resident "
                         "(use one or the other)")
    if getattr(args, "device_data", False) and args.cuda:
        # HBM-resident data layer: under torchrun every rank walks its strided share of the epoch's permutation
        from .device_loader import DeviceGraspLoader
        dev = torch.device("cuda", local_rank if world > 1 else (args.gpu if args.gpu != -1 else 0))
        train_loader = DeviceGraspLoader(tr, args.rank_batch, dev, shuffle=True, seed=args.seed or 0,
                                         max_keep=16384 if cfg["fullview"] else 8192, rank=rank, world=world)
        return train_loader, torch.utils.data.DataLoader(te, **common), train_loader   # set_epoch() like a sampler
    return (torch.utils.data.DataLoader(tr, **common_tr), torch.utils.data.DataLoader(te, **common), sampler)


class _RefPathPickler(pickle._Pickler):
    """Pure-Python pickler (its ``save_global`` is overridable) that names this package's ``model.pointnet`` classes
    by the REFERENCE's module path: no ``__module__`` rewriting, no ``sys.modules`` entry needed at save time, no
    process-global state, thread-safe."""

    def save_global(self, obj, name=None):
        from .model import pointnet as pn
        if isinstance(obj, type) and obj.__module__ == pn.__name__ and "." not in obj.__qualname__:
            # the GLOBAL opcode (valid in every protocol) with the reference's module path
            self.write(pickle.GLOBAL + b"model.pointnet\n" + obj.__qualname__.encode("utf-8") + b"\n")
            self.memoize(obj)
            return
        return super().save_global(obj, name)
    # no ``dispatch`` override: pickle._Pickler.save_type (which special-cases NoneType / NotImplementedType /
    # ellipsis) already ends in ``self.save_global`` for every other class, i.e. in the method above


class _RefPathPickleModule:
    """The ``pickle_module`` handed to ``torch.save`` (which only needs ``Pickler`` and ``__name__``)."""
    __name__ = "pickle"
    Pickler = _RefPathPickler


def save_model(model, path, world=1):
    """Whole-module pickle like the reference's ``torch.save(model, path)`` (main_1v.py:177-178), written so that BOTH
    implementations can load it: the classes are pickled under the reference's module path ``model.pointnet`` (which
    ``install_reference_aliases`` maps to this package on load, and which IS the reference's own module in its
    scripts — kinect2grasp.py / main_test.py); the per-instance cache of folded inference weights is never pickled.
    Single process: a failure raises, as the reference's ``torch.save`` does.  ``world > 1``: only rank 0 saves and an
    exception here would strand the other ranks in their next collective, so the failure is reported and returned
    (False) — ``run()`` broadcasts the flag and every rank aborts together."""
    try:
        torch.save(model, path, pickle_module=_RefPathPickleModule)   # _HipModule.__getstate__ drops the fold plans
        # sidecar: the plain state_dict (tensors only, loads with weights_only=True on any torch and into the reference's
        # own PointNetCls) next to the whole-module pickle the reference's scripts expect (main_1v.py:177-178)
        torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path + ".state_dict")
        return True
    except Exception as e:      # noqa: BLE001
        if world <= 1:
            raise
        print(f"ERROR: could not save {path}: {type(e).__name__}: {e}")
        return False


def run(variant, argv=None):
    cfg = VARIANTS[variant]
    args = build_parser().parse_args(argv)
    args.cuda = args.cuda if torch.cuda.is_available else False      # sic: main_1v.py:35 never calls it
    os.makedirs(args.model_path, exist_ok=True)
    # Debug switch shared with bench.py (never set by a launcher): every rank on cuda:0 over gloo, so that the
    # N-rank control flow of `torchrun --nproc-per-node 8 main_1v_mc.py ...` runs on a 1-GPU box (tests/test_gpu_ddp.py)
    one_gpu_debug = os.environ.get("PNGPD_BENCH_DEBUG_ONE_GPU") == "1"
    rank, world, local_rank = ddp.init_from_env("nccl" if (args.cuda and not one_gpu_debug) else "gloo")
    if one_gpu_debug:
        local_rank = 0
        if args.cuda and world > 1:
            # (debug construct only) the ranks take turns on their first train-mode HIP forward: eight processes loading the
            # training kernels' code objects on ONE device at the same instant occasionally lose a rank (bench.py, HISTORY 9)
            import torch.distributed as _dist
            time.sleep(0.5 * rank)
            _m = PointNetCls(num_points=64, input_chann=3, k=2).cuda().train()
            with torch.no_grad():
                _m(torch.randn(4, 3, 64, device="cuda"))
            torch.cuda.synchronize()
            del _m
            _dist.barrier()
    if args.cuda:
        torch.cuda.manual_seed(1)
    if args.seed is None:
        np.random.seed(int(time.time()))
    else:
        np.random.seed(args.seed); torch.manual_seed(args.seed)
    logger = _ScalarLog(os.path.join(args.log_dir, args.tag)) if rank == 0 else None
    train_loader, test_loader, sampler = _make_loaders(cfg, args, world, rank, local_rank)
    if os.environ.get("PNGPD_HOST_BUDGET_REPORT"):
        # test hook: what this rank took of the node's host budget (tests/test_gpu_ddp.py sums the ranks' reports)
        from . import hostbudget
        with open(os.path.join(os.environ["PNGPD_HOST_BUDGET_REPORT"], f"rank{rank}.json"), "w") as f:
            json.dump({"rank": rank, "loader_workers": args.rank_workers, "eig_threads": hostbudget.threads_per_rank(),
                       "cpus": hostbudget.cpus(), "local_world": hostbudget.local_world()}, f)

    device = torch.device("cpu")
    if args.cuda:
        dev_index = local_rank if world > 1 else (args.gpu if args.gpu != -1 else 0)
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)

    is_resume = 1 if (args.load_model and args.load_epoch != -1) else 0
    if is_resume or args.mode == "test":
        from . import install_reference_aliases
        install_reference_aliases()
        model = torch.load(args.load_model, map_location=device, weights_only=False)
        if isinstance(model, torch.nn.DataParallel):     # a reference checkpoint saved with --gpu -1 (main_1v.py:158-165)
            model = model.module
        print("load model {}".format(args.load_model))
    else:
        model = PointNetCls(num_points=cfg["num_points"], input_chann=3, k=cfg["k"])
    model = model.to(device)
    averager = ddp.GradAverager(model) if world > 1 else None
    if args.cuda:
        if hasattr(model, "set_precision"):
            model.set_precision(args.precision)      # per model (arith.py): stored on the module, pickled with it
        elif args.precision != "fp32":
            raise SystemExit("--precision needs a pointnetgpd_amd model (the loaded module has no set_precision)")
    elif args.precision != "fp32":
        raise SystemExit("--precision needs --cuda (the CPU path is the plain ATen composite)")

    state = {"optimizer": None, "scheduler": None, "graph": None}
    use_graph = bool(args.hip_graph and args.cuda and averager is None)
    if args.hip_graph and not use_graph and rank == 0:
        print("--hip-graph ignored (needs --cuda and a single process)")

    def new_optimizer():
        if args.cuda:
            # the same Adam (main_1v.py:61) over one flat buffer: one launch per step, gradients written in place by
            # the fused backward; for a captured graph the lr lives in a device tensor (StepLR fills it in place)
            from .optim import FlatAdam
            lr = torch.tensor(float(args.lr), device=device) if use_graph else args.lr
            state["optimizer"] = FlatAdam(model.parameters(), lr=lr, capturable=use_graph)
            if averager is not None:
                averager.attach(state["optimizer"])
        else:
            state["optimizer"] = optim.Adam(model.parameters(), lr=args.lr)
        state["scheduler"] = StepLR(state["optimizer"], step_size=30, gamma=0.5)
        state["graph"] = None

    recreate = cfg["recreate_optimizer"] and not args.persistent_optimizer
    if not recreate:
        new_optimizer()

    if args.cuda and hasattr(model, "forward_loss"):
        # HIP path: F.nll_loss and its backward run inside the FC head's own foreign calls (no ATen kernel in a step)
        from .train import loss_backward
        forward_loss = model.forward_loss
    else:
        def forward_loss(data, target, reduction):
            output, trans = model(data)
            return F.nll_loss(output, target, reduction=reduction), output, trans

        def loss_backward(loss):
            loss.backward()

    def train(epoch):
        if recreate:
            new_optimizer()
        optimizer, scheduler = state["optimizer"], state["scheduler"]
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            scheduler.step()                                           # legacy order, main_1v.py:62
        model.train()
        torch.set_grad_enabled(True)
        if sampler is not None:
            sampler.set_epoch(epoch)
        # running accuracy stays on the device: one host read at the end of the epoch instead of the reference's
        # per-batch ``.cpu()`` (main_1v.py:78), which serialises host and device every step
        correct, dataset_size = torch.zeros((), dtype=torch.long, device=device), 0
        loss = None
        t_epoch = time.perf_counter()
        for batch_idx, batch in enumerate(train_loader):
            if args.max_batches and batch_idx >= args.max_batches:
                break
            n_local = 0 if batch is None else int(batch[0].shape[0])
            if n_local:
                data, target = batch[0].float(), batch[1].long().reshape(-1)
                data, target = data.to(device), target.to(device)
            dataset_size += n_local
            runnable = n_local >= 2          # train-mode BatchNorm needs two samples (as nn.BatchNorm1d enforces)
            output = None
            if averager is not None:
                # global per-sample mean (what DataParallel's gathered nll_loss computes): back-propagate the SUM
                # of this rank's losses; the all-reduce adds gradients and kept-sample counts.  A rank whose batch
                # was dropped to < 2 samples sends zeros and still joins the collectives.
                optimizer.zero_grad()
                loss_sum = None
                if runnable:
                    loss_sum, output, _ = forward_loss(data, target, "sum")
                    loss = loss_sum.detach() / n_local
                total = averager.backward(loss_sum, n_local)
                if total is not None:
                    optimizer.step(grad_div=total)
                else:
                    optimizer.step()
            elif not runnable:
                if n_local == 1:
                    raise ValueError("Expected more than 1 value per channel when training, got a batch of 1 "
                                     "(the reference's BatchNorm1d raises here too)")
                continue
            elif use_graph and n_local == args.rank_batch:
                if state["graph"] is None:
                    from .train import GraphedTrainStep
                    state["graph"] = GraphedTrainStep(model, data.shape[0], data.shape[2], optimizer=optimizer)
                loss, output = state["graph"](data, target)
            else:
                optimizer.zero_grad()
                loss, output, _ = forward_loss(data, target, "mean")      # main_1v.py:73-74
                loss_backward(loss)                                       # :75
                optimizer.step()
            if output is not None:
                pred = output.data.max(1, keepdim=True)[1]
                correct += pred.eq(target.view_as(pred)).sum()
            if batch_idx % args.log_interval == 0 and rank == 0 and loss is not None:
                percentage = 100. * batch_idx * args.batch_size / len(train_loader.dataset)   # global samples
                print(f"Train Epoch: {epoch} [{batch_idx * args.batch_size}/{len(train_loader.dataset)} "
                      f"({percentage}%)]\tLoss: {loss.item()}\t{args.tag}")
                logger.add_scalar("train_loss", loss.cpu().item(), batch_idx + epoch * len(train_loader))
        acc = float(correct.item()) / float(max(dataset_size, 1))      # (the .item() also drains the device queue)
        if rank == 0 and logger is not None and dataset_size:
            # training throughput of the epoch, loader included: this rank's samples x ranks / wall time
            logger.add_scalar("train_grasps_per_s", dataset_size * world / max(time.perf_counter() - t_epoch, 1e-9), epoch)
        return acc

    def test():
        if averager is not None:
            averager.sync_buffers()      # rank 0's running statistics everywhere (DataParallel: replica 0 wins)
        model.eval()
        torch.set_grad_enabled(False)
        test_loss, correct, dataset_size, res = 0, 0, 0, []
        for batch_idx, batch in enumerate(test_loader):
            if args.max_batches and batch_idx >= args.max_batches:
                break
            if batch is None:
                continue
            data, target, obj_name = batch
            dataset_size += data.shape[0]
            data, target = data.float().to(device), target.long().reshape(-1).to(device)
            output, _ = model(data)
            test_loss += F.nll_loss(output, target, reduction="sum").cpu().item()   # size_average=False
            pred = output.data.max(1, keepdim=True)[1]
            correct += pred.eq(target.view_as(pred)).long().cpu().sum()
            for i, j, k in zip(obj_name, pred.data.cpu().numpy(), target.data.cpu().numpy()):
                res.append((i, j[0], k))
        test_loss /= len(test_loader.dataset)
        torch.set_grad_enabled(True)
        return float(correct) / float(max(dataset_size, 1)), test_loss

    result = {}
    if args.mode == "train":
        for epoch in range(is_resume * args.load_epoch, args.epoch):
            acc_train = train(epoch)
            if rank == 0:
                print("Train done, acc={}".format(acc_train))
            acc, loss = test()
            saved = True
            if rank == 0:
                print("Test done, acc={}, loss={}".format(acc, loss))
                logger.add_scalar("train_acc", acc_train, epoch)
                logger.add_scalar("test_acc", acc, epoch)
                logger.add_scalar("test_loss", loss, epoch)
                if epoch % args.save_interval == 0:
                    path = os.path.join(args.model_path, args.tag + "_{}.model".format(epoch))
                    saved = save_model(model, path, world)
                    if saved:
                        print("Save model @ {}".format(path))
            if world > 1 and epoch % args.save_interval == 0:
                # a failed checkpoint ends the run on EVERY rank (rank 0 alone raising would hang the others)
                flag = torch.tensor([1.0 if saved else 0.0], device=device)
                torch.distributed.broadcast(flag, src=0)
                if flag.item() == 0.0:
                    raise RuntimeError("checkpoint could not be written on rank 0 (see its log); aborting all ranks")
            result = dict(train_acc=acc_train, test_acc=acc, test_loss=loss, epoch=epoch)
    else:
        print("testing...")
        acc, loss = test()
        print("Test done, acc={}, loss={}".format(acc, loss))
        result = dict(test_acc=acc, test_loss=loss)
    if world > 1:
        torch.distributed.barrier()
    return result
