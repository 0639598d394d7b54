"""CPU oracle for the PointNetGPD grasp-evaluation hot path (PointNetCls).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` may be imported by the
product package ``pointnetgpd_amd``; only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg use it, and only as the checker / the
CPU baseline, never as the thing measured or shipped.

What it restates (citations are relative to /root/reference/):

* ``STN3d.forward``        PointNetGPD/model/pointnet.py:27-45
* ``PointNetfeat.forward`` PointNetGPD/model/pointnet.py:137-154
* ``PointNetCls.forward``  PointNetGPD/model/pointnet.py:189-194
* ``nn.BatchNorm1d`` semantics used at pointnet.py:21-25,130-132,185-186
* the train step ``nll_loss`` + backward of PointNetGPD/main_1v.py:72-76
* ``test_network``         PointNetGPD/main_test.py:59-69

Two independent restatements live here:

1. ``forward_numpy``  – plain numpy (matmul / mean / var / max), eval AND train
   mode, any float dtype.  No torch ops.
2. ``forward_torch``  – a *functional* restatement on the same ATen ops the
   reference dispatches (conv1d, batch_norm, max_pool1d, linear, bmm,
   log_softmax) in the same order, so autograd through it is the gradient
   oracle and timing it on the host cores is the "reference CPU PyTorch path"
   baseline (``cpu_baseline.kind == "port"``).

Parity pin: both are checked against the UNMODIFIED reference module imported
from /root/reference by ``oracle/make_golden.py`` (run in the build container,
where the reference exists); the resulting vectors are committed under
``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py`` everywhere.
The reference's own test-suite holds no golden vector for this path
(SURVEY.md §8c), so these generated vectors are the pin.

All state is passed in as a ``state_dict``-style mapping whose keys are the
reference's parameter / buffer names (``feat.stn.conv1.weight`` …).
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-5        # nn.BatchNorm1d default, pointnet.py:21
BN_MOMENTUM = 0.1    # nn.BatchNorm1d default


# --------------------------------------------------------------------------
# numpy restatement
# --------------------------------------------------------------------------
def _np(sd, key, dtype):
    v = sd[key]
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v).astype(dtype, copy=False)


def _bn_numpy(z, sd, prefix, training, dtype, reduce_axes, new_stats):
    """BatchNorm1d.  z: (B,C,N) with reduce_axes=(0,2) or (B,C) with (0,)."""
    g = _np(sd, prefix + ".weight", dtype)
    b = _np(sd, prefix + ".bias", dtype)
    if training:
        mean = z.mean(axis=reduce_axes)
        var = z.var(axis=reduce_axes)  # biased, used for normalisation
        if new_stats is not None:
            m = 1
            for a in reduce_axes:
                m *= z.shape[a]
            unbiased = var * (m / max(m - 1, 1))
            rm = _np(sd, prefix + ".running_mean", dtype)
            rv = _np(sd, prefix + ".running_var", dtype)
            new_stats[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
            new_stats[prefix + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * unbiased
    else:
        mean = _np(sd, prefix + ".running_mean", dtype)
        var = _np(sd, prefix + ".running_var", dtype)
    shape = [1, -1] + [1] * (z.ndim - 2)
    inv = 1.0 / np.sqrt(var + BN_EPS)
    return (z - mean.reshape(shape)) * (inv * g).reshape(shape) + b.reshape(shape)


def _conv1x1_numpy(x, sd, prefix, dtype):
    """Conv1d(kernel=1): (B,Cin,N) -> (B,Cout,N).  pointnet.py:12-14."""
    w = _np(sd, prefix + ".weight", dtype)[:, :, 0]
    b = _np(sd, prefix + ".bias", dtype)
    return np.einsum("oc,bcn->bon", w, x) + b[None, :, None]


def _linear_numpy(x, sd, prefix, dtype):
    w = _np(sd, prefix + ".weight", dtype)
    b = _np(sd, prefix + ".bias", dtype)
    return x @ w.T + b[None, :]


def _trunk_numpy(x, sd, prefix, training, dtype, relu_last, new_stats):
    """3->64->128->1024 per-point MLP + max over N.

    STN trunk (relu_last=True):  pointnet.py:29-33
    feat trunk (relu_last=False): pointnet.py:144-149
    """
    relu = lambda t: np.maximum(t, 0)
    h = relu(_bn_numpy(_conv1x1_numpy(x, sd, prefix + "conv1", dtype), sd, prefix + "bn1",
                       training, dtype, (0, 2), new_stats))
    h = relu(_bn_numpy(_conv1x1_numpy(h, sd, prefix + "conv2", dtype), sd, prefix + "bn2",
                       training, dtype, (0, 2), new_stats))
    h = _bn_numpy(_conv1x1_numpy(h, sd, prefix + "conv3", dtype), sd, prefix + "bn3",
                  training, dtype, (0, 2), new_stats)
    if relu_last:
        h = relu(h)
    return h.max(axis=2)  # MaxPool1d(num_points) with N == num_points, then view(-1,1024)


def forward_numpy(sd, x, training=False, dtype=np.float64, return_new_stats=False,
                  return_intermediates=False):
    """PointNetCls.forward restated in numpy.  x: (B,3,N).

    Returns (log_probs (B,k), trans (B,3,3)) [+ dict of updated running stats]
    [+ dict of intermediates: stn_pool, trans, feat_pool, fc3 logits].
    """
    x = np.asarray(x).astype(dtype)
    new_stats = {} if (training and return_new_stats) else None
    relu = lambda t: np.maximum(t, 0)
    # --- STN3d  pointnet.py:27-45
    g = _trunk_numpy(x, sd, "feat.stn.", training, dtype, True, new_stats)
    stn_pool = g
    g = relu(_bn_numpy(_linear_numpy(g, sd, "feat.stn.fc1", dtype), sd, "feat.stn.bn4",
                       training, dtype, (0,), new_stats))
    g = relu(_bn_numpy(_linear_numpy(g, sd, "feat.stn.fc2", dtype), sd, "feat.stn.bn5",
                       training, dtype, (0,), new_stats))
    g = _linear_numpy(g, sd, "feat.stn.fc3", dtype)
    trans = (g + np.eye(3, dtype=dtype).reshape(1, 9)).reshape(-1, 3, 3)
    # --- PointNetfeat  pointnet.py:140-149 : x' = (x^T @ trans)^T = trans^T @ x
    xt = np.einsum("bin,bij->bjn", x, trans)
    f = _trunk_numpy(xt, sd, "feat.", training, dtype, False, new_stats)
    feat_pool = f
    # --- head  pointnet.py:191-194
    f = relu(_bn_numpy(_linear_numpy(f, sd, "fc1", dtype), sd, "bn1", training, dtype, (0,), new_stats))
    f = relu(_bn_numpy(_linear_numpy(f, sd, "fc2", dtype), sd, "bn2", training, dtype, (0,), new_stats))
    logits = _linear_numpy(f, sd, "fc3", dtype)
    zmax = logits.max(axis=1, keepdims=True)
    logp = logits - zmax - np.log(np.exp(logits - zmax).sum(axis=1, keepdims=True))
    out = [logp, trans]
    if return_new_stats:
        out.append(new_stats)
    if return_intermediates:
        out.append({"stn_pool": stn_pool, "trans": trans, "feat_pool": feat_pool, "logits": logits})
    return tuple(out)


# --------------------------------------------------------------------------
# torch functional restatement (same ATen ops / order as the reference)
# --------------------------------------------------------------------------
def _bn_torch(F, z, sd, prefix, training):
    return F.batch_norm(z, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"],
                        training, BN_MOMENTUM, BN_EPS)


CONV_AS_MATMUL = False   # see _conv1x1_torch


def _conv1x1_torch(F, x, w, b):
    """Conv1d(kernel_size=1) (pointnet.py:12-14).  Default: the reference's own ATen op (F.conv1d).  With
    CONV_AS_MATMUL the same contraction is written as ``W[:, :, 0] @ x + b`` — used ONLY when a test runs this
    oracle in fp64 on the GPU, where the conv backend (MIOpen) has no double kernels; the arithmetic
    (sum over input channels, then bias) is identical."""
    if CONV_AS_MATMUL:
        import torch
        return torch.matmul(w[:, :, 0], x) + b[None, :, None]
    return F.conv1d(x, w, b)


def _trunk_torch(F, x, sd, prefix, training, relu_last, choice=None):
    """``choice`` (see ``forward_torch``): None = the reference's ops; else ``(idx (B,1024) int64, keep (B,1024)
    bool or None)`` — the max over N is evaluated AT the given points (a gather) and, for the STN trunk, the ReLU
    ahead of the max keeps exactly the pooled entries ``keep`` marks."""
    x = F.relu(_bn_torch(F, _conv1x1_torch(F, x, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"]),
                         sd, prefix + "bn1", training))
    x = F.relu(_bn_torch(F, _conv1x1_torch(F, x, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"]),
                         sd, prefix + "bn2", training))
    x = _bn_torch(F, _conv1x1_torch(F, x, sd[prefix + "conv3.weight"], sd[prefix + "conv3.bias"]),
                  sd, prefix + "bn3", training)
    if choice is not None:
        import torch
        idx, keep = choice
        g = torch.gather(x, 2, idx.view(idx.shape[0], 1024, 1)).view(-1, 1024)
        if relu_last:
            g = g * keep.to(g.dtype)
        return g
    if relu_last:
        x = F.relu(x)
    n = x.shape[2]
    return F.max_pool1d(x, n).view(-1, 1024)


def _relu_choice(F, y, keep):
    """ReLU, or — with a recorded activation pattern ``keep`` — the linear map that pattern selects."""
    return F.relu(y) if keep is None else y * keep.to(y.dtype)


def forward_torch(sd, x, training=False, choices=None):
    """PointNetCls.forward on the reference's ATen op sequence (pointnet.py:27-45,
    137-154, 189-194).  ``sd`` maps reference names -> tensors (running stats are
    updated IN PLACE when training, exactly like nn.BatchNorm1d).

    ``choices`` (tests only; default None = the reference's own ops): the DISCRETE decisions of another run of the
    same step — ``{"stn_idx", "stn_keep", "feat_idx", "fc_keep": [4 masks: STN fc1, STN fc2, head fc1, head fc2]}``.
    With them imposed the network is a smooth function of its inputs (max -> gather at the recorded arg-max points,
    the five ReLUs that act on single values per sample -> their recorded patterns), so two correct implementations
    agree to rounding error and a gradient comparison needs no allowance for arg-max / ReLU flips at fp32 near-ties.
    The per-point ReLUs of trunk layers 1-2 stay the reference's: one flipped (point, channel) entry in 10^8 moves
    nothing measurable."""
    import torch
    import torch.nn.functional as F
    b = x.shape[0]
    c = choices or {}
    fk = c.get("fc_keep", [None] * 4)
    g = _trunk_torch(F, x, sd, "feat.stn.", training, True,
                     (c["stn_idx"], c["stn_keep"]) if choices else None)
    g = _relu_choice(F, _bn_torch(F, F.linear(g, sd["feat.stn.fc1.weight"], sd["feat.stn.fc1.bias"]),
                                  sd, "feat.stn.bn4", training), fk[0])
    g = _relu_choice(F, _bn_torch(F, F.linear(g, sd["feat.stn.fc2.weight"], sd["feat.stn.fc2.bias"]),
                                  sd, "feat.stn.bn5", training), fk[1])
    g = F.linear(g, sd["feat.stn.fc3.weight"], sd["feat.stn.fc3.bias"])
    iden = torch.eye(3, dtype=x.dtype, device=x.device).view(1, 9).repeat(b, 1)
    trans = (g + iden).view(-1, 3, 3)
    xt = torch.bmm(x.transpose(2, 1), trans).transpose(2, 1)
    f = _trunk_torch(F, xt, sd, "feat.", training, False, (c["feat_idx"], None) if choices else None)
    f = _relu_choice(F, _bn_torch(F, F.linear(f, sd["fc1.weight"], sd["fc1.bias"]), sd, "bn1", training), fk[2])
    f = _relu_choice(F, _bn_torch(F, F.linear(f, sd["fc2.weight"], sd["fc2.bias"]), sd, "bn2", training), fk[3])
    logits = F.linear(f, sd["fc3.weight"], sd["fc3.bias"])
    return F.log_softmax(logits, dim=-1), trans


def head_stack_torch(inp, P, tail, training=True, keep=None):
    """One FC stack of the reference: ``fc1 -> BatchNorm1d -> ReLU -> fc2 -> BatchNorm1d -> ReLU -> fc3`` followed by
    ``+ eye(3)`` (STN3d, pointnet.py:35-43; ``tail="iden"``) or ``log_softmax`` (PointNetCls, pointnet.py:191-194;
    ``tail="log_softmax"``) or nothing.  ``P`` maps ``W1 b1 g1 be1 rm1 rv1 W2 b2 g2 be2 rm2 rv2 W3 b3`` to tensors
    (running statistics updated in place when training, like nn.BatchNorm1d).  The gradient oracle of the FC-stack
    training kernels is autograd through this function in fp64.  Returns (out, [y1, y2]) — the two hidden activations
    are returned so a test can assert that no pre-activation sits near its ReLU threshold."""
    import torch
    import torch.nn.functional as F
    keep = keep or [None, None]
    z1 = F.linear(inp, P["W1"], P["b1"])
    a1 = F.batch_norm(z1, P["rm1"], P["rv1"], P["g1"], P["be1"], training, BN_MOMENTUM, BN_EPS)
    y1 = _relu_choice(F, a1, keep[0])
    z2 = F.linear(y1, P["W2"], P["b2"])
    a2 = F.batch_norm(z2, P["rm2"], P["rv2"], P["g2"], P["be2"], training, BN_MOMENTUM, BN_EPS)
    y2 = _relu_choice(F, a2, keep[1])
    out = F.linear(y2, P["W3"], P["b3"])
    if tail == "iden":
        out = out + torch.eye(3, dtype=out.dtype, device=out.device).view(1, 9)
    elif tail == "log_softmax":
        out = F.log_softmax(out, dim=-1)
    return out, [a1, a2]


def train_step_torch(sd, x, target, dtype=None, choices=None):
    """One forward + nll_loss + backward (main_1v.py:72-75) through
    ``forward_torch``.  Returns (loss, logp, trans, grads{name: tensor},
    new_running_stats{name: tensor}).  ``sd`` is not modified.  ``choices``: see ``forward_torch``."""
    import torch
    import torch.nn.functional as F
    work = {}
    params = []
    for k, v in sd.items():
        v = v.detach().clone()
        if dtype is not None and v.is_floating_point():
            v = v.to(dtype)
        if v.is_floating_point() and not ("running_" in k):
            v.requires_grad_(True)
            params.append(k)
        work[k] = v
    xx = x.detach().clone()
    if dtype is not None:
        xx = xx.to(dtype)
    logp, trans = forward_torch(work, xx, training=True, choices=choices)
    loss = F.nll_loss(logp, target)
    loss.backward()
    grads = {k: work[k].grad.detach() if work[k].grad is not None else torch.zeros_like(work[k])
             for k in params}
    stats = {k: v.detach() for k, v in work.items() if "running_" in k}
    return loss.detach(), logp.detach(), trans.detach(), grads, stats


def test_network_oracle(sd, local_pc):
    """main_test.py:59-69 — B=1 inference on an (N,3) in-gripper cloud:
    returns (pred (1,), probs ndarray (1,k)); softmax(log_softmax) == softmax."""
    import torch
    with torch.no_grad():
        pc = torch.from_numpy(np.ascontiguousarray(np.asarray(local_pc).T[np.newaxis, ...])).float()
        out, _ = forward_torch(sd, pc, training=False)
        out = out.softmax(1)
        pred = out.max(1, keepdim=True)[1]
    return pred[0], out.numpy()


# --------------------------------------------------------------------------
# deterministic synthetic model state: default init under a
# seed, then non-trivial BN affine + running statistics.
# --------------------------------------------------------------------------
PARAM_SHAPES_TRUNK = [("conv1", (64, 3, 1)), ("conv2", (128, 64, 1)), ("conv3", (1024, 128, 1))]


def randomize_bn_(sd, seed=4321, negative_gamma_frac=0.1):
    """In-place: running_mean~N(0,0.1), running_var~U(0.5,1.5), gamma~U(0.5,1.5)
    with ~10% negative gammas on every *bn3* (exercises the min-tracking path of
    the fused max-pool), beta~N(0,0.1)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    for k in sorted(sd.keys()):
        v = sd[k].data
        if k.endswith("running_mean"):
            v.copy_(torch.randn(v.shape, generator=g) * 0.1)
        elif k.endswith("running_var"):
            v.copy_(torch.rand(v.shape, generator=g) + 0.5)
        elif ".bn" in k or k.startswith("bn"):
            if k.endswith(".weight"):
                w = torch.rand(v.shape, generator=g) + 0.5
                if "bn3" in k:
                    flip = torch.rand(v.shape, generator=g) < negative_gamma_frac
                    w = torch.where(flip, -w, w)
                v.copy_(w)
            elif k.endswith(".bias"):
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
    return sd
