"""First-maximum semantics of training pass C (``torch.max`` / ``MaxPool1d`` return the FIRST maximal index — the
gradient of the reference's ``torch.max(x, 2)`` flows to that point, pointnet.py:33,148) under exact ties.

The pass finds the arg-max without compares (``lane_max_moments``, pngpd_common.h: v_max3 tree, then the smallest key
``(bits(m - v) & ~63) | row``), merges the two row halves of a tile through v_permlane32_swap and the tiles through an
LDS running maximum.  Clouds built from REPEATED points make every maximum a tie between bit-identical values, at
exactly the distance that exercises one of those three levels; the arg-max must always be the earlier copy."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _operands(dev, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    w1, b1 = r(64, 3), r(64) * 0.1
    s1c, t1c = (torch.rand(64, generator=g) + 0.5).to(dev), r(64) * 0.1
    w2, w3 = r(128, 64) / 8, r(1024, 128) / 11
    s2c, t2c = (torch.rand(128, generator=g) + 0.5).to(dev), r(128) * 0.1
    return w1, b1, s1c, t1c, w2, w3, s2c, t2c


def _run(x, nterms, seed=3):
    from pointnetgpd_amd import ops
    dev = x.device
    w1, b1, s1c, t1c, w2, w3, s2c, t2c = _operands(dev, seed)
    B, _, N = x.shape
    if nterms == 0:
        w2p, w3p = ops.pack_mfma_b(w2), ops.pack_mfma_b(w3)
        _, z2t = ops.trunk_bn2_stats(x, None, w1, b1, s1c, t1c, w2p, 1)
        pmax, parg, _, _ = ops.trunk_fwd_train(x, None, w1, b1, s1c, t1c, w2p, s2c, t2c, w3p, 1, z2t)
    else:
        w2x, w3x = ops.split_pack_bf16(w2), ops.split_pack_bf16(w3)
        _, z2t = ops.trunk_bn2_stats_bf(x, None, w1, b1, s1c, t1c, w2x, 1, nterms)
        pmax, parg = ops.trunk_fwd_train_bf(x, None, w1, b1, s1c, t1c, w2x, s2c, t2c, w3x, 1, nterms, z2t)[:2]
    torch.cuda.synchronize()
    return pmax[:, 0], parg[:, 0].long()


@pytest.mark.parametrize("nterms", [0, 3, 1])
@pytest.mark.parametrize("period", [1, 4, 32, 64, 128, 512])
def test_first_copy_wins(period, nterms, cuda_device):
    """x = blocks of `period` distinct points, each block stored twice in a row: point p and p + period are identical.
    period 1: neighbouring rows of one lane; 4: the two row halves of a wave (v_permlane32_swap merge); 32: the lane's
    two accumulator blocks; 64 / 128: consecutive tiles (LDS running maximum, strict >); 512: the cloud's two halves."""
    B, N = 6, 1024
    g = torch.Generator().manual_seed(100 + period)
    base = torch.rand(B, 3, N // 2, generator=g) * 2 - 1
    blocks = base.view(B, 3, -1, period)                       # (B,3,nb,period)
    x = torch.cat([blocks, blocks], dim=3).reshape(B, 3, N).contiguous().to(cuda_device)
    pmax, parg = _run(x, nterms)
    assert int(parg.min()) >= 0 and int(parg.max()) < N
    assert torch.all((parg // period) % 2 == 0), f"a later copy won {(parg // period % 2 != 0).sum().item()} ties"
    # and it is a maximum: the same kernel on the cloud without the copies finds the same value
    pmax1, _ = _run(torch.cat([blocks, blocks.flip(3)], dim=3).reshape(B, 3, N).contiguous().to(cuda_device), nterms)
    assert torch.equal(pmax, pmax1)


def test_ragged_tail_and_all_equal(cuda_device):
    """A cloud of ONE repeated point (every value of a channel ties, in every tile, incl. the ragged last tile of
    N = 750): the arg-max is point 0 for every channel."""
    B, N = 3, 750
    x = torch.rand(B, 3, 1).repeat(1, 1, N).contiguous().to(cuda_device)
    for nterms in (0, 3):
        _, parg = _run(x, nterms)
        assert torch.all(parg == 0)
