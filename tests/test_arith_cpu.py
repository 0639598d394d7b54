"""arith.py on the CPU: per-model settings, propagation to sub-modules, pickling as builtins, legacy shims."""
import io
import pickle

import pytest
import torch

from pointnetgpd_amd import arith, train
from pointnetgpd_amd.model import pointnet as pn


def test_set_precision_is_per_model_and_propagates():
    a, b = pn.PointNetCls(64, 3, 2), pn.PointNetCls(64, 3, 2)
    a.set_precision("bf16x3", refine=True, refine_pool=1)
    for m in (a, a.feat, a.feat.stn):
        r = m.get_precision()
        assert (r.infer, r.train, r.infer_refine, r.refine_pool) == ("bf16x3", "bf16x3", True, 1)
    r = b.get_precision()
    assert (r.infer, r.train, r.infer_refine, r.sequencing) == ("fp32", "fp32", False, "fused")
    a.set_precision(train="bf16")                                   # one aspect only
    assert a.get_precision().train == "bf16" and a.get_precision().infer == "bf16x3"
    a.set_precision("fp32")                                         # refine is meaningless in fp32
    assert not a.get_precision().infer_refine
    with pytest.raises(ValueError):
        a.set_precision("fp16")
    with pytest.raises(ValueError):
        a.set_precision(sequencing="eager")


def test_setting_is_pickled_as_builtins_and_not_in_state_dict():
    a = pn.PointNetCls(64, 3, 3).set_precision("bf16", fp32_side_passes=True)
    assert not any("arith" in k for k in a.state_dict())
    buf = io.BytesIO()
    torch.save(a, buf)
    raw = buf.getvalue()
    assert b"pointnetgpd_amd.arith" not in raw and b"pointnetgpd_amd/arith" not in raw    # loads without this package's arith
    b = torch.load(io.BytesIO(raw), weights_only=False)
    r = b.feat.get_precision()
    assert (r.train, r.infer, r.fp32_side_passes) == ("bf16", "bf16", True)
    assert type(b.__dict__["_arith"]) is dict
    assert all(type(v) in (str, bool, int) for v in b.__dict__["_arith"].values())


def test_legacy_setters_edit_the_default_only():
    m_own = pn.PointNetCls(64, 3, 2).set_precision("fp32")
    m_def = pn.PointNetCls(64, 3, 2)
    try:
        train.set_train_precision("bf16x3", refine_pool=0)
        pn.set_inference_precision("bf16", refine=True)
        train.set_sequencing("passes")
        d = m_def.get_precision()
        assert (d.train, d.infer, d.infer_refine, d.refine_pool, d.sequencing) == ("bf16x3", "bf16", True, 0, "passes")
        o = m_own.get_precision()
        assert (o.train, o.infer, o.infer_refine) == ("fp32", "fp32", False)
        assert pn.get_inference_precision() == "bf16"
    finally:
        train.set_train_precision("fp32", refine_pool=2)
        pn.set_inference_precision("fp32")
        train.set_sequencing("fused")
    assert m_def.get_precision() == arith.Arith("fp32", False, "fp32", False, 2, "fused")
