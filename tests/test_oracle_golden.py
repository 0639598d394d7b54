"""CPU: the oracle restatements and the host-side mirror module reproduce the golden vectors
that oracle/make_golden.py recorded from the UNMODIFIED reference."""
import numpy as np
import pytest
import torch

from oracle import pointnet_oracle as po
from tests.helpers import golden_files, build_model, assert_checksums, state_dict_cpu

EVAL = golden_files("pointnet_eval_")
TRAIN = golden_files("pointnet_train_")


def test_fixtures_present():
    assert len(EVAL) >= 5 and len(TRAIN) >= 2


@pytest.mark.parametrize("path", EVAL, ids=lambda p: p.split("pointnet_eval_")[-1][:-4])
def test_eval_golden(path):
    fx = np.load(path)
    m = build_model(fx["num_points"], fx["k"], fx["seed_w"], fx["seed_bn"]).eval()
    assert_checksums(m, fx)          # seeded init of the mirror == the reference's, bit for bit
    sd = state_dict_cpu(m)
    x = torch.from_numpy(fx["x"])
    # (1) torch-functional oracle: same ATen ops as the reference -> tight
    with torch.no_grad():
        logp, trans = po.forward_torch(sd, x, training=False)
    np.testing.assert_allclose(logp.numpy(), fx["logp"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(trans.numpy(), fx["trans"], rtol=0, atol=2e-6)
    # (2) numpy fp64 oracle: independent arithmetic -> fp32 round-off of the reference
    logp64, trans64, inter = po.forward_numpy(sd, fx["x"], training=False, return_intermediates=True)
    np.testing.assert_allclose(logp64, fx["logp"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(trans64, fx["trans"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(inter["stn_pool"], fx["stn_pool"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(inter["feat_pool"], fx["feat_pool"], rtol=1e-4, atol=2e-5)
    assert (logp64.argmax(1) == fx["logp"].argmax(1)).all()
    # (3) the mirror module's CPU (ATen composite) path
    with torch.no_grad():
        lp_m, tr_m = m(x)
    np.testing.assert_allclose(lp_m.numpy(), fx["logp"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(tr_m.numpy(), fx["trans"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("path", TRAIN, ids=lambda p: p.split("pointnet_train_")[-1][:-4])
def test_train_golden(path):
    fx = np.load(path)
    m = build_model(fx["num_points"], fx["k"], fx["seed_w"], fx["seed_bn"]).train()
    assert_checksums(m, fx)
    sd = state_dict_cpu(m)
    x = torch.from_numpy(fx["x"]); y = torch.from_numpy(fx["y"])
    loss, logp, trans, grads, stats = po.train_step_torch(sd, x, y)
    assert abs(loss.item() - float(fx["loss"])) < 2e-6
    np.testing.assert_allclose(logp.numpy(), fx["logp"], atol=5e-6, rtol=0)
    names = [str(n) for n in fx["grad_names"]]
    for i, n in enumerate(names):
        g = grads[n]
        ref_norm = float(fx["grad_norm"][i])
        assert abs(g.double().norm().item() - ref_norm) <= 1e-4 * max(ref_norm, 1e-3), n
        if "grad/" + n in fx:
            np.testing.assert_allclose(g.numpy(), fx["grad/" + n], atol=1e-5 + 1e-4 * ref_norm, rtol=1e-3)
        else:
            np.testing.assert_allclose(g.flatten().numpy()[fx["gradidx/" + n]], fx["gradsample/" + n],
                                       atol=1e-5 + 1e-4 * ref_norm, rtol=1e-3)
    for k in [k for k in fx.files if k.startswith("stat/")]:
        np.testing.assert_allclose(stats[k[5:]].numpy(), fx[k], atol=1e-6, rtol=1e-5)
    # numpy fp64 train-mode forward incl. running-stat update rule (momentum 0.1, unbiased var)
    logp64, trans64, new_stats = po.forward_numpy(sd, fx["x"], training=True, return_new_stats=True)
    np.testing.assert_allclose(logp64, fx["logp"], atol=3e-4, rtol=0)  # fp64 vs the fp32 reference
    for k, v in new_stats.items():
        np.testing.assert_allclose(v, fx["stat/" + k], atol=2e-5, rtol=1e-4)
    # mirror module, CPU train mode: same loss and same running stats
    lp_m, _ = m(x)
    loss_m = torch.nn.functional.nll_loss(lp_m, y)
    assert abs(loss_m.item() - float(fx["loss"])) < 2e-6
    for k in [k for k in fx.files if k.startswith("stat/")]:
        np.testing.assert_allclose(m.state_dict()[k[5:]].numpy(), fx[k], atol=1e-6, rtol=1e-5)


def test_survey_kats():
    """SURVEY.md §8c KAT-A / KAT-B (regenerated from the reference by make_golden.py)."""
    fx = np.load(golden_files("pointnet_kat")[0])
    torch.manual_seed(0)
    m = build_model(750, 2, 0, -1).eval()
    torch.manual_seed(0)
    from pointnetgpd_amd.model.pointnet import PointNetCls
    m = PointNetCls(750, 3, 2).eval()
    x = torch.randn(64, 3, 750)
    with torch.no_grad():
        logp, trans = po.forward_torch(state_dict_cpu(m), x)
    np.testing.assert_allclose(logp.numpy(), fx["a_logp"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(trans[0].numpy(), fx["a_trans0"], atol=2e-6, rtol=0)
    assert abs(logp.double().sum().item() - float(fx["a_logp_sum"])) < 1e-3
    # the survey's printed values
    np.testing.assert_allclose(logp[0].numpy(), [-0.7147452, -0.6720058], atol=1e-6)
    torch.manual_seed(1)
    m2 = PointNetCls(1024, 3, 3).train()
    x2 = torch.randn(16, 3, 1024); y2 = torch.arange(16) % 3
    loss, _, _, grads, stats = po.train_step_torch(state_dict_cpu(m2), x2, y2)
    assert abs(loss.item() - float(fx["b_loss"])) < 2e-6
    assert abs(grads["fc3.weight"].double().norm().item() - float(fx["b_gn_fc3"])) < 1e-4
    assert abs(grads["feat.conv3.weight"].double().norm().item() - float(fx["b_gn_conv3"])) < 1e-3
    assert abs(grads["feat.stn.conv1.weight"].double().norm().item() - float(fx["b_gn_stn_conv1"])) < 1e-3
    np.testing.assert_allclose(stats["feat.bn3.running_mean"][:3].numpy(), fx["b_bn3_rm3"], atol=1e-6)
