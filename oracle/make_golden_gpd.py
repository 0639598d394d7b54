"""Generate tests/golden/gpd_*.npz by EXECUTING the unmodified reference code of the GPD baseline row
in the build container.  TEST INFRASTRUCTURE ONLY; runs where /root/reference exists, never on the GPU box.

* projection images — ``BaseGraspDataset.project_pc`` / ``cal_projection`` (PointNetGPD/model/dataset.py:88-198) of the
  reference's own ``model.dataset`` module (imported with ``open3d`` stubbed); the one thing that cannot run here,
  ``get_normal`` (open3d's estimate_normals, dataset.py:78-86), is replaced on the INSTANCE by a function returning
  the normals this script supplies — everything downstream of the normals is the reference's code;
* depth registration — ``registerDepthMap`` / ``registeredDepthMapToPointCloud`` (PointNetGPD/ycb_cloud_generate.py:
  60-184): the module imports h5py / imageio at the top, so the two function definitions are cut out of the file
  with ``ast`` and executed unmodified;
* ``GPDClassifier`` (PointNetGPD/model/gpd.py:5-31) imported as is.

Usage:  python oracle/make_golden_gpd.py
"""
import ast
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def reference_dataset_module():
    sys.path.insert(0, os.path.join(REF, "PointNetGPD"))
    os.environ["PointNetGPD_FOLDER"] = REF
    sys.modules.setdefault("open3d", types.ModuleType("open3d"))
    from model import dataset as ref_dataset
    return ref_dataset


def reference_cloudgen_functions():
    path = os.path.join(REF, "PointNetGPD", "ycb_cloud_generate.py")
    tree = ast.parse(open(path).read())
    wanted = {"registerDepthMap", "registeredDepthMapToPointCloud"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert {n.name for n in body} == wanted
    ns = {"np": np}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns["registerDepthMap"], ns["registeredDepthMapToPointCloud"]


# ---------------------------------------------------------------------------------------------------------
def projection_cases(ref_dataset):
    rng = np.random.default_rng(31)
    cases = []

    def hand_cloud(m, w, clusters=0, flat=False):
        pts = rng.uniform(-1, 1, size=(m, 3)) * np.array([w / 4, w / 2, w / 4])
        if clusters:                                  # > 50 points inside single voxels (voxel_point_num cap)
            c = rng.uniform(-1, 1, size=(clusters, 3)) * np.array([w / 5, w / 3, w / 5])
            extra = c[rng.integers(0, clusters, size=120 * clusters)] + rng.normal(size=(120 * clusters, 3)) * 2e-4
            pts = np.vstack([pts, extra])
            pts = pts[rng.permutation(len(pts))]
        if flat:
            pts[:, 2] = 0.0
        n = rng.normal(size=pts.shape)
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        return pts, n

    for tag, m, w, kw, nan_rows in [("dense", 900, 0.085, dict(clusters=3), 0), ("sparse", 60, 0.085, {}, 0),
                                    ("nan_normals", 400, 0.07, dict(clusters=1), 17), ("flat", 300, 0.085, dict(flat=True), 0),
                                    ("one_point", 1, 0.085, {}, 0), ("narrow", 500, 0.031, {}, 0)]:
        pts, nrm = hand_cloud(m, w, **kw)
        if nan_rows:
            bad = rng.choice(len(pts), nan_rows, replace=False)
            nrm[bad, rng.integers(0, 3, nan_rows)] = np.nan
        # the "whole cloud" of project_pc: the in-box points interleaved with outsiders, selected through in_ind
        P = len(pts) + 50
        pc = rng.uniform(-0.3, 0.3, size=(P, 3))
        normals = rng.normal(size=(P, 3))
        in_ind = np.sort(rng.choice(P, len(pts), replace=False))
        pc[in_ind] = pts
        normals[in_ind] = nrm
        rec = dict(pc=pc, normals=normals, in_ind=in_ind, width=w)
        for chann in (3, 12):
            ds = ref_dataset.BaseGraspDataset.__new__(ref_dataset.BaseGraspDataset)
            ds.project_chann, ds.project_size, ds.voxel_point_num, ds.projection_margin = chann, 60, 50, 1
            ds.in_ind = in_ind
            ds.get_normal = lambda points, _n=normals: _n.copy()       # the only substitution: open3d's normals
            rec[f"out{chann}"] = ds.project_pc(pc, w)
        cases.append((tag, rec))
        print("projection", tag, len(pts), rec["out12"].shape, float(rec["out12"][..., 0].max()))
    np.savez_compressed(os.path.join(OUT, "gpd_projection.npz"),
                        tags=np.array([t for t, _ in cases]),
                        **{f"{t}/{k}": v for t, r in cases for k, v in r.items()})


def cloudgen_cases():
    """Inputs come from tests/synth_gpd.py (regenerated identically by the tests); the small scene's outputs are
    stored in full, the VGA scene's as counts + checksums + a fixed row sample (the full arrays are 10 MB)."""
    from tests import synth_gpd
    register, to_cloud = reference_cloudgen_functions()
    rec = {}
    for tag in synth_gpd.SCENES:
        sc = synth_gpd.cloudgen_scene(tag)
        reg = register(sc["depth"], sc["rgb"], sc["depthK"], sc["rgbK"], sc["H"])
        reg_m = reg.copy(); reg_m[sc["mask"]] = 0
        cloud = to_cloud(reg_m, sc["rgb"], sc["rgbK"], sc["refFromRGB"], sc["objFromref"])[0]
        rows = synth_gpd.sample_rows(len(cloud))
        pix = synth_gpd.sample_rows(reg.size, seed=9)
        rec[f"{tag}/reg_nonzero"] = int((reg > 0).sum())
        rec[f"{tag}/reg_sum"] = float(reg.sum())
        rec[f"{tag}/reg_pix"] = pix; rec[f"{tag}/reg_val"] = reg.reshape(-1)[pix]
        rec[f"{tag}/cloud_len"] = len(cloud)
        rec[f"{tag}/cloud_colsum"] = cloud.sum(0)
        rec[f"{tag}/cloud_rows"] = rows; rec[f"{tag}/cloud_val"] = cloud[rows]
        if tag == "small":
            rec[f"{tag}/registered"] = reg; rec[f"{tag}/cloud"] = cloud
            # organized=True (:130-155): the (h,w,6) grid with NaN xyz / zero colour at the pixels without depth
            rec[f"{tag}/cloud_organized"] = to_cloud(reg_m, sc["rgb"], sc["rgbK"], sc["refFromRGB"], sc["objFromref"],
                                                     organized=True)
        print("cloudgen", tag, "registered nonzero", int((reg > 0).sum()), "cloud points", len(cloud))
    np.savez_compressed(os.path.join(OUT, "gpd_cloudgen.npz"), tags=np.array(list(synth_gpd.SCENES)), **rec)


def classifier_cases():
    sys.path.insert(0, os.path.join(REF, "PointNetGPD"))
    from model.gpd import GPDClassifier
    rec = {}
    for chann in (3, 12):
        torch.manual_seed(100 + chann)
        m = GPDClassifier(chann).eval()
        g = torch.Generator().manual_seed(200 + chann)
        x = torch.rand(5, chann, 60, 60, generator=g) * (torch.rand(5, chann, 60, 60, generator=g) < 0.2)
        with torch.no_grad():
            logp = m(x)
        rec[f"x{chann}"] = x.numpy(); rec[f"logp{chann}"] = logp.numpy()
        sd = m.state_dict()
        rec[f"names{chann}"] = np.array(sorted(sd))
        rec[f"checksums{chann}"] = np.array([[sd[k].double().sum().item(), sd[k].double().abs().sum().item()]
                                             for k in sorted(sd)])
        print("classifier", chann, logp[0].tolist())
    np.savez_compressed(os.path.join(OUT, "gpd_classifier.npz"), **rec)


if __name__ == "__main__":
    projection_cases(reference_dataset_module())
    cloudgen_cases()
    classifier_cases()
