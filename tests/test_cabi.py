"""CPU: libpngpd.so loads, exports every symbol include/pngpd.h declares, and the ctypes
binding table mirrors the header.  No compute calls (no GPU here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pngpd.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pngpd_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_functions():
    fns = header_functions()
    assert "pngpd_trunk_fwd_infer" in fns and "pngpd_fc_fwd" in fns and len(fns) >= 6


def test_library_exports_header_symbols():
    from pointnetgpd_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build libpngpd.so first (__graft_entry__.build())"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (pngpd_[a-z0-9_]+)", out))
    missing = [f for f in header_functions() if f not in exported]
    assert not missing, f"header declares but library lacks: {missing}"


def test_ctypes_table_matches_header():
    from pointnetgpd_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()


def header_arity():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for name, params in re.findall(r"\b(pngpd_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        params = params.strip()
        out[name] = 0 if params in ("", "void") else params.count(",") + 1
    return out


def test_ctypes_arity_matches_header():
    """Every ctypes argtypes list has exactly as many entries as the C prototype has parameters."""
    from pointnetgpd_amd import _lib
    ar = header_arity()
    bad = {n: (len(a), ar.get(n)) for n, (_, a) in _lib.SIGNATURES.items() if len(a) != ar.get(n)}
    assert not bad, bad


def test_library_loads_and_reports_abi():
    from pointnetgpd_amd import _lib
    lib = _lib.load()
    assert lib.pngpd_abi_version() == _lib.ABI_VERSION
    assert lib.pngpd_strerror(0) == b"ok"
    assert lib.pngpd_strerror(1) == b"invalid argument"
    # argument validation happens before any launch, so it is testable without a GPU
    assert lib.pngpd_trunk_fwd_infer(None, 1, 1, None, None, None, None, None, None, None, 0, 0, None, None, 0, None) == 1
    assert lib.pngpd_fc_fwd(None, 1, 8, None, None, 1, 0, None, None) == 1
    # splits are pure functions of their arguments (no process-global tuning state) and size the workspace
    assert lib.pngpd_trunk_infer_splits(4, 100, 0) == 2                    # ceil(100/64) tiles bound the default
    assert lib.pngpd_trunk_infer_splits(1024, 1024, 0) == 1 and lib.pngpd_trunk_infer_splits(1024, 1024, 2048) == 2
    assert lib.pngpd_trunk_workspace_bytes(4, 100, 2) == 4 * 2 * 1024 * 4
    assert lib.pngpd_trunk_workspace_bytes(4, 100, 0) == 4 * 2 * 1024 * 4   # 0 -> the default rule
    assert lib.pngpd_trunk_workspace_bytes(4, 100, 1) == 0                  # one workgroup per cloud: no scratch
    assert lib.pngpd_trunk_splits(1024, 1024, 0) == 1 and lib.pngpd_trunk_splits(16, 1024, 0) == 16
    assert lib.pngpd_trunk_splits(320, 64, 0) == 1 and lib.pngpd_trunk_splits(256, 128, 2048) == 2
    assert lib.pngpd_trunk_g2t_bytes(3, 100) == 3 * 2 * 64 * 128 * 4


def test_cuda_ops_refuse_cpu_tensors():
    """The product path has no CPU fallback behind the HIP ops."""
    import torch
    from pointnetgpd_amd import ops
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.fc_fwd(torch.zeros(2, 8), torch.zeros(3, 8), torch.zeros(3), ops.EPI_NONE)


def test_missing_library_fails_loudly(monkeypatch):
    from pointnetgpd_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_load_error", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpngpd.so")
    with pytest.raises(RuntimeError, match="libpngpd.so not found"):
        _lib.load()


def test_argument_validation_of_every_family():
    """Each entry family rejects bad arguments with a status code BEFORE any launch (testable without a GPU).
    Non-null dummy addresses are never dereferenced on these paths."""
    import ctypes
    from pointnetgpd_amd import _lib
    lib = _lib.load()
    p = 0x1000                                             # dummy non-null "device pointer"
    INV, UNSUP = 1, 3
    # sampler
    assert lib.pngpd_gpg_normal_moments(None, 0, p, 10, p, 1, 0.1, 100, p, p, None) == INV
    assert lib.pngpd_gpg_normal_moments(p, 0, p, 10, p, 1, -1.0, 100, p, p, None) == INV          # radius <= 0
    assert lib.pngpd_hand_box_counts(p, 0, 10, p, 5, p, 2, p, None) == UNSUP                       # 1 or 4 boxes only
    assert lib.pngpd_hand_box_counts(p, 0, 0, p, 5, p, 4, p, None) == INV                          # empty cloud
    assert lib.pngpd_hand_box_counts_indexed(p, 0, 130, p, 2, p, 5, p, 4, p, None) == INV          # C != ceil(P/64)
    assert lib.pngpd_hand_box_counts_indexed(p, 0, 130, p, 3, p, 5, p, 3, p, None) == UNSUP
    # crop
    assert lib.pngpd_crop_count_compact(p, 0, 10, p, 0, 16, p, p, None) == INV                     # G == 0
    assert lib.pngpd_crop_resample(p, 0, 10, p, None, None, 0, 1, p, p, 16, 8, 2, 20, 0, 0, None, None, p, p, None) == INV   # mode
    assert lib.pngpd_crop_resample(p, 0, 10, p, None, p, 0, 1, p, p, 16, 8, 1, 20, 0, 0, None, None, p, p, None) == INV      # gather, Pg 0
    assert lib.pngpd_crop_resample(p, 0, 10, p, None, None, 0, 1, p, p, 1 << 20, 8, 1, 20, 0, 0, None, None, p, p, None) == UNSUP  # LDS
    # training passes
    assert lib.pngpd_trunk_bn2_stats(p, 4, 100, None, p, p, p, p, p, 0, p, None, None) == INV         # S < 1
    assert lib.pngpd_trunk_bn2_stats(p, 4, 100, None, p, p, p, p, p, 3, p, None, None) == INV         # S > ceil(N/64)
    assert lib.pngpd_trunk_fwd_train(p, 4, 100, None, p, p, p, p, p, p, p, p, 3, p, p, p, p, None, None) == INV
    assert lib.pngpd_trunk_bwd_d(p, 4, 100, None, *([p] * 14), None, 3, p, p, p, None) == INV
    assert lib.pngpd_trunk_bwd_e(p, 4, 100, None, *([p] * 14), p, 0, p, p, p, None) == INV
    assert lib.pngpd_reduce_partials4(*([None, 0, 0, 0, None] * 4), None) == INV                   # no segment
    assert lib.pngpd_reduce_partials4(p, 1, 0, 8, p, *([None, 0, 0, 0, None] * 3), None) == INV    # R == 0
    assert lib.pngpd_fold_conv_bn(p, p, None, None, None, None, 1e-5, 30, 8, 1, p, p, None) == INV      # MFMA_B needs C % 32 == 0
    fm = _lib.FoldModel()
    assert lib.pngpd_fold_model(None, None) == INV
    assert lib.pngpd_fold_model(ctypes.addressof(fm), None) == INV                                 # n == 0
    fm.n = 1
    fm.layer[0].W, fm.layer[0].bf, fm.layer[0].C, fm.layer[0].K = 64, 64, 30, 8
    assert lib.pngpd_fold_model(ctypes.addressof(fm), None) == INV                                 # no output requested
    fm.layer[0].mfma = 64
    assert lib.pngpd_fold_model(ctypes.addressof(fm), None) == INV                                 # MFMA_B needs C % 32 == 0
    fm.layer[0].mfma, fm.layer[0].x3, fm.layer[0].C = None, 64, 32
    assert lib.pngpd_fold_model(ctypes.addressof(fm), None) == INV                                 # split-bf16 needs K % 16 == 0
    assert lib.pngpd_fc_fwd(p, 4, 10, p, p, 3, 0, p, None) == INV                                  # K % 4 != 0
    assert lib.pngpd_fc_fwd(p, 4, 16, p, p, 40, 3, p, None) == INV                                 # log_softmax needs Nout <= 32
    assert lib.pngpd_strerror(UNSUP) == b"unsupported configuration"


def _header_struct_fields(name):
    """Field names of ``typedef struct <name> { ... } <name>_t;`` in declaration order."""
    src = open(HEADER).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s_t;" % (name, name), src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        # "const float *w1, *b1, *g1, *be1" / "int B, N" / "void *save" / "size_t save_bytes" / "long long *nbt1"
        decl = re.sub(r"\[[^\]]*\]", "", decl)          # array extents are not field names
        first, *rest = decl.split(",")
        fields.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", first)[-1])
        fields += [re.findall(r"[A-Za-z_][A-Za-z0-9_]*", r)[-1] for r in rest]
    return fields


@pytest.mark.parametrize("cname,which,cls", [("pngpd_trunk_train", 0, "TrunkTrainArgs"), ("pngpd_head_train", 1, "HeadTrainArgs"),
                                             ("pngpd_fold_layer", None, "FoldLayer"), ("pngpd_fold_model", 2, "FoldModel")])
def test_argument_structs_mirror_the_header(cname, which, cls):
    """The ctypes.Structure mirrors of the two training-entry argument structs: same fields in the same order as the
    header, and the same size as the compiled library's sizeof (no compute call)."""
    import ctypes
    from pointnetgpd_amd import _lib
    st = getattr(_lib, cls)
    assert [f[0] for f in st._fields_] == _header_struct_fields(cname)
    if which is not None:
        assert _lib.load().pngpd_struct_bytes(which) == ctypes.sizeof(st)


def test_training_entries_validate_arguments():
    """No GPU here: the fused entries must reject NULL / inconsistent descriptors before touching the device."""
    import ctypes
    from pointnetgpd_amd import _lib
    lib = _lib.load()
    a = _lib.TrunkTrainArgs()
    assert lib.pngpd_trunk_train_save_bytes(ctypes.addressof(a)) == 0          # B = N = 0
    a.B, a.N, a.S = 4, 100, 1
    assert lib.pngpd_trunk_train_save_bytes(ctypes.addressof(a)) > 0
    assert lib.pngpd_trunk_train_scratch_bytes(ctypes.addressof(a)) > 0
    assert lib.pngpd_trunk_train_fwd(ctypes.addressof(a), None) == 1           # PNGPD_ERR_INVALID_ARG: NULL pointers
    a.S = 3                                                                     # more splits than tiles (T = 2)
    assert lib.pngpd_trunk_train_save_bytes(ctypes.addressof(a)) == 0
    h = _lib.HeadTrainArgs()
    assert lib.pngpd_head_train_fwd(ctypes.addressof(h), None) == 1
    assert lib.pngpd_adam_flat(None, None, None, None, 16, 0.005, None, 0.9, 0.999, 1e-8, 1.0, None, 1.0, None, None) == 1
    for B, N in [(1, 500), (64, 750), (1024, 1024)]:
        T = (N + 63) // 64
        assert 1 <= lib.pngpd_trunk_splits(B, N, 0) <= T and 1 <= lib.pngpd_trunk_infer_splits(B, N, 0) <= T
    assert lib.pngpd_trunk_splits(64, 750, 0) == 4 and lib.pngpd_trunk_infer_splits(64, 750, 0) == 4


def test_probe_entry_validates_arguments():
    """pngpd_probe_mfma_rate rejects bad descriptors before touching the device (no GPU here)."""
    from pointnetgpd_amd import _lib
    lib = _lib.load()
    for args in [(2, 1, 10, 1, None, None), (0, 3, 10, 1, None, None), (0, 1, 0, 1, None, None), (1, 2, 10, None, None, None)]:
        assert lib.pngpd_probe_mfma_rate(*args) == 1
