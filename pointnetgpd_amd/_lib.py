"""ctypes binding of libpngpd.so (the C ABI declared in include/pngpd.h).

The library is built in-tree (``pointnetgpd_amd/libpngpd.so``) by ``__graft_entry__.build()``
or ``make -C pointnetgpd_amd/csrc``.  There is NO fallback: any CUDA-tensor op raises
``RuntimeError`` if the library is missing or fails to load.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PNGPD_LIB") or os.path.join(_HERE, "libpngpd.so")   # PNGPD_LIB: A/B builds only
ABI_VERSION = 7

_lib = None
_load_error = None

c_f32p = ctypes.c_void_p
c_void = ctypes.c_void_p

_P, _I, _F, _Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t


class TrunkTrainArgs(ctypes.Structure):
    """``pngpd_trunk_train_t`` of include/pngpd.h (field for field)."""
    _fields_ = ([("x", _P), ("trans", _P), ("B", _I), ("N", _I), ("S", _I), ("relu_last", _I), ("precision", _I),
                 ("fp32_side", _I), ("refine", _I), ("need_bwd", _I), ("eps", _F), ("momentum", _F)] +
                [(n, _P) for n in ("w1", "b1", "g1", "be1", "w2", "b2", "g2", "be2", "w3", "b3", "g3", "be3",
                                   "rm1", "rv1", "nbt1", "rm2", "rv2", "nbt2", "rm3", "rv3", "nbt3",
                                   "pooled", "idx", "zhat", "dp",
                                   "dW1", "db1", "dg1", "dbe1", "dW2", "db2", "dg2", "dbe2", "dW3", "db3", "dg3", "dbe3",
                                   "dT")] +
                [("save", _P), ("save_bytes", _Z), ("scratch", _P), ("scratch_bytes", _Z)])


class HeadTrainArgs(ctypes.Structure):
    """``pngpd_head_train_t`` of include/pngpd.h (field for field)."""
    _fields_ = ([("inp", _P), ("B", _I), ("K0", _I), ("H1", _I), ("H2", _I), ("k", _I), ("epilogue", _I),
                 ("eps", _F), ("momentum", _F)] +
                [(n, _P) for n in ("W1", "b1", "g1", "be1", "W2", "b2", "g2", "be2", "W3", "b3",
                                   "rm1", "rv1", "nbt1", "rm2", "rv2", "nbt2", "out", "gout", "dinp",
                                   "dW1", "db1", "dg1", "dbe1", "dW2", "db2", "dg2", "dbe2", "dW3", "db3")] +
                [("save", _P), ("save_bytes", _Z), ("scratch", _P), ("scratch_bytes", _Z),
                 ("target", _P), ("loss", _P), ("gloss", _P), ("loss_mean", _I)])


FOLD_MAX_LAYERS = 16


class FoldLayer(ctypes.Structure):
    """``pngpd_fold_layer_t`` of include/pngpd.h."""
    _fields_ = ([(n, _P) for n in ("W", "b", "gamma", "beta", "mean", "var")] + [("eps", _F), ("C", _I), ("K", _I)] +
                [(n, _P) for n in ("row", "mfma", "x3", "bf")])


class FoldModel(ctypes.Structure):
    """``pngpd_fold_model_t`` of include/pngpd.h."""
    _fields_ = [("n", _I), ("layer", FoldLayer * FOLD_MAX_LAYERS)]


# name -> (restype, argtypes); mirrors include/pngpd.h one-to-one (checked by tests).
SIGNATURES = {
    "pngpd_abi_version": (ctypes.c_int, []),
    "pngpd_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "pngpd_fold_conv_bn": (ctypes.c_int, [c_f32p] * 6 + [ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                                       ctypes.c_int, c_f32p, c_f32p, c_void]),
    "pngpd_fold_model": (ctypes.c_int, [c_void, c_void]),
    "pngpd_trunk_infer_splits": (ctypes.c_int, [ctypes.c_int] * 3),
    "pngpd_trunk_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "pngpd_trunk_fwd_infer": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, c_f32p] + [c_f32p] * 6 +
                              [ctypes.c_int, ctypes.c_int, c_f32p, c_void, ctypes.c_size_t, c_void]),
    "pngpd_fc_fwd": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, ctypes.c_int,
                                    ctypes.c_int, c_f32p, c_void]),
    "pngpd_split_pack_bf16": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, c_void]),
    "pngpd_trunk_infer_bf_splits": (ctypes.c_int, [ctypes.c_int] * 3),
    "pngpd_trunk_fwd_infer_bf": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p] + [c_f32p] * 6 +
                                 [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, c_void, ctypes.c_size_t, c_void]),
    "pngpd_trunk_fwd_infer_bf_arg": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p] +
                                     [c_f32p] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, c_void, c_void,
                                                     ctypes.c_size_t, c_void]),
    "pngpd_trunk_fwd_train_bf": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int] + [c_f32p] * 9 +
                                 [ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_void]),
    # ---- training passes (S = workgroups per cloud is an explicit argument everywhere)
    "pngpd_trunk_splits": (ctypes.c_int, [ctypes.c_int] * 3),
    "pngpd_trunk_g2t_bytes": (ctypes.c_size_t, [ctypes.c_int] * 2),
    "pngpd_cloud_moments": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, c_f32p, c_void]),
    "pngpd_trunk_bn2_stats": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int] + [c_f32p] * 6 + [ctypes.c_int,
                                                                                                  c_f32p, c_f32p, c_void]),
    "pngpd_trunk_fwd_train": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int] + [c_f32p] * 9 + [ctypes.c_int] +
                              [c_f32p] * 5 + [c_void]),
    "pngpd_trunk_bwd_gather": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int] + [c_f32p] * 10 +
                               [ctypes.c_int, c_f32p, c_void]),
    "pngpd_trunk_pool_refine": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int] + [c_f32p] * 12 +
                                [ctypes.c_int, ctypes.c_int, c_f32p, c_void]),
    "pngpd_trunk_bwd_d": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int] + [c_f32p] * 16 + [ctypes.c_int] +
                          [c_f32p] * 3 + [c_void]),
    "pngpd_trunk_bwd_e": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int] + [c_f32p] * 16 + [ctypes.c_int] +
                          [c_f32p] * 3 + [c_void]),
    # the same passes on bf16 / bf16x3 operands (nterms 1 / 3)
    "pngpd_trunk_bn2_stats_bf": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int] + [c_f32p] * 5 +
                                 [c_void, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, c_void]),
    "pngpd_trunk_bwd_gather_bf": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int] + [c_f32p] * 5 +
                                  [c_void, ctypes.c_int] + [c_f32p] * 4 + [ctypes.c_int, c_f32p, c_void]),
    "pngpd_trunk_bwd_d_bf": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int] + [c_f32p] * 4 +
                             [c_void, ctypes.c_int] + [c_f32p] * 5 + [ctypes.c_int] + [c_f32p] * 3 + [c_void]),
    "pngpd_trunk_bwd_e_bf": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int] + [c_f32p] * 12 +
                             [c_void, ctypes.c_int, c_f32p, c_f32p, ctypes.c_int] + [c_f32p] * 3 + [c_void]),
    "pngpd_fc_bwd": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p,
                                    c_f32p, c_void]),
    "pngpd_bn1d_fwd_train": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, ctypes.c_float,
                                            ctypes.c_int, c_f32p, c_f32p, c_f32p, ctypes.c_float, c_f32p, c_f32p,
                                            c_void, c_void]),
    "pngpd_bn1d_bwd": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, c_f32p,
                                      ctypes.c_float, ctypes.c_int, c_f32p, c_f32p, c_f32p, c_void]),
    "pngpd_log_softmax_bwd": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, c_f32p, c_void]),
    "pngpd_nll_fwd": (ctypes.c_int, [c_f32p, c_void, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, c_void]),
    "pngpd_nll_log_softmax_bwd": (ctypes.c_int, [c_f32p, c_f32p, c_void, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 c_f32p, c_void]),
    # ---- finalize kernels
    "pngpd_bn1_finalize": (ctypes.c_int, [c_void, c_void, ctypes.c_int, ctypes.c_int] + [c_void] * 4 +
                           [ctypes.c_float, ctypes.c_float] + [c_void] * 5 + [c_void]),
    "pngpd_bn2_finalize": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int] + [c_void] * 3 +
                           [ctypes.c_float, ctypes.c_float] + [c_void] * 5 + [c_void]),
    "pngpd_bn3_finalize": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, c_void,
                                          ctypes.c_float] + [c_void] * 4 + [c_void]),
    "pngpd_pool_finalize": (ctypes.c_int, [c_void, c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, c_void,
                                           ctypes.c_float, ctypes.c_int, c_void, c_void, c_void, c_void]),
    "pngpd_bn3_bwd_prep": (ctypes.c_int, [c_void, c_void, c_void, ctypes.c_int, ctypes.c_int, c_void, c_void,
                                          ctypes.c_float, ctypes.c_int] + [c_void] * 4 + [c_void]),
    "pngpd_reduce_partials": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void, c_void]),
    "pngpd_reduce_partials4": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void] * 4 + [c_void]),
    "pngpd_a_cvec_finalize": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int] + [c_void] * 4 + [ctypes.c_float] +
                              [c_void] * 2 + [c_void]),
    "pngpd_dw3_finalize": (ctypes.c_int, [c_void, c_void, c_void, ctypes.c_int, ctypes.c_int] + [c_void] * 4 +
                           [ctypes.c_float] + [c_void] + [c_void]),
    "pngpd_bwd_e_prep": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, ctypes.c_float,
                                        c_void, c_void, c_void, c_void]),
    "pngpd_dw1_finalize": (ctypes.c_int, [c_void, c_void, c_void, ctypes.c_int, ctypes.c_int] + [c_void] * 5 +
                           [ctypes.c_float] + [c_void] * 4 + [c_void]),
    # ---- one entry per direction of the training graph + flat Adam
    "pngpd_struct_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "pngpd_probe_mfma_rate": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void, c_void, c_void]),
    "pngpd_trunk_train_save_bytes": (ctypes.c_size_t, [c_void]),
    "pngpd_trunk_train_scratch_bytes": (ctypes.c_size_t, [c_void]),
    "pngpd_trunk_train_fwd": (ctypes.c_int, [c_void, c_void]),
    "pngpd_trunk_train_bwd": (ctypes.c_int, [c_void, c_void]),
    "pngpd_head_train_save_bytes": (ctypes.c_size_t, [c_void]),
    "pngpd_head_train_scratch_bytes": (ctypes.c_size_t, [c_void]),
    "pngpd_head_train_fwd": (ctypes.c_int, [c_void, c_void]),
    "pngpd_head_train_bwd": (ctypes.c_int, [c_void, c_void]),
    "pngpd_adam_flat": (ctypes.c_int, [c_void] * 4 + [ctypes.c_longlong, ctypes.c_float, c_void, ctypes.c_float,
                                                      ctypes.c_float, ctypes.c_float, ctypes.c_float, c_void,
                                                      ctypes.c_float, c_void, c_void]),
    "pngpd_adam_step_inc": (ctypes.c_int, [c_void, c_void]),
    # ---- crop / resample
    "pngpd_crop_count_compact": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, ctypes.c_int,
                                                ctypes.c_int, c_void, c_void, c_void]),
    "pngpd_crop_count_compact_ranges": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, ctypes.c_int,
                                                       ctypes.c_int, c_void, c_void, c_void]),
    "pngpd_crop_count_compact_gather": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, ctypes.c_int,
                                                       ctypes.c_int, ctypes.c_int, c_void, c_void, c_void]),
    "pngpd_crop_resample": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, c_void, ctypes.c_int,
                                           ctypes.c_int, c_void, c_void, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_ulonglong, ctypes.c_longlong, c_void, c_void, c_void,
                                           c_void, c_void]),
    "pngpd_crop_count_compact_indexed": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, ctypes.c_int, c_void,
                                                        ctypes.c_int, ctypes.c_int, c_void, c_void, c_void]),
    "pngpd_crop_indexed": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, ctypes.c_int, c_void, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong,
                                          ctypes.c_longlong, c_void, c_void, c_void, c_void]),
    "pngpd_batch_keep_rows": (ctypes.c_int, [c_void, c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, c_void, c_void]),
    "pngpd_stack_gather_lists": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong,
                                                ctypes.c_longlong, c_void, c_void]),
    "pngpd_train_batch": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, c_void, c_void, ctypes.c_int,
                                         ctypes.c_int, c_void, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_ulonglong, ctypes.c_longlong] + [c_void] * 9),
    # ---- GPG sampler (device half)
    "pngpd_gpg_normal_moments": (ctypes.c_int, [c_void, ctypes.c_int, c_void, ctypes.c_int, c_void, ctypes.c_int,
                                                ctypes.c_double, ctypes.c_int, c_void, c_void, c_void]),
    "pngpd_gpg_normal_moments_indexed": (ctypes.c_int, [c_void, ctypes.c_int, c_void, c_void, ctypes.c_int, c_void,
                                                        ctypes.c_int, c_void, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                                        c_void, c_void, c_void]),
    "pngpd_hand_box_counts": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, ctypes.c_int, c_void,
                                             ctypes.c_int, c_void, c_void]),
    "pngpd_hand_box_counts_indexed_n": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, ctypes.c_int, c_void,
                                                       ctypes.c_int, c_void, ctypes.c_int, c_void, ctypes.c_int, c_void,
                                                       c_void]),
    "pngpd_gpg_frames": (ctypes.c_int, [c_void, c_void, c_void, ctypes.c_int, c_void, c_void, c_void]),
    "pngpd_gpg_enumerate": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void, c_void, c_void, c_void]),
    "pngpd_gpg_select": (ctypes.c_int, [c_void, c_void, c_void, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void, c_void,
                                        c_void, c_void, c_void, c_void]),
    "pngpd_gpg_sweep_select": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, ctypes.c_int, c_void, c_void,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void, c_void, ctypes.c_double,
                                              c_void, c_void, c_void, c_void, c_void, c_void, c_void]),
    "pngpd_gpg_pushin_sweep": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, ctypes.c_int, c_void, c_void,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void, ctypes.c_int,
                                              ctypes.c_double, c_void, c_void, c_void, c_void]),
    "pngpd_gpg_pushin": (ctypes.c_int, [c_void] * 6 + [ctypes.c_int] * 4 + [c_void] * 4 + [c_void]),
    "pngpd_gpg_finish": (ctypes.c_int, [c_void] * 7 + [ctypes.c_int] * 4 + [c_void] * 5 + [c_void]),
    # ---- GPD baseline + depth registration
    "pngpd_gpd_projection": (ctypes.c_int, [c_void, c_void, c_void, c_void, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, c_void, c_void]),
    "pngpd_depth_register": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, ctypes.c_int, ctypes.c_int,
                                            c_void, c_void]),
    "pngpd_depth_cloud_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "pngpd_depth_to_cloud": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, c_void, c_void, c_void,
                                            c_void, ctypes.c_size_t, c_void]),
    "pngpd_conv5_pool2": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, ctypes.c_int,
                                         c_f32p, c_void]),
    "pngpd_conv5_pool2_arg": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, ctypes.c_int,
                                             c_f32p, c_void, c_void]),
    "pngpd_conv5_pool2_bwd_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "pngpd_conv5_pool2_bwd": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_int, c_f32p,
                                             c_void, c_f32p, c_f32p, c_f32p, c_void, ctypes.c_size_t, c_void]),
    "pngpd_relu_bwd": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_longlong, c_void]),
    "pngpd_fc_fwd_splitk_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "pngpd_fc_fwd_splitk": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int,
                                           c_f32p, c_void, ctypes.c_size_t, c_void]),
    "pngpd_hand_box_counts_indexed": (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_void, ctypes.c_int, c_void,
                                                     ctypes.c_int, c_void, ctypes.c_int, c_void, c_void]),
}


def load():
    """Return the loaded library (cached).  Raises RuntimeError — never falls back."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is not None:
        raise RuntimeError(_load_error)
    # torch must be imported first so that libamdhip64.so.7 resolves to the HIP runtime
    # torch already loaded (one runtime per process; device pointers are shared with torch).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        _load_error = (f"libpngpd.so not found at {LIB_PATH}: build it with "
                       f"`python -c 'import __graft_entry__ as g; g.build()'` "
                       f"(hipcc --offload-arch=gfx950).  There is no CPU/PyTorch fallback for CUDA tensors.")
        raise RuntimeError(_load_error)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        _load_error = f"failed to load {LIB_PATH}: {e}"
        raise RuntimeError(_load_error)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.pngpd_abi_version()
    if got != ABI_VERSION:
        _load_error = f"libpngpd ABI version {got} != expected {ABI_VERSION}; rebuild the library"
        raise RuntimeError(_load_error)
    _lib = lib
    return lib


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NULL = _NullCtx()


def device_guard(device):
    """``torch.cuda.device(device)`` only when ``device`` is not already current: the context manager costs two
    hipSetDevice round trips per kernel launch, which is most of a launch's host time at B = 1."""
    import torch
    return _NULL if torch.cuda.current_device() == device.index else torch.cuda.device(device)


def check(code, what):
    if code != 0:
        msg = load().pngpd_strerror(code).decode()
        raise RuntimeError(f"libpngpd: {what} failed: {msg} (code {code})")
