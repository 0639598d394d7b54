"""Per-model arithmetic of the HIP path: which matrix arithmetic the trunks contract in, whether the reduced-precision
pooled values are re-evaluated in fp32, and how a training step is sequenced.

The setting lives ON the module (``model.set_precision(...)``, stored as a plain dict of builtins in the module's
``__dict__['_arith']``), so two models in one process — or two threads driving two models, as the reference's
``nn.DataParallel`` does (main_1v.py:158-165) — can differ, and a whole-module pickle (main_1v.py:177-178) carries its
arithmetic; the dict holds no class of this package, so the pickle still loads in the unmodified reference (its
``nn.Module.__setstate__`` just keeps the extra ``__dict__`` entry).  SURVEY.md §8(b): "no global mutable state".

Fields (all optional; an absent field falls back to the process default below):
    infer             "fp32" | "bf16x3" | "bf16"      eval-mode trunks
    infer_refine      bool                            eval: re-evaluate the pooled maxima in exact fp32
    train             "fp32" | "bf16x3" | "bf16"      train-mode passes
    fp32_side_passes  bool                            train: keep passes B / gather / D / E on the exact fp32 kernels
    refine_pool       0 | 1 | 2                       train, reduced precision: off / matrix pipe / VALU
    sequencing        "fused" | "passes"              one C call per direction, or pass by pass from Python

The module-level setters of rounds 1-5 (``model.pointnet.set_inference_precision``, ``train.set_train_precision``,
``train.set_sequencing``) remain as shims that edit the PROCESS DEFAULT — what a model without its own setting uses.
"""
MODES = ("fp32", "bf16x3", "bf16")
NTERMS = {"fp32": 0, "bf16x3": 3, "bf16": 1}

_DEFAULT = {"infer": "fp32", "infer_refine": False, "train": "fp32", "fp32_side_passes": False, "refine_pool": 2,
            "sequencing": "fused"}


def _check(field, value):
    if field in ("infer", "train"):
        if value not in MODES:
            raise ValueError("precision must be 'fp32', 'bf16x3' or 'bf16'")
        return value
    if field in ("infer_refine", "fp32_side_passes"):
        return bool(value)
    if field == "refine_pool":
        if int(value) not in (0, 1, 2):
            raise ValueError("refine_pool must be 0 (off), 1 (matrix pipe) or 2 (VALU)")
        return int(value)
    if field == "sequencing":
        if value not in ("fused", "passes"):
            raise ValueError("sequencing must be 'fused' or 'passes'")
        return value
    raise KeyError(field)


def set_default(**fields):
    """Edit the process default (legacy setters; tests that sweep modes).  Not thread-safe by nature — per-model
    settings are the supported way to run different arithmetic side by side."""
    for f, v in fields.items():
        if v is not None:
            _DEFAULT[f] = _check(f, v)
    if _DEFAULT["infer"] == "fp32":
        _DEFAULT["infer_refine"] = False


def default(field):
    return _DEFAULT[field]


def set_on(mod, **fields):
    """Store the given fields on ``mod`` (None = leave as is).  Values are validated builtins only."""
    own = dict(mod.__dict__.get("_arith") or {})
    for f, v in fields.items():
        if v is not None:
            own[f] = _check(f, v)
    if own.get("infer", _DEFAULT["infer"]) == "fp32":
        own.pop("infer_refine", None) if "infer" not in own else own.__setitem__("infer_refine", False)
    mod.__dict__["_arith"] = own


def clear_on(mod):
    mod.__dict__.pop("_arith", None)


def resolve(mod):
    """The effective arithmetic of ``mod`` as a hashable tuple record (see ``Arith``)."""
    own = mod.__dict__.get("_arith") if mod is not None else None
    g = _DEFAULT
    if not own:
        return Arith(g["infer"], g["infer_refine"], g["train"], g["fp32_side_passes"], g["refine_pool"],
                     g["sequencing"])
    get = own.get
    infer = get("infer", g["infer"])
    return Arith(infer, bool(get("infer_refine", g["infer_refine"])) and infer != "fp32", get("train", g["train"]),
                 get("fp32_side_passes", g["fp32_side_passes"]), get("refine_pool", g["refine_pool"]),
                 get("sequencing", g["sequencing"]))


class Arith(tuple):
    """(infer, infer_refine, train, fp32_side_passes, refine_pool, sequencing) — immutable, passed by value into the
    autograd Functions so a backward sees the arithmetic its forward ran in."""
    __slots__ = ()

    def __new__(cls, *a):
        return tuple.__new__(cls, a)

    infer = property(lambda s: s[0])
    infer_refine = property(lambda s: s[1])
    train = property(lambda s: s[2])
    fp32_side_passes = property(lambda s: s[3])
    refine_pool = property(lambda s: s[4])
    sequencing = property(lambda s: s[5])
