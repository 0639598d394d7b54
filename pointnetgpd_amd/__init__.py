"""pointnetgpd_amd — MI355X-native implementation of PointNetGPD's grasp-evaluation hot path.

Host side: a mirror of the reference's Python module surface for this path
(``model.pointnet`` / ``model.dataset`` / ``main_*`` loops) over ``libpngpd.so``, a C-ABI
library of hand-written HIP kernels for gfx950 (``include/pngpd.h``).  PyTorch is used for
device memory, streams and ``torch.distributed`` only.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401


def install_reference_aliases():
    """Make ``model.pointnet`` / ``model.dataset`` importable under the reference's module
    paths, so whole-module pickles written by the reference (``torch.save(model, path)``,
    PointNetGPD/main_1v.py:177-178) unpickle onto this implementation."""
    import sys
    from . import model as _model
    from .model import pointnet as _pn
    sys.modules.setdefault("model", _model)
    sys.modules.setdefault("model.pointnet", _pn)
    try:
        from .model import dataset as _ds
        sys.modules.setdefault("model.dataset", _ds)
    except Exception:  # dataset needs PointNetGPD_FOLDER only when instantiated
        pass
