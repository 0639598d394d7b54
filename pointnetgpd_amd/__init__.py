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


def install_sampler_alias():
    """Route ``from dexnet.grasping import GpgGraspSamplerPcl`` (dex-net/apps/kinect2grasp.py:33) to the GPU sampler
    of ``pointnetgpd_amd.gpg``: patches the real ``dexnet.grasping`` module when dex-net is installed, otherwise
    registers a minimal ``dexnet`` / ``dexnet.grasping`` pair holding only that class.  Opt-in; returns the class."""
    import importlib
    import sys
    import types
    from .gpg import GpgGraspSamplerPcl
    try:
        grasping = importlib.import_module("dexnet.grasping")
    except Exception:
        dexnet = sys.modules.get("dexnet") or types.ModuleType("dexnet")
        grasping = types.ModuleType("dexnet.grasping")
        dexnet.grasping = grasping
        sys.modules["dexnet"] = dexnet
        sys.modules["dexnet.grasping"] = grasping
    grasping.GpgGraspSamplerPcl = GpgGraspSamplerPcl
    return GpgGraspSamplerPcl
