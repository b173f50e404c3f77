"""pyposegraphbuilder -- MI355X-native pairwise relative-pose engine.

The name is the reference's (empty) Python package (src/pyposegraphbuilder/__init__.py:1).
All compute runs in libpgi.so (HIP, gfx950); there is no CPU fallback.
"""
from . import synthetic  # noqa: F401
from ._lib import EDGE_DTYPE, PgiError, default_params  # noqa: F401


def __getattr__(name):  # torch-dependent parts are imported lazily
    if name in ("Engine",):
        from .engine import Engine
        return Engine
    if name in ("PoseGraphBuilder", "findEssentialMatrix", "bindProcessToDeviceNode"):
        from . import builder
        return getattr(builder, name)
    raise AttributeError(name)
