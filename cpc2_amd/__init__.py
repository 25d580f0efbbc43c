"""cpc2_amd -- MI355X-native CPC training hot path.

Keeps the reference's Python boundary (cpc.model.CPCModel / cpc.criterion.CPCUnsupersivedCriterion
of MarvinLvn/CPC2) and runs everything under it as hand-written HIP kernels for gfx950, reached
through the C ABI of ``libcpc2_hip.so`` (see include/cpc2_hip.h).  There is NO CPU fallback: the
modules raise if the library is missing or the tensors are not on a GPU.
"""
from . import _lib  # noqa: F401
from .criterion import CPCUnsupersivedCriterion, NoneCriterion, PredictionNetwork  # noqa: F401
from .model import ChannelNorm, CPCAR, CPCEncoder, CPCModel  # noqa: F401

__version__ = "0.1.0"
