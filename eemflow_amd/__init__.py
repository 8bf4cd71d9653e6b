"""eemflow_amd - MI355X (gfx950) implementation of EEMFlow's dense-flow hot path.

Host code is Python on PyTorch-ROCm and mirrors the reference's interfaces; every kernel lives in
libeemflow_hip.so (hand-written HIP, C ABI in include/eemflow_hip.h)."""
import os as _os

# Frames are kept in flight on separate HIP streams (one context per stream).  The runtime gives a process GPU_MAX_HW_QUEUES hardware
# queues (default 4, the null stream included) and streams that share one serialise, so a fourth stream loses throughput instead of adding
# it; the variable is read at the runtime's first call.  A value the user exported wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

from .eemflow import EEMFlow            # noqa: F401
from .padder import InputPadder         # noqa: F401
from .voxelizer import EventSequence, EventSequenceToVoxelGrid_Pytorch   # noqa: F401
