"""eemflow_amd - MI355X (gfx950) implementation of EEMFlow's dense-flow hot path.

Host code is Python on PyTorch-ROCm and mirrors the reference's interfaces; every kernel lives in
libeemflow_hip.so (hand-written HIP, C ABI in include/eemflow_hip.h)."""
from .eemflow import EEMFlow            # noqa: F401
from .padder import InputPadder         # noqa: F401
from .voxelizer import EventSequence, EventSequenceToVoxelGrid_Pytorch   # noqa: F401
