"""lcr-net_amd — MI355X-native (gfx950) implementation of LCR-Net's per-scan hot path.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every op on the path is a
hand-written HIP kernel in ``liblcr_hip.so`` behind the C ABI of ``include/lcr_hip.h``, bound with ctypes in
``_lib``.  The operator / model API mirrors the reference (``experiments/lcrnet/modules/ops``, ``data.py``,
``model_family``) so it is a drop-in for that path.  There is NO CPU fallback: ops raise if the HIP library or a
GPU is missing.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401
