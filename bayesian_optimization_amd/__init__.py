"""MI355X-native Optimistic Expected Improvement evaluation path.

Layout
------
csrc/        HIP kernels (gfx950) and the C ABI of include/oei_hip.h -> liboei_hip.so
_lib.py      ctypes binding (fails loudly if the library is missing: no CPU fallback)
engine.py    torch-tensor handle around the C ABI (device memory, streams)
methods/     the reference's call signatures (methods/oei.py, methods/sdp.py, methods/bo.py)
multistart.py  sharded multistart evaluation and best-start selection over torch.distributed
"""
from . import _lib  # noqa: F401  (raises if liboei_hip.so is absent)
from .engine import DeviceGP, sdp_batched, solution_derivative_batched, argmin  # noqa: F401
