"""adyolo_amd: MI355X-native (gfx950) hot path of sadPororo/AD-YOLO.

Layout:
  csrc/            hand-written HIP kernels + the C ABI (include/adyolo_hip.h) -> libadyolo_hip.so
  _lib.py, ops.py  ctypes binding / tensor-level wrappers (PyTorch = device memory + streams only)
  functional.py    block-granular autograd nodes built from those kernels
  wrapper.py, models/   host-side mirror of the reference plugin surface (wrapper.py, models/*)
  features.py      K1 front end (raw 4-channel audio -> 7-channel features on the GPU)
  datasets.py      label encoder / collate / synthetic workload (host logic)
  train.py, dist.py     train step (flat-buffer fused Adam) and RCCL data parallelism
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
