"""ferrum-infer-rs_amd — MI355X (gfx950) native decode hot path for ferrum-infer-rs.

The product is `lib/libferrum_hip.so` (hand-written HIP kernels + C++ runner behind the C ABI in
`include/ferrum_hip.h`).  This Python package is the host-side harness used by tests and
bench.py: a ctypes mirror of the reference's `Backend` operator traits (`backend.py`) and of the
`ModelExecutor` unified-decode contract (`executor.py`).  It never computes on the CPU: every op
raises if the HIP library is missing.

The directory name is not an importable identifier; load it with
`__graft_entry__.load_package()` (registers it as module `ferrum_infer_rs_amd`).
"""
from .backend import HipBackend, Context, GptqLinear, ExpertStack, load_library, build_library, LIB_PATH  # noqa: F401
from .executor import HipModel, ModelConfig, BatchItem, Checkpoint, Comm  # noqa: F401
