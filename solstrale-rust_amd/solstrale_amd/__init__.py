"""solstrale_amd -- Python harness over the MI355X path-tracing library (tests/, bench.py).

The product is native: `_build/libsolstrale_hip.so` (hand-written HIP kernels for gfx950 behind the C ABI of
include/solstrale_hip.h) and `_build/libsolstrale_host.so` (C++ mirror of the reference's host surface). This package only
binds them with ctypes; it contains no rendering code and no CPU fallback.
"""
from . import _abi  # noqa: F401
from .host import (AlbedoShader, BloomPostProcessor, CameraConfig, HostError, NopPostProcessor, NormalShader, OidnPostProcessor,  # noqa: F401
                   PathTracingShader, RenderConfig, RotationX, RotationY, RotationZ, Scale, Scene, SceneBuilder, SimpleShader,
                   Translation)
from .device import DeviceError, DeviceScene, background_blocks, comm_unique_id, device_count, record_sizes, world_tree_check  # noqa: F401
