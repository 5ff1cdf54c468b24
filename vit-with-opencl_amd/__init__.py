"""MI355X-native ViT inference path behind the ViT_opencl() call surface.

The product is the C-ABI shared library `libvit_hip.so` built from `csrc/`
(host code in C, kernels in HIP for gfx950).  `host/` holds the ctypes plumbing
that tests and bench.py use to reach it.  The directory name is not a valid
Python identifier; load it with `__graft_entry__.load_package()`.
"""
from .host import binding  # noqa: F401
from .host.binding import (  # noqa: F401
    DeviceBuffer, ViTHip, ViTHipMulti, VitConfig, VitHipError, build_library, lib, preset, shard_range,
    synth_images, synth_weights,
)
