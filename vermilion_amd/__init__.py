"""vermilion_amd — MI355X-native hot path of the Vermilion path tracer.

The product is ``libvermilion_hip.so`` (hand-written gfx950 kernels behind the
C ABI of ``include/vermilion_hip.h``).  This package is the host-side mirror of
the reference's Integrator plugin surface plus synthetic scene generators.
Importing `vermilion_amd` does not need a GPU; creating a Scene does.
"""
from . import _lib
from ._lib import (VMX_SAMPLING_CORRECTED, VMX_SAMPLING_ELIDE_DEAD, VMX_SAMPLING_LIBM_DOUBLE, VMX_SAMPLING_PARITY, VmxError)
from .scene import (MultiScene, Scene, default_spheres, local_row_indices, local_rows, make_camera, make_opts,
                    spheres_array)
from .api import (Camera, Integrator, MeshEngine, PathTracer, RenderEngine, cameraSettings, float3,
                  pixelValue, vermRenderMode)
from . import scenes

__all__ = [
    "Scene", "MultiScene", "make_camera", "make_opts", "spheres_array", "default_spheres", "local_rows",
    "local_row_indices", "Camera", "Integrator", "MeshEngine", "PathTracer", "RenderEngine",
    "cameraSettings", "float3", "pixelValue", "vermRenderMode", "scenes", "VmxError",
    "VMX_SAMPLING_PARITY", "VMX_SAMPLING_CORRECTED", "VMX_SAMPLING_LIBM_DOUBLE", "VMX_SAMPLING_ELIDE_DEAD",
]
