"""raytracerwin_amd -- MI355X-native drop-in for the per-pixel ray-trace hot path of
aosyang/RayTracerWin.

The product is ``librtwin.so`` (hand-written HIP for gfx950 behind the C ABI of
``include/rtwin.h``).  This package is only the host-side mirror of the reference's
interface for that path, over ctypes: same names, same argument meaning
(``RayTracerScene.AddShape / RayTrace / FindIntersectionWithScene``, ``RMeshShape.Create``, ``RSphere / RPlane / RCapsule``,
``SurfaceMaterial_*``, ``RenderOption``, ``ThreadWorker_Render``).  There is no CPU
fallback: without the built library or without a GPU every device call raises.
"""
from .api import (  # noqa: F401
    Comm, Context, Framebuffer, RayTracerScene, RCapsule, RMeshShape, RPlane, RSphere, RTriangle, RenderOption, RtwError,
    SurfaceMaterial_Blend, SurfaceMaterial_Combine, SurfaceMaterial_Diffuse, SurfaceMaterial_DiffuseChecker,
    SurfaceMaterial_Emissive, SurfaceMaterial_Null, SurfaceMaterial_Reflective,
    ThreadWorker_Render, build_library, library, library_path, material_nodes_from_array, png_load,
)

__all__ = [n for n in dir() if not n.startswith("_")]
