"""RayTracerProgram::SetupScene (Src/RayTracerProgram.cpp:467-552) through the host-side mirror: the reference's default scene --
four spheres, a capsule, the checkered ground plane and the unitychan mesh -- as data, and a builder that adds it to a scene.

SHAPES entries, in insertion order: (kind, geometry..., material) with material a nested tuple
    ("diffuse", rgb) | ("checker", rgb, size) | ("reflective", rgb, fuzz) | ("emissive", rgb) | ("blend", a, b, factor) | ("combine", a, b)
"""
import numpy as np

from . import api as R

_WHITE = (1.0, 1.0, 1.0)
_GOLD = (0.95, 0.75, 0.1)
_HALF_GOLD = tuple(float(np.float32(v) * np.float32(0.5)) for v in _GOLD)        # RVec3 * 0.5f in float

SHAPES = [
    ("sphere", (1.5, 2.5, -2.0), 0.9, ("blend", ("reflective", _WHITE, 0.0), ("diffuse", (1.0, 0.5, 0.1)), 0.5)),
    ("sphere", (-1.5, -0.5, -3.0), 0.5, ("diffuse", (0.1, 1.0, 0.2))),
    ("sphere", (0.8, -1.5, -1.0), 0.5, ("blend", ("reflective", _WHITE, 0.0), ("diffuse", (0.5, 0.0, 0.2)), 0.5)),
    ("sphere", (2.8, -1.2, -4.0), 1.5, ("combine", ("blend", ("reflective", _GOLD, 0.0), ("diffuse", _GOLD), 0.5), ("emissive", _HALF_GOLD))),
    ("capsule", (-1.5, -1.5, -1.5), (-2.0, -1.5, 0.0), 0.5, ("blend", ("reflective", (0.8, 0.75, 0.6), 0.2), ("diffuse", (0.25, 0.75, 0.6)), 0.2)),
    ("plane", (0.0, 1.0, 0.0), (0.0, -2.0, 0.0), ("blend", ("reflective", _WHITE, 0.1), ("checker", _WHITE, 5.0), 0.5)),
    ("mesh", "unitychan.obj", ("blend", ("reflective", _WHITE, 0.2), ("diffuse", _WHITE), 1.0)),
]


def material(m):
    k = m[0]
    if k == "diffuse":
        return R.SurfaceMaterial_Diffuse(m[1])
    if k == "checker":
        return R.SurfaceMaterial_DiffuseChecker(m[1], m[2])
    if k == "reflective":
        return R.SurfaceMaterial_Reflective(m[1], m[2])
    if k == "emissive":
        return R.SurfaceMaterial_Emissive(m[1])
    if k == "blend":
        return R.SurfaceMaterial_Blend(material(m[1]), material(m[2]), m[3])
    if k == "combine":
        return R.SurfaceMaterial_Combine(material(m[1]), material(m[2]))
    raise ValueError(k)


def SetupScene(scene, mesh_path):
    """scene: a RayTracerScene; mesh_path: the reference's Data/unitychan.obj"""
    for sh in SHAPES:
        if sh[0] == "sphere":
            scene.AddShape(R.RSphere.Create(sh[1], sh[2]), material(sh[3]))
        elif sh[0] == "capsule":
            scene.AddShape(R.RCapsule.Create(sh[1], sh[2], sh[3]), material(sh[4]))
        elif sh[0] == "plane":
            scene.AddShape(R.RPlane.Create(sh[1], sh[2]), material(sh[3]))
        else:
            scene.AddShape(R.RMeshShape.Create(mesh_path), material(sh[2]))
    return scene
