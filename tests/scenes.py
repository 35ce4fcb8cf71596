"""Multi-shape scene descriptions shared by the golden generator, the oracle tests and the GPU parity tests (data only).

A scene is a list of shapes in insertion order:
    ("sphere", center, radius, material) | ("plane", normal, point, material) | ("capsule", start, end, radius, material)
    | ("triangle", p0, p1, p2, material)
    | ("mesh", asset name, material)
with material = a preorder node list for oracle.materials() / None (no material).  All numbers are float32 values."""
import numpy as np

D, DC, RF, EM, BL, CO, NU = range(7)
f32 = np.float32


def diffuse(c=(1, 1, 1)):
    return [(D, c, 0, 0, 0)]


def checker(c=(1, 1, 1), size=5.0):
    return [(DC, c, size, 0, 0)]


def reflective(c=(1, 1, 1), fuzz=0.0):
    return [(RF, c, fuzz, 0, 0)]


def emissive(c):
    return [(EM, c, 0, 0, 0)]


def _pair(kind, a, b, param):
    ia, ib = 1, 1 + len(a)
    shift = lambda nodes, off: [(t, c, p, (x + off) if t in (BL, CO) else x, (y + off) if t in (BL, CO) else y) for t, c, p, x, y in nodes]  # noqa: E731
    return [(kind, (0, 0, 0), param, ia, ib)] + shift(a, ia) + shift(b, ib)


def blend(a, b, factor):
    return _pair(BL, a, b, factor)


def combine(a, b):
    return _pair(CO, a, b, 0)


def half(c):
    return tuple(float(f32(v) * f32(0.5)) for v in c)


def default_scene(fuzz=True):
    """RayTracerProgram::SetupScene (Src/RayTracerProgram.cpp:467-552) as data; fuzz=False zeroes the three fuzziness values
    (the device draws fuzzy directions with double-precision transcendentals, so only the fuzz-free variant is compared with the
    reference's frames bit for bit on the GPU)."""
    fz = (lambda v: v) if fuzz else (lambda v: 0.0)
    return [
        ("sphere", (1.5, 2.5, -2.0), 0.9, blend(reflective(), diffuse((1.0, 0.5, 0.1)), 0.5)),
        ("sphere", (-1.5, -0.5, -3.0), 0.5, diffuse((0.1, 1.0, 0.2))),
        ("sphere", (0.8, -1.5, -1.0), 0.5, blend(reflective(), diffuse((0.5, 0.0, 0.2)), 0.5)),
        ("sphere", (2.8, -1.2, -4.0), 1.5, combine(blend(reflective((0.95, 0.75, 0.1)), diffuse((0.95, 0.75, 0.1)), 0.5),
                                                   emissive(half((0.95, 0.75, 0.1))))),
        ("capsule", (-1.5, -1.5, -1.5), (-2.0, -1.5, 0.0), 0.5, blend(reflective((0.8, 0.75, 0.6), fz(0.2)), diffuse((0.25, 0.75, 0.6)), 0.2)),
        ("plane", (0.0, 1.0, 0.0), (0.0, -2.0, 0.0), blend(reflective((1, 1, 1), fz(0.1)), checker(), 0.5)),
        ("mesh", "unitychan", blend(reflective((1, 1, 1), fz(0.2)), diffuse((1.0, 1.0, 1.0)), 1.0)),
    ]


def quirk_scene():
    """A textured mesh FIRST, analytic shapes after it: a sphere / plane / capsule side that hits nearer than the mesh did keeps the
    mesh hit's sampled colour and alpha (one RayHitResult serves all shapes, Src/RayTracerScene.cpp:99-125; Src/RRay.cpp:53-58,75-80),
    a capsule end resets them (Src/Shapes.cpp:34-62).  Shapes overlap the mesh on screen from in front."""
    return [
        ("mesh", "unitychan", diffuse((1, 1, 1))),
        ("sphere", (0.3, 0.9, 1.5), 0.45, diffuse((0.9, 0.9, 0.9))),
        ("capsule", (-0.9, -0.6, 1.2), (0.5, -0.2, 1.6), 0.3, diffuse((0.8, 1.0, 0.8))),
        ("plane", (0.0, 0.0, 1.0), (0.0, 0.0, 0.4), blend(reflective((0.9, 0.9, 1.0)), diffuse((1.0, 0.9, 0.8)), 0.5)),
        ("sphere", (-0.5, 0.2, 2.5), 0.35, None),
    ]


def shapes_scene():
    """Analytic shapes only, including rays that start inside shapes, a tilted plane with a non-unit normal and a degenerate capsule."""
    return [
        ("plane", (0.1, 1.0, 0.05), (0.0, -1.5, 0.0), checker((0.9, 0.9, 0.9), 1.0)),
        ("sphere", (0.0, 0.0, 0.0), 1.0, reflective((0.9, 0.9, 0.9))),
        ("sphere", (0.0, 0.0, 7.0), 0.25, diffuse((1.0, 0.2, 0.2))),          # the camera sits inside this one
        ("capsule", (-2.0, -1.0, 0.0), (2.0, 1.5, -1.0), 0.4, blend(reflective(), diffuse((0.3, 0.5, 1.0)), 0.5)),
        ("capsule", (1.5, -1.0, 1.0), (1.5, -1.0, 1.0), 0.3, diffuse((1.0, 1.0, 0.3))),   # Start == End: only the end spheres can hit
        ("sphere", (-1.8, 1.2, 1.0), 0.6, combine(diffuse((0.2, 0.8, 0.2)), emissive((0.3, 0.1, 0.1)))),
    ]


def room_scene():
    """SetupScene's shapes inside the room its source keeps commented out (ceiling light sphere, ceiling, back / front / right / left
    walls): 13 shapes, planes after the mesh (those are tested by the trace waves), an emissive light, a plane behind the camera."""
    room = [
        ("sphere", (0.0, 5.0, 0.0), 0.5, emissive((5.0, 2.0, 6.0))),
        ("plane", (0.0, -1.0, 0.0), (0.0, 5.0, 0.0), diffuse((1.2, 1.2, 1.5))),
        ("plane", (0.0, 0.0, 1.0), (0.0, 0.0, -5.0), checker()),
        ("plane", (0.0, 0.0, -1.0), (0.0, 0.0, 10.0), checker()),
        ("plane", (1.0, 0.0, 0.0), (-5.0, 0.0, 0.0), checker()),
        ("plane", (-1.0, 0.0, 0.0), (5.0, 0.0, 0.0), checker()),
    ]
    d = default_scene(False)
    return d[:6] + room[:3] + [("mesh", "TorusKnot", diffuse((0.9, 0.9, 0.9)))] + room[3:] + d[6:7]


def tris_scene():
    """RTriangle shapes (Src/Shapes.h:106-130: single-sided, face normal, culled by their own box -- an axis-aligned one has a
    zero-thickness box the reference's slab test never passes for a ray with a component on that axis, so it is invisible),
    with a sphere, the ground and a textured mesh before one of them (its hits inherit the mesh's texel)."""
    return [
        ("triangle", (-2.5, -1.0, -1.0), (0.5, -1.2, 0.5), (-1.0, 2.0, -0.5), blend(reflective((0.9, 0.9, 0.9)), diffuse((0.9, 0.4, 0.2)), 0.5)),
        ("triangle", (0.5, -1.0, 1.0), (2.5, -1.0, 1.0), (1.5, 1.5, 1.0), diffuse((0.2, 0.9, 0.3))),        # z = 1 plane: invisible
        ("mesh", "unitychan", diffuse((1, 1, 1))),
        ("triangle", (-0.8, -0.5, 1.5), (0.8, -0.6, 1.2), (0.1, 1.0, 1.6), diffuse((0.8, 0.8, 1.0))),       # in front of the mesh
        ("triangle", (0.8, -0.5, 1.5), (-0.8, -0.6, 1.2), (0.1, 1.0, 1.6), diffuse((1.0, 0.2, 0.2))),       # wound away from the camera
        ("sphere", (1.8, 0.0, 0.0), 0.7, reflective((0.9, 0.9, 0.9))),
        ("plane", (0.0, 1.0, 0.0), (0.0, -2.0, 0.0), checker((0.9, 0.9, 0.9), 2.0)),
        ("triangle", (2.0, 1.0, 0.0), (2.0, 1.0, 0.0), (3.0, 2.0, 0.0), diffuse((1, 1, 1))),                # degenerate
    ]


SCENES = {"tris": tris_scene, "room": room_scene, "default": lambda: default_scene(True), "default_nofuzz": lambda: default_scene(False),
          "quirk": quirk_scene, "shapes": shapes_scene}


def scene_bounds(scene, mesh_bounds):
    """a box around the finite shapes (for drawing test rays)"""
    lo, hi = np.full(3, 1e30), np.full(3, -1e30)
    for s in scene:
        if s[0] == "sphere":
            lo = np.minimum(lo, np.array(s[1]) - s[2]); hi = np.maximum(hi, np.array(s[1]) + s[2])
        elif s[0] == "capsule":
            for c in (s[1], s[2]):
                lo = np.minimum(lo, np.array(c) - s[3]); hi = np.maximum(hi, np.array(c) + s[3])
        elif s[0] == "triangle":
            for c in s[1:4]:
                lo = np.minimum(lo, np.array(c)); hi = np.maximum(hi, np.array(c))
        elif s[0] == "mesh":
            b = mesh_bounds[s[1]]
            lo = np.minimum(lo, b[:3]); hi = np.maximum(hi, b[3:])
    return np.concatenate([lo, hi]).astype(np.float32)
