"""ctypes front-end of the CPU oracle (oracle/rt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by bench.py's cpu_baseline leg and by
__graft_entry__.smoke().  The product package (raytracerwin_amd) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liboracle.so")
REF_HARNESS = os.path.join(HERE, "_ref", "ref_harness")

MAT_DIFFUSE, MAT_DIFFUSE_CHECKER, MAT_REFLECTIVE, MAT_EMISSIVE, MAT_BLEND, MAT_COMBINE, MAT_NULL = range(7)
UNITVEC_LIBM, UNITVEC_F64 = 0, 1

MATERIAL_DTYPE = np.dtype([("type", "<i4"), ("r", "<f4"), ("g", "<f4"), ("b", "<f4"), ("param", "<f4"),
                           ("child_a", "<i4"), ("child_b", "<i4"), ("pad", "<i4")])


def build(force=False):
    """Compile the oracle (and, when /root/reference is present, oracle/_ref)."""
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "rt_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "-j8", "oracle"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/Src/KdTree.cpp"):
        subprocess.check_call(["make", "-C", HERE, "-j8", "ref"], stdout=subprocess.DEVNULL)


_lib = None


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "box_tests", "tri_tests", "shaded_hits", "tex_samples", "camera_rays")]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_framebuffer_create.restype = C.c_void_p
        L.orc_render_pass_pool.restype = C.c_double
        L.orc_rand31.restype = C.c_uint32
        for name in ("orc_scene_destroy", "orc_framebuffer_destroy", "orc_framebuffer_clear", "orc_set_unitvec_mode",
                     "orc_set_combine_order", "orc_stats_reset", "orc_stats_get", "orc_unit_table_entry",
                     "orc_gamma_thresholds", "orc_texel_lut"):
            getattr(L, name).restype = None
        _lib = L
    return _lib


def materials(nodes):
    """nodes: list of (type, (r,g,b), param, child_a, child_b) in preorder, root first."""
    arr = np.zeros(len(nodes), dtype=MATERIAL_DTYPE)
    for i, (t, rgb, param, a, b) in enumerate(nodes):
        arr[i] = (t, rgb[0], rgb[1], rgb[2], param, a, b, 0)
    return arr


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Framebuffer:
    def __init__(self, width, height):
        self.width, self.height = width, height
        self.h = C.c_void_p(lib().orc_framebuffer_create(width, height))

    def clear(self):
        lib().orc_framebuffer_clear(self.h)

    def read(self):
        n = self.width * self.height
        accum = np.empty((n, 4), np.float32)
        argb = np.empty(n, np.uint32)
        lib().orc_framebuffer_read(self.h, _p(accum), _p(argb))
        return accum, argb

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_framebuffer_destroy(self.h)
            self.h = None


class Scene:
    def __init__(self):
        self.h = C.c_void_p(lib().orc_scene_create())

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_scene_destroy(self.h)
            self.h = None

    def add_mesh_obj(self, path, load_textures=True):
        s = lib().orc_scene_add_mesh_obj(self.h, path.encode())
        if s < 0:
            raise RuntimeError("oracle: cannot load %s" % path)
        if load_textures:
            from PIL import Image
            for m in range(lib().orc_mesh_num_materials(self.h, s)):
                p = self.texture_path(s, m)
                if p and os.path.exists(p):
                    im = Image.open(p)
                    if im.mode not in ("RGB", "RGBA"):   # reference accepts only 8-bit RGB/RGBA
                        continue
                    px = np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
                    self.set_texture(s, m, px)
        return s

    def add_sphere(self, center, radius):
        return lib().orc_scene_add_sphere(self.h, _p(np.asarray(center, np.float32)), C.c_float(radius))

    def add_plane(self, normal, point):
        return lib().orc_scene_add_plane(self.h, _p(np.asarray(normal, np.float32)), _p(np.asarray(point, np.float32)))

    def add_capsule(self, start, end, radius):
        return lib().orc_scene_add_capsule(self.h, _p(np.asarray(start, np.float32)), _p(np.asarray(end, np.float32)), C.c_float(radius))

    def add_triangle(self, p0, p1, p2):
        f = lambda v: _p(np.asarray(v, np.float32))  # noqa: E731
        return lib().orc_scene_add_triangle(self.h, f(p0), f(p1), f(p2))

    def add_shapes(self, scene, asset_path):
        """scene: a list from tests/scenes.py; asset_path(name) -> OBJ path"""
        for sh in scene:
            if sh[0] == "sphere":
                i = self.add_sphere(sh[1], sh[2])
            elif sh[0] == "plane":
                i = self.add_plane(sh[1], sh[2])
            elif sh[0] == "capsule":
                i = self.add_capsule(sh[1], sh[2], sh[3])
            elif sh[0] == "triangle":
                i = self.add_triangle(sh[1], sh[2], sh[3])
            else:
                i = self.add_mesh_obj(asset_path(sh[1]))
            if sh[-1] is not None:
                self.set_material(i, sh[-1])
        return self

    def texture_path(self, shape, mat):
        buf = C.create_string_buffer(8192)
        lib().orc_mesh_texture_path(self.h, shape, mat, buf, 8192)
        return buf.value.decode()

    def set_texture(self, shape, mat, px):
        px = np.ascontiguousarray(px, dtype=np.uint8)
        h, w, c = px.shape
        if lib().orc_mesh_set_texture(self.h, shape, mat, _p(px), w, h, c) != 0:
            raise RuntimeError("oracle: set_texture failed")

    def set_material(self, shape, nodes):
        arr = nodes if isinstance(nodes, np.ndarray) else materials(nodes)
        if lib().orc_scene_set_material(self.h, shape, _p(arr), len(arr)) != 0:
            raise RuntimeError("oracle: bad material")

    def set_unitvec_mode(self, mode):
        lib().orc_set_unitvec_mode(self.h, mode)

    def set_combine_order(self, b_first):
        lib().orc_set_combine_order(self.h, int(b_first))

    def counts(self, shape):
        out = np.zeros(8, np.int32)
        lib().orc_mesh_counts(self.h, shape, _p(out))
        return dict(points=int(out[0]), texcoords=int(out[1]), normals=int(out[2]), tris=int(out[3]),
                    materials=int(out[4]), nodes=int(out[5]), textures=int(out[6]))

    def mesh_arrays(self, shape):
        c = self.counts(shape)
        spec = [("points", 0, (c["points"], 3), np.float32), ("texcoords", 1, (c["texcoords"], 3), np.float32),
                ("normals", 2, (c["normals"], 3), np.float32), ("pidx", 3, (c["tris"], 3), np.int32),
                ("tidx", 4, (c["tris"], 3), np.int32), ("nidx", 5, (c["tris"], 3), np.int32),
                ("matid", 6, (c["tris"],), np.int32)]
        out = {}
        for name, which, shp, dt in spec:
            a = np.zeros(shp, dt)
            if lib().orc_mesh_copy(self.h, shape, which, _p(a), a.nbytes) != 0:
                raise RuntimeError("oracle: mesh_copy")
            out[name] = a
        return out

    def tree_preorder(self, shape):
        n = self.counts(shape)["nodes"]
        b = np.zeros((n, 6), np.float32)
        t = np.zeros(n, np.int32)
        got = lib().orc_mesh_tree_preorder(self.h, shape, _p(b), _p(t), n)
        assert got == n, (got, n)
        return b, t

    def shape_bounds(self, shape):
        o = np.zeros(6, np.float32)
        lib().orc_shape_bounds(self.h, shape, _p(o))
        return o

    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 7)
        n = len(rays)
        hf = np.zeros((n, 11), np.float32)
        hs = np.zeros(n, np.int32)
        ht = np.zeros(n, np.int32)
        lib().orc_trace_closest(self.h, _p(rays), C.c_int64(n), _p(hf), _p(hs), _p(ht))
        return hf, hs, ht

    def texture_sample(self, shape, mat, uv):
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        out = np.zeros((len(uv), 4), np.float32)
        if lib().orc_texture_sample(self.h, shape, mat, _p(uv), C.c_int64(len(uv)), _p(out)) != 0:
            raise RuntimeError("oracle: no such texture")
        return out

    def ray_trace(self, rays, keys, max_bounce, use_base_color, seed, width, height):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 7)
        keys = np.ascontiguousarray(keys, np.uint32).reshape(-1, 2)
        out = np.zeros((len(rays), 3), np.float32)
        lib().orc_ray_trace(self.h, _p(rays), _p(keys), C.c_int64(len(rays)), max_bounce, int(use_base_color),
                            C.c_uint32(seed), width, height, _p(out))
        return out

    def render_range(self, fb, begin, end, max_bounce, use_base_color=False, pass_index=0, ns=4, seed=12345, history=None):
        """history: an optional uint32 array (one word per pixel of the frame) into which the hit / miss history of every rendered
        pixel's paths is folded (test instrumentation: equal words = the paths took the same branches)"""
        if history is not None:
            assert history.dtype == np.uint32 and history.flags["C_CONTIGUOUS"]
            lib().orc_set_history_buffer(_p(history))
        try:
            r = lib().orc_render_range(self.h, fb.h, begin, end, max_bounce, int(use_base_color), pass_index, ns,
                                       C.c_uint32(seed))
        finally:
            if history is not None:
                lib().orc_set_history_buffer(None)
        if r != 0:
            raise RuntimeError("oracle: render_range bad arguments")

    def render_pass_pool(self, fb, max_bounce, use_base_color=False, pass_index=0, ns=4, seed=12345, threads=0,
                         task_rows=10):
        return lib().orc_render_pass_pool(self.h, fb.h, max_bounce, int(use_base_color), pass_index, ns,
                                          C.c_uint32(seed), threads, task_rows)


def stats_reset():
    lib().orc_stats_reset()


def stats_get():
    s = Stats()
    lib().orc_stats_get(C.byref(s))
    return {n: int(getattr(s, n)) for n, _ in Stats._fields_}


def rand31(seed, pixel, sample, counter):
    return int(lib().orc_rand31(C.c_uint32(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_uint32(counter)))


def unit_table_entry(i):
    o = np.zeros(3, np.float32)
    lib().orc_unit_table_entry(C.c_uint32(i), _p(o))
    return o


def gamma_thresholds():
    o = np.zeros(256, np.float32)
    lib().orc_gamma_thresholds(_p(o))
    return o


def texel_lut():
    o = np.zeros(256, np.float32)
    lib().orc_texel_lut(_p(o))
    return o


def make_pixel_colors(rgb):
    rgb = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
    out = np.zeros(len(rgb), np.uint32)
    lib().orc_make_pixel_colors.restype = None
    lib().orc_make_pixel_colors(_p(rgb), C.c_int64(len(rgb)), _p(out))
    return out


def hw_threads():
    return int(lib().orc_hw_threads())
