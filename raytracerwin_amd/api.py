"""ctypes binding of librtwin.so + the host-side mirror of the reference interface."""
import atexit
import ctypes as C
import os
import subprocess
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("RTW_LIB") or os.path.join(_HERE, "librtwin.so")      # RTW_LIB: another BUILD of the same sources (timing instrumentation; the sanitizer build of tests/cpu_emul) -- never a fallback: unset, it is librtwin.so or an ImportError

MATERIAL_DTYPE = np.dtype([("type", "<i4"), ("r", "<f4"), ("g", "<f4"), ("b", "<f4"), ("param", "<f4"),
                           ("child_a", "<i4"), ("child_b", "<i4"), ("pad", "<i4")])
MAT_DIFFUSE, MAT_DIFFUSE_CHECKER, MAT_REFLECTIVE, MAT_EMISSIVE, MAT_BLEND, MAT_COMBINE, MAT_NULL = range(7)


class RtwError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("librtwin error %d: %s" % (code, msg))
        self.code = code


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "box_tests", "tri_tests", "shaded_hits", "tex_samples", "camera_rays")]


def library_path():
    return _LIB


def build_library(force=False, extra=""):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if extra:
        cmd.append("EXTRA=" + extra)
    if force:
        subprocess.check_call(cmd + ["clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def library():
    """Load librtwin.so; fails loudly when the extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise ImportError("raytracerwin_amd: %s is missing -- run __graft_entry__.build() "
                              "(there is no CPU fallback)" % _LIB)
        L = C.CDLL(_LIB)
        L.rtw_last_error.restype = C.c_char_p
        L.rtw_version.restype = C.c_char_p
        if b"EMULATION" in L.rtw_version() and os.environ.get("RTW_TEST_EMULATION") != "1":
            # tests/cpu_emul's host build of the same sources (AddressSanitizer runs): never a way to render without a GPU
            raise ImportError("raytracerwin_amd: %s is the sanitizer harness's host build; it is loaded by tests/test_host_cpu.py only "
                              "(RTW_TEST_EMULATION=1)" % _LIB)
        L.rtw_rand31.restype = C.c_uint32
        L.rtw_png_free.restype = None
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise RtwError(rc, library().rtw_last_error().decode())
    return rc


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------------------------
# materials: same constructors as Src/SurfaceMaterials.h, flattened to rtw_material_node[]
# ---------------------------------------------------------------------------------------------
class ISurfaceMaterial:
    def _flatten(self, out):
        raise NotImplementedError

    def nodes(self):
        out = []
        self._flatten(out)
        arr = np.zeros(len(out), dtype=MATERIAL_DTYPE)
        for i, n in enumerate(out):
            arr[i] = n
        return arr


class _Leaf(ISurfaceMaterial):
    TYPE = None

    def __init__(self, rgb=(1.0, 1.0, 1.0), param=0.0):
        self.rgb, self.param = tuple(float(v) for v in rgb), float(param)

    def _flatten(self, out):
        out.append((self.TYPE, self.rgb[0], self.rgb[1], self.rgb[2], self.param, 0, 0, 0))
        return len(out) - 1


class SurfaceMaterial_Diffuse(_Leaf):
    TYPE = MAT_DIFFUSE

    def __init__(self, InAlbedo=(1.0, 1.0, 1.0)):
        super().__init__(InAlbedo)


class SurfaceMaterial_DiffuseChecker(_Leaf):
    TYPE = MAT_DIFFUSE_CHECKER

    def __init__(self, InAlbedo=(1.0, 1.0, 1.0), InPatternSize=5.0):
        super().__init__(InAlbedo, InPatternSize)


class SurfaceMaterial_Reflective(_Leaf):
    TYPE = MAT_REFLECTIVE

    def __init__(self, InAlbedo=(1.0, 1.0, 1.0), InFuzziness=0.0):
        super().__init__(InAlbedo, InFuzziness)


class SurfaceMaterial_Emissive(_Leaf):
    TYPE = MAT_EMISSIVE

    def __init__(self, InColor):
        super().__init__(InColor)


class SurfaceMaterial_Null(_Leaf):
    TYPE = MAT_NULL

    def __init__(self):
        super().__init__((0.0, 0.0, 0.0))


class _Pair(ISurfaceMaterial):
    TYPE = None

    def __init__(self, a, b, param=0.0):
        self.a, self.b, self.param = a, b, float(param)

    def _flatten(self, out):
        me = len(out)
        out.append(None)
        ia = self.a._flatten(out)
        ib = self.b._flatten(out)
        out[me] = (self.TYPE, 0.0, 0.0, 0.0, self.param, ia, ib, 0)
        return me


class SurfaceMaterial_Blend(_Pair):
    TYPE = MAT_BLEND

    def __init__(self, InMaterialA, InMaterialB, InBlendFactor):
        super().__init__(InMaterialA, InMaterialB, InBlendFactor)


class SurfaceMaterial_Combine(_Pair):
    TYPE = MAT_COMBINE

    def __init__(self, InMaterialA, InMaterialB):
        super().__init__(InMaterialA, InMaterialB)


class _RawMaterial(ISurfaceMaterial):
    def __init__(self, arr):
        self.arr = np.ascontiguousarray(arr, dtype=MATERIAL_DTYPE)

    def nodes(self):
        return self.arr


def material_nodes_from_array(arr):
    return _RawMaterial(arr)


class RenderOption:
    """Src/RayTracerScene.h:27-35"""

    def __init__(self, UseBaseColor=False):
        self.UseBaseColor = bool(UseBaseColor)


class RMeshShape:
    """RMeshShape::Create(path) (Src/MeshShape.h:23): a deferred description; the OBJ is
    parsed by the library when the shape is added to a scene."""

    def __init__(self, Filename=None, arrays=None):
        self.Filename, self.arrays = Filename, arrays

    @staticmethod
    def Create(Filename):
        return RMeshShape(Filename=Filename)

    @staticmethod
    def FromArrays(points, texcoords, normals, pidx, tidx, nidx, matid=None, bounds=None, textures=None):
        return RMeshShape(arrays=dict(points=points, texcoords=texcoords, normals=normals, pidx=pidx, tidx=tidx,
                                      nidx=nidx, matid=matid, bounds=bounds, textures=textures or {}))


class RSphere:
    """RSphere::Create(center, radius) (Src/Shapes.h:46-60)"""

    def __init__(self, Center, Radius):
        self.Center, self.Radius = tuple(float(v) for v in Center), float(Radius)

    @staticmethod
    def Create(InCenter, InRadius):
        return RSphere(InCenter, InRadius)


class RPlane:
    """RPlane::Create(normal, point) (Src/Shapes.h:63-79): infinite, never culled"""

    def __init__(self, Normal, Point):
        self.Normal, self.Point = tuple(float(v) for v in Normal), tuple(float(v) for v in Point)

    @staticmethod
    def Create(InNormal, InPoint):
        return RPlane(InNormal, InPoint)


class RCapsule:
    """RCapsule::Create(start, end, radius) (Src/Shapes.h:82-104)"""

    def __init__(self, Start, End, Radius):
        self.Start, self.End, self.Radius = tuple(float(v) for v in Start), tuple(float(v) for v in End), float(Radius)

    @staticmethod
    def Create(InStart, InEnd, InRadius):
        return RCapsule(InStart, InEnd, InRadius)


class RTriangle:
    """RTriangle::Create(p0, p1, p2) (Src/Shapes.h:106-130)"""

    def __init__(self, p0, p1, p2):
        self.Points = [tuple(float(v) for v in p) for p in (p0, p1, p2)]

    @staticmethod
    def Create(p0, p1, p2):
        return RTriangle(p0, p1, p2)


# ---------------------------------------------------------------------------------------------
_live_contexts = weakref.WeakSet()


@atexit.register
def _close_all_contexts():
    # release device objects while the HIP runtime is still alive (not from __del__ at interpreter teardown)
    for c in list(_live_contexts):
        c.close()


class Context:
    """One per (process, GPU): stream + the device tables shared by every scene."""

    def __init__(self, device=0, stream=None):
        self.h = C.c_void_p()
        self._children = weakref.WeakSet()      # scenes / framebuffers: destroyed before the context
        _check(library().rtw_context_create(int(device), C.byref(self.h)))
        self.device = device
        _live_contexts.add(self)
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, hip_stream):
        _check(library().rtw_context_set_stream(self.h, C.c_void_p(int(hip_stream))))

    def synchronize(self):
        _check(library().rtw_context_synchronize(self.h))

    def set_option(self, name, value):
        _check(library().rtw_context_set_option(self.h, name.encode(), int(value)))

    def last_pass_kernel_ms(self):
        out = (C.c_float * 3)()
        _check(library().rtw_last_pass_kernel_ms(self.h, out))
        return [float(v) for v in out]

    def last_group_passes(self):
        return int(library().rtw_last_group_passes(self.h))

    def last_pass_pipeline(self):
        """the pipeline the latest render call actually ran (a fallback shows here)"""
        return int(library().rtw_last_pass_pipeline(self.h))

    def memory_bytes(self):
        """device memory the context holds (workspaces + its share of the unit-vector table)"""
        L = library()
        L.rtw_context_memory_bytes.restype = C.c_longlong
        return int(L.rtw_context_memory_bytes(self.h))

    def workspace_bytes(self):
        L = library()
        L.rtw_context_workspace_bytes.restype = C.c_longlong
        return int(L.rtw_context_workspace_bytes(self.h))

    def trim(self):
        _check(library().rtw_context_trim(self.h))

    def fallbacks(self):
        return int(library().rtw_context_fallbacks(self.h))

    def stats_enable(self, on=True):
        _check(library().rtw_stats_enable(self.h, int(on)))

    def stats_reset(self):
        _check(library().rtw_stats_reset(self.h))

    def stats(self):
        s = Stats()
        _check(library().rtw_stats_get(self.h, C.byref(s)))
        return {n: int(getattr(s, n)) for n, _ in Stats._fields_}

    def close(self):
        if getattr(self, "h", None):
            for child in list(self._children):
                child.close()
            library().rtw_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """rtw_comm: an RCCL communicator for the one exchange of the path (rtw_gather_rows).  Rank 0 makes an id with Comm.unique_id() and hands its
    128 bytes to every rank (torch.distributed, MPI, a file ...); every rank then constructs Comm(ctx, rank, world, ident).  With world == 1 no
    id is needed."""

    @staticmethod
    def unique_id():
        ident = (C.c_uint8 * 128)()
        _check(library().rtw_comm_unique_id(ident))
        return bytes(ident)

    def __init__(self, ctx, rank, world, ident=None):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self.h = C.c_void_p()
        if ident is None:
            if self.world != 1:
                raise ValueError("Comm: every rank of a world > 1 needs rank 0's id")
            ident = Comm.unique_id()
        raw = (C.c_uint8 * 128).from_buffer_copy(bytes(ident))
        _check(library().rtw_comm_create(ctx.h, raw, self.rank, self.world, C.byref(self.h)))
        ctx._children.add(self)

    def gather_rows(self, fb, task_rows, argb_only=False):
        _check(library().rtw_gather_rows(self.h, fb.h, int(task_rows), 1 if argb_only else 0))

    def messages(self):
        """ncclSend / ncclRecv operations issued so far: a gather costs a sender one, the root world - 1"""
        L = library()
        L.rtw_comm_messages.restype = C.c_longlong
        return int(L.rtw_comm_messages(self.h))

    def close(self):
        if getattr(self, "h", None):
            library().rtw_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Framebuffer:
    """accuBuffer[] + bitcolor[] (Src/RayTracerProgram.cpp:49-77) on the device."""

    def __init__(self, ctx, width, height, accum_ptr=None, argb_ptr=None):
        self.ctx, self.width, self.height = ctx, int(width), int(height)
        self.h = C.c_void_p()
        ctx._children.add(self)
        if accum_ptr is None:
            _check(library().rtw_framebuffer_create(ctx.h, self.width, self.height, C.byref(self.h)))
        else:
            _check(library().rtw_framebuffer_wrap(ctx.h, self.width, self.height, C.c_void_p(int(accum_ptr)),
                                                  C.c_void_p(int(argb_ptr)), C.byref(self.h)))

    def clear(self):
        _check(library().rtw_framebuffer_clear(self.h))

    def read_float(self):
        out = np.empty((self.width * self.height, 4), np.float32)
        _check(library().rtw_framebuffer_read_float(self.h, _p(out)))
        return out

    def resolve_argb(self):
        out = np.empty(self.width * self.height, np.uint32)
        _check(library().rtw_framebuffer_resolve_argb(self.h, _p(out)))
        return out

    def save_png(self, path):
        argb = self.resolve_argb()
        _check(library().rtw_png_save_argb(path.encode(), _p(argb), self.width, self.height))

    def close(self):
        if getattr(self, "h", None):
            library().rtw_framebuffer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RayTracerScene:
    """Src/RayTracerScene.h:43-62.  AddShape() collects shapes; the first query commits."""

    def __init__(self, ctx):
        # ctx=None gives a host-only scene (parse / build / inspect); device queries then raise
        self.ctx = ctx
        self.h = C.c_void_p()
        if ctx is not None:
            ctx._children.add(self)
        _check(library().rtw_scene_create(ctx.h if ctx is not None else None, C.byref(self.h)))
        self.committed = False
        self.n_shapes = 0

    def AddShape(self, Shape, SurfaceMaterial=None):
        L = library()
        idx = C.c_int(-1)
        v3 = lambda v: (C.c_float * 3)(*v)  # noqa: E731
        if isinstance(Shape, RSphere):
            _check(L.rtw_scene_add_sphere(self.h, v3(Shape.Center), C.c_float(Shape.Radius), C.byref(idx)))
        elif isinstance(Shape, RPlane):
            _check(L.rtw_scene_add_plane(self.h, v3(Shape.Normal), v3(Shape.Point), C.byref(idx)))
        elif isinstance(Shape, RTriangle):
            _check(L.rtw_scene_add_triangle(self.h, v3(Shape.Points[0]), v3(Shape.Points[1]), v3(Shape.Points[2]), C.byref(idx)))
        elif isinstance(Shape, RCapsule):
            _check(L.rtw_scene_add_capsule(self.h, v3(Shape.Start), v3(Shape.End), C.c_float(Shape.Radius), C.byref(idx)))
        elif Shape.Filename is not None:
            _check(L.rtw_scene_add_mesh_obj(self.h, Shape.Filename.encode(), C.byref(idx)))
        else:
            a = Shape.arrays
            f = lambda k: np.ascontiguousarray(a[k], np.float32).reshape(-1, 3)  # noqa: E731
            i = lambda k: np.ascontiguousarray(a[k], np.int32).reshape(-1, 3)  # noqa: E731
            pts, tcs, nrm, pi, ti, ni = f("points"), f("texcoords"), f("normals"), i("pidx"), i("tidx"), i("nidx")
            mat = None if a["matid"] is None else np.ascontiguousarray(a["matid"], np.int32)
            bnd = None if a["bounds"] is None else np.ascontiguousarray(a["bounds"], np.float32)
            _check(L.rtw_scene_add_mesh(self.h, _p(pts), len(pts), _p(tcs), len(tcs), _p(nrm), len(nrm), _p(pi), _p(ti),
                                        _p(ni), None if mat is None else _p(mat), len(pi),
                                        None if bnd is None else _p(bnd), C.byref(idx)))
            for mid, px in a["textures"].items():
                px = np.ascontiguousarray(px, np.uint8)
                h, w, c = px.shape
                _check(L.rtw_scene_set_texture(self.h, idx.value, int(mid), _p(px), w, h, c))
        if SurfaceMaterial is not None:
            nodes = SurfaceMaterial.nodes()
            _check(L.rtw_scene_set_material(self.h, idx.value, _p(nodes), len(nodes)))
        self.n_shapes += 1
        return idx.value

    def set_prune(self, enabled):
        _check(library().rtw_scene_set_prune(self.h, int(enabled)))

    def set_traversal(self, mode):
        _check(library().rtw_scene_set_traversal(self.h, int(mode)))

    def commit(self):
        if not self.committed:
            _check(library().rtw_scene_commit(self.h))
            self.committed = True

    def mesh_info(self, shape=0):
        info = np.zeros(8, np.int32)
        b = np.zeros(6, np.float32)
        _check(library().rtw_scene_mesh_info(self.h, shape, _p(info), _p(b)))
        keys = ("points", "texcoords", "normals", "tris", "materials", "nodes", "textures", "max_depth")
        d = {k: int(v) for k, v in zip(keys, info)}
        d["bounds"] = b
        return d

    def mesh_nodes(self, shape=0):
        self.commit()
        n = self.mesh_info(shape)["nodes"]
        b = np.zeros((n, 6), np.float32)
        s = np.zeros(n, np.int32)
        t = np.zeros(n, np.int32)
        _check(library().rtw_scene_mesh_nodes(self.h, shape, _p(b), _p(s), _p(t), n))
        return b, s, t

    def mesh_flat(self, level, shape=0):
        """boxes (n, 6) of one level of the flat leaf hierarchy (0 = leaves in preorder, 1 = groups of 16, 2 = groups of 256)"""
        self.commit()
        n = library().rtw_scene_mesh_flat(self.h, shape, int(level), None, 0)
        _check(n)
        b = np.zeros((n, 6), np.float32)
        _check(library().rtw_scene_mesh_flat(self.h, shape, int(level), _p(b), n))
        return b

    def mesh_bins(self, width, height, bin_w, bin_h, shape=0):
        """(offsets, entries) of the reference camera's screen bins, or None when the mesh gets none"""
        self.commit()
        counts = np.zeros(2, np.int64)
        rc = library().rtw_scene_mesh_bins(self.h, shape, int(width), int(height), int(bin_w), int(bin_h), None, C.c_int64(0), None,
                                           C.c_int64(0), _p(counts))
        _check(rc)
        if rc == 0:
            return None
        off = np.zeros(int(counts[0]), np.uint32)
        ent = np.zeros(max(1, int(counts[1])), np.uint32)
        _check(library().rtw_scene_mesh_bins(self.h, shape, int(width), int(height), int(bin_w), int(bin_h), _p(off), C.c_int64(len(off)),
                                             _p(ent), C.c_int64(int(counts[1])), _p(counts)))
        return off, ent[:int(counts[1])]

    def FindIntersectionWithScene(self, rays):
        """rays: (n,7) origin, direction, distance -> (hits (n,11), shape (n,), triangle (n,))"""
        self.commit()
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 7)
        n = len(rays)
        hits = np.zeros((n, 11), np.float32)
        shape = np.zeros(n, np.int32)
        tri = np.zeros(n, np.int32)
        _check(library().rtw_trace_closest(self.h, _p(rays), C.c_int64(n), _p(hits), _p(shape), _p(tri)))
        return hits, shape, tri

    def RayTrace(self, rays, keys, MaxBounceTimes, InOption=None, seed=12345, width=1920, height=1080):
        self.commit()
        opt = InOption or RenderOption()
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 7)
        keys = np.ascontiguousarray(keys, np.uint32).reshape(-1, 2)
        out = np.zeros((len(rays), 3), np.float32)
        _check(library().rtw_ray_trace(self.h, _p(rays), _p(keys), C.c_int64(len(rays)), int(MaxBounceTimes),
                                       int(opt.UseBaseColor), C.c_uint32(seed), int(width), int(height), _p(out)))
        return out

    def texture_sample(self, shape, material_id, uv):
        self.commit()
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        out = np.zeros((len(uv), 4), np.float32)
        _check(library().rtw_texture_sample(self.h, shape, material_id, _p(uv), C.c_int64(len(uv)), _p(out)))
        return out

    def render_tasks(self, fb, task_rows, rank, world, MaxBounceCount, InOption=None, pass_index=0, sub_samples=4,
                     seed=12345):
        self.commit()
        opt = InOption or RenderOption()
        _check(library().rtw_render_tasks(self.h, fb.h, int(task_rows), int(rank), int(world), int(MaxBounceCount),
                                          int(opt.UseBaseColor), int(pass_index), int(sub_samples), C.c_uint32(seed)))

    def render_passes(self, fb, task_rows, rank, world, MaxBounceCount, InOption=None, first_pass=0, n_passes=1,
                      sub_samples=4, seed=12345):
        """UpdateBitmapPixels' sample loop (Src/RayTracerProgram.cpp:317-361): n_passes accumulated passes, same images
        as render_tasks pass by pass; after the first passes one captured launch graph is replayed per pass."""
        self.commit()
        opt = InOption or RenderOption()
        _check(library().rtw_render_passes(self.h, fb.h, int(task_rows), int(rank), int(world), int(MaxBounceCount),
                                           int(opt.UseBaseColor), int(first_pass), int(n_passes), int(sub_samples),
                                           C.c_uint32(seed)))

    def render_reserve(self, fb, task_rows, rank, world, MaxBounceCount, n_passes, sub_samples=4):
        """rtw_render_reserve: bins, tile tables and workspace of a later render_passes call with these arguments (renders nothing)"""
        self.commit()
        _check(library().rtw_render_reserve(self.h, fb.h, int(task_rows), int(rank), int(world), int(MaxBounceCount), int(n_passes), int(sub_samples)))

    def close(self):
        if getattr(self, "h", None):
            library().rtw_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ThreadWorker_Render(scene, fb, begin, end, MaxBounceCount, InOption=None, pass_index=0, sub_samples=4, seed=12345):
    """void ThreadWorker_Render(int begin, int end, int MaxBounceCount, const RenderOption&)
    (Src/RayTracerProgram.cpp:131): the scene and buffers the reference reaches through globals
    are explicit here; begin..end are inclusive linear pixel indices.  Asynchronous."""
    scene.commit()
    opt = InOption or RenderOption()
    _check(library().rtw_render_range(scene.h, fb.h, int(begin), int(end), int(MaxBounceCount), int(opt.UseBaseColor),
                                      int(pass_index), int(sub_samples), C.c_uint32(seed)))


def rand31(seed, pixel, sample, counter):
    return int(library().rtw_rand31(C.c_uint32(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_uint32(counter)))


def unit_table_entry(i):
    o = np.zeros(3, np.float32)
    _check(library().rtw_unit_table_entry(C.c_uint32(i), _p(o)))
    return o


def gamma_thresholds():
    o = np.zeros(256, np.float32)
    _check(library().rtw_gamma_thresholds(_p(o)))
    return o


def texel_lut():
    o = np.zeros(256, np.float32)
    _check(library().rtw_texel_lut(_p(o)))
    return o


def png_load(path):
    px = C.POINTER(C.c_uint8)()
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    _check(library().rtw_png_load(path.encode(), C.byref(px), C.byref(w), C.byref(h), C.byref(c)))
    try:
        return np.ctypeslib.as_array(px, shape=(h.value, w.value, c.value)).copy()
    finally:
        library().rtw_png_free(px)
