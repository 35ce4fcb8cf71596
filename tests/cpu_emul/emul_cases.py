"""TEST INFRASTRUCTURE: run under LD_PRELOAD=<asan runtime> with RTW_LIB=tests/cpu_emul/_build/librtwin_emul.so (tests/test_host_cpu.py does
that).  The product's own sources -- every kernel, every launch wrapper, the C ABI -- run on the CPU against the HIP stand-in of shim/ under
AddressSanitizer + UBSan; any out-of-bounds access (global memory or LDS), undefined operation, or cross-lane operation reached by only part
of a wave aborts the run.  Every frame is also compared, bit for bit, with the oracle's.
usage: emul_cases.py [case ...]      (no argument: all cases)"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracerwin_amd as R  # noqa: E402
import scenes as SC  # noqa: E402
from conftest import asset  # noqa: E402
from oracle import oracle as O  # noqa: E402

assert os.path.basename(R.library_path()) == "librtwin_emul.so", "emul_cases.py is for the sanitizer build of tests/cpu_emul only"


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def product_scene(ctx, shapes):
    s = R.RayTracerScene(ctx)
    for sh in shapes:
        mat = None if sh[-1] is None else R.material_nodes_from_array(O.materials(sh[-1]))
        if sh[0] == "sphere":
            s.AddShape(R.RSphere.Create(sh[1], sh[2]), mat)
        elif sh[0] == "plane":
            s.AddShape(R.RPlane.Create(sh[1], sh[2]), mat)
        elif sh[0] == "capsule":
            s.AddShape(R.RCapsule.Create(sh[1], sh[2], sh[3]), mat)
        elif sh[0] == "triangle":
            s.AddShape(R.RTriangle.Create(sh[1], sh[2], sh[3]), mat)
        else:
            s.AddShape(R.RMeshShape.Create(asset(sh[1] + ".obj")), mat)
    s.commit()
    return s


def oracle_frame(shapes, W, H, ns, depth, preview, passes, seed, f64=False):
    os_ = O.Scene().add_shapes(shapes, lambda n: asset(n + ".obj"))
    if f64:
        os_.set_unitvec_mode(O.UNITVEC_F64)
    fb = O.Framebuffer(W, H)
    for p in range(passes):
        os_.render_pass_pool(fb, depth, bool(preview), p, ns, seed, threads=0, task_rows=10)
    return fb.read()


def frame_case(ctx, name, shapes, W, H, ns, depth, passes, preview=False, seed=12345, options=(), f64=False, pipelines=(4,)):
    want = oracle_frame(shapes, W, H, ns, depth, preview, passes, seed, f64)
    for pl in pipelines:
        t0 = time.time()
        ctx.set_option("pipeline", pl)
        for k, v in options:
            ctx.set_option(k, v)
        s = product_scene(ctx, shapes)
        fb = R.Framebuffer(ctx, W, H)
        s.render_passes(fb, 10, 0, 1, depth, R.RenderOption(bool(preview)), 0, passes, ns, seed)
        used = ctx.last_pass_pipeline()
        accum, argb = fb.read_float(), fb.resolve_argb()
        fb.close(); s.close()
        for k, v in options:
            ctx.set_option(k, DEFAULTS[k])
        assert used == pl, (name, pl, used)
        same = bool((bits(accum) == bits(want[0])).all() and (argb == want[1]).all())
        print("  %-34s pipeline %d  %3dx%-3d %d spp depth %d x %d passes: %s  (%.1f s)" % (name, pl, W, H, ns, depth, passes, "bit-identical to the oracle" if same else "DIFFERS", time.time() - t0), flush=True)
        assert same, name
    ctx.set_option("pipeline", 4)


DEFAULTS = {"backface_filter": 1, "group_parts": 2, "workspace_limit_mb": 0, "wave_below": 80000, "group_max": 256, "device_build": 1, "visit_budget": 384, "split_min": 8, "split_paths": 400000, "primary_passes": 0}
mesh = lambda name, mat: [("mesh", name, mat)]  # noqa: E731


def case_mirror(ctx):
    frame_case(ctx, "TorusKnot mirror", mesh("TorusKnot", SC.reflective()), 96, 54, 1, 4, 3, options=[("device_build", 0)], pipelines=(4, 3, 0))


def case_textured(ctx):
    frame_case(ctx, "unitychan diffuse, textured", mesh("unitychan", SC.diffuse()), 64, 64, 2, 3, 2, options=[("device_build", 0)], pipelines=(4, 3))


def case_preview(ctx):
    frame_case(ctx, "unitychan preview", mesh("unitychan", SC.diffuse()), 80, 45, 1, 0, 1, preview=True, options=[("device_build", 0)])


def case_ragged(ctx):
    frame_case(ctx, "BlenderMonkey blend, ragged frame", mesh("BlenderMonkey", SC.blend(SC.reflective(), SC.diffuse(), 0.5)), 77, 43, 3, 4, 2, options=[("device_build", 0)])


def case_setup_scene(ctx):
    frame_case(ctx, "SetupScene (fuzz, lead shapes)", SC.SCENES["default"](), 48, 48, 1, 4, 2, options=[("device_build", 0)], f64=True, pipelines=(4, 3))


def case_setup_scene_lanes(ctx):
    """the default scene's bounce rounds through the persistent ray-per-lane kernel (leading analytic shapes + one mesh), also with a small visit budget"""
    frame_case(ctx, "SetupScene, ray per lane", SC.SCENES["default"](), 48, 48, 2, 5, 2, options=[("device_build", 0), ("wave_below", 0)], f64=True)
    frame_case(ctx, "SetupScene, ray per lane, budget 24", SC.SCENES["default"](), 48, 48, 1, 4, 2, options=[("device_build", 0), ("wave_below", 0), ("visit_budget", 24)], f64=True)


def case_quirk(ctx):
    frame_case(ctx, "texel inheritance (carry)", SC.SCENES["quirk"](), 48, 48, 2, 3, 2, options=[("device_build", 0)])


def case_room(ctx):
    frame_case(ctx, "room: 13 shapes, two meshes", SC.SCENES["room"](), 40, 40, 1, 4, 1, options=[("device_build", 0)])


def case_tris(ctx):
    frame_case(ctx, "RTriangle shapes", SC.SCENES["tris"](), 40, 40, 1, 3, 1, options=[("device_build", 0)])


def case_shapes(ctx):
    frame_case(ctx, "analytic shapes only", SC.SCENES["shapes"](), 48, 40, 2, 4, 2, options=[("device_build", 0)])


def case_trace_variants(ctx):
    m = mesh("TorusKnot", SC.reflective())
    frame_case(ctx, "wave-per-ray trace kernel", m, 64, 48, 1, 4, 2, options=[("device_build", 0), ("wave_below", 1 << 30)])
    frame_case(ctx, "ray-per-lane, persistent waves", m, 64, 48, 1, 4, 2, options=[("device_build", 0), ("wave_below", 0)])
    frame_case(ctx, "... without the back-face filter", m, 64, 48, 1, 4, 2, options=[("device_build", 0), ("wave_below", 0), ("backface_filter", 0)])
    frame_case(ctx, "ray-per-lane, budget 8 + overflow", m, 64, 48, 1, 4, 2, options=[("device_build", 0), ("wave_below", 0), ("visit_budget", 8)])
    frame_case(ctx, "groups of 2 passes", m, 64, 48, 1, 4, 5, options=[("device_build", 0), ("group_max", 2)])
    # the primary kernel's shared bins walk: four passes of a tile per wave (7 passes: 4 + 3), two passes x two sub-samples, and the one-ray-set order
    frame_case(ctx, "primary: 4 passes per wave", m, 64, 48, 1, 4, 7, options=[("device_build", 0), ("primary_passes", 4), ("split_min", 64)])
    frame_case(ctx, "primary: 2 passes x 2 sub-samples", m, 64, 48, 2, 3, 3, options=[("device_build", 0), ("primary_passes", 2), ("split_min", 64)])
    frame_case(ctx, "primary: one ray set at a time", m, 64, 48, 2, 3, 2, options=[("device_build", 0), ("primary_passes", -1)])
    u = mesh("unitychan", SC.diffuse())
    frame_case(ctx, "ray-per-lane, tree partly in LDS", u, 64, 64, 1, 3, 2, options=[("device_build", 0), ("wave_below", 0)])
    frame_case(ctx, "... and a budget of 40 visits", u, 64, 64, 1, 3, 2, options=[("device_build", 0), ("wave_below", 0), ("visit_budget", 40)])


def case_split(ctx):
    """a group as two halves (two streams, two workspaces; forced for these small frames)"""
    o = [("device_build", 0), ("split_min", 2), ("split_paths", 0)]
    frame_case(ctx, "two halves: mirror", mesh("TorusKnot", SC.reflective()), 96, 54, 1, 4, 7, options=o)
    frame_case(ctx, "two halves: SetupScene", SC.SCENES["default"](), 48, 48, 1, 4, 4, options=o, f64=True)
    frame_case(ctx, "three parts, uneven", mesh("TorusKnot", SC.reflective()), 96, 54, 2, 4, 8, options=o + [("group_parts", 3)])
    frame_case(ctx, "four parts", mesh("TorusKnot", SC.reflective()), 96, 54, 1, 4, 7, options=o + [("group_parts", 4)])


def case_workspace(ctx):
    """the workspace follows the groups a call forms; a refusal (here: the context's limit; on a GPU also a failing hipMalloc) re-forms smaller groups"""
    m = mesh("TorusKnot", SC.reflective())
    ctx.trim()
    assert ctx.workspace_bytes() == 0
    frame_case(ctx, "one pass per call", m, 96, 54, 1, 4, 1, options=[("device_build", 0)])
    one = ctx.workspace_bytes()
    frame_case(ctx, "7 passes, whole", m, 96, 54, 1, 4, 7, options=[("device_build", 0), ("split_min", 64)])
    whole = ctx.workspace_bytes()
    assert 0 < one and 4 * one <= whole <= 9 * one, (one, whole)          # 7 passes -> 8 slots per path
    ctx.trim()
    before = ctx.fallbacks()
    frame_case(ctx, "7 passes under a 1 MB limit", m, 96, 54, 1, 4, 7, options=[("device_build", 0), ("workspace_limit_mb", 1), ("split_min", 64)])
    assert ctx.fallbacks() > before and ctx.workspace_bytes() <= 1 << 20, (ctx.fallbacks(), before, ctx.workspace_bytes())
    # a refused SPLIT group re-formed into a smaller split group (the second round must not see the first one's refusal)
    before = ctx.fallbacks()
    frame_case(ctx, "7 passes, split, 1 MB limit", SC.SCENES["shapes"](), 192, 64, 1, 2, 7, options=[("device_build", 0), ("workspace_limit_mb", 1), ("group_max", 4), ("split_min", 2), ("split_paths", 0)])
    assert ctx.fallbacks() > before
    # a reserve ahead of the call: the call itself then allocates nothing
    ctx.trim()
    s = product_scene(ctx, m)
    fb = R.Framebuffer(ctx, 96, 54)
    s.render_reserve(fb, 10, 0, 1, 4, 7, 1)
    reserved = ctx.workspace_bytes()
    s.render_passes(fb, 10, 0, 1, 4, None, 0, 7, 1, 12345)
    assert reserved > 0 and ctx.workspace_bytes() == reserved, (reserved, ctx.workspace_bytes())
    fb.close(); s.close()
    print("  workspace: one pass %d B, 7 passes %d B, reserved %d B" % (one, whole, reserved), flush=True)


def case_device_build(ctx):
    """KdNode::Build, the derived layouts and the screen bins on the device (rtw_build_kernels.h) against the host build"""
    for name in ("TorusKnot", "BlenderMonkey"):
        out = {}
        for dev in (1, 0):
            t0 = time.time()
            ctx.set_option("device_build", dev)
            s = R.RayTracerScene(ctx)
            s.AddShape(R.RMeshShape.Create(asset(name + ".obj")), R.SurfaceMaterial_Diffuse())
            s.commit()
            out[dev] = (s.mesh_nodes(), [s.mesh_flat(l) for l in range(3)], s.mesh_bins(64, 48, 8, 8))
            s.close()
            print("  %-34s device_build %d  (%.1f s)" % (name, dev, time.time() - t0), flush=True)
        ctx.set_option("device_build", 1)
        same = lambda u, v: np.array_equal(np.ascontiguousarray(u).view(np.uint8), np.ascontiguousarray(v).view(np.uint8))  # noqa: E731
        (n1, f1, b1), (n0, f0, b0) = out[1], out[0]
        assert all(same(x, y) for x, y in zip(n1, n0)), name
        assert all(same(x, y) for x, y in zip(f1, f0)), name
        assert (b1 is None) == (b0 is None) and (b1 is None or all(same(x, y) for x, y in zip(b1, b0))), name
    frame_case(ctx, "frame over device-built tree + bins", mesh("TorusKnot", SC.reflective()), 64, 48, 1, 3, 2)


def case_queries(ctx):
    """FindIntersectionWithScene / RayTrace / texture sampling entry points"""
    ctx.set_option("device_build", 0)
    g = np.load(os.path.join(ROOT, "tests", "golden", "closest_unitychan.npz"))
    s = product_scene(ctx, mesh("unitychan", SC.diffuse()))
    n = 4096
    hits, shape, tri = s.FindIntersectionWithScene(g["rays"][:n])
    assert (shape == g["shape"][:n]).all()
    print("  %-34s %d rays: shapes identical to the reference's" % ("FindIntersectionWithScene", n), flush=True)
    s.close()
    ctx.set_option("device_build", 1)


CASES = {k[5:]: v for k, v in list(globals().items()) if k.startswith("case_")}

if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    ctx = R.Context(0)
    for n in names:
        print("case %s" % n, flush=True)
        CASES[n](ctx)
    print("emul ok: %d cases" % len(names))
