"""GPU parity: the HIP path (through the C ABI) against the golden vectors from the real reference,
against the CPU oracle on seeded inputs, and -- at BASELINE.json's full sizes -- through
size-independent properties.  Integer/index outputs and every float output except the
fuzzy-reflection frames are compared BIT-EXACTLY; the tolerance for the rest is written in place."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, asset

pytestmark = pytest.mark.gpu

import raytracerwin_amd as R  # noqa: E402

DEFAULT_PIPELINE = 4        # pass-batched: screen bins + a ray per lane, K passes per set of launches


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def same(a, b):
    """bit-identical, except that any NaN equals any NaN (payload/sign of a NaN is not specified)"""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return bool(((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))).all())


@pytest.fixture(scope="module")
def ctx():
    c = R.Context(0)
    yield c
    c.close()


def gpu_scene(ctx, mesh, material):
    s = R.RayTracerScene(ctx)
    s.AddShape(R.RMeshShape.Create(asset(mesh + ".obj")), material)
    s.commit()
    return s


def render_frame(ctx, scene, W, H, ns, depth, preview, seed, pass0=0, npass=1):
    fb = R.Framebuffer(ctx, W, H)
    for p in range(pass0, pass0 + npass):
        R.ThreadWorker_Render(scene, fb, 0, W * H - 1, depth, R.RenderOption(bool(preview)), p, ns, seed)
    return fb.read_float(), fb.resolve_argb()


MESHES = ["TorusKnot", "BlenderMonkey", "unitychan"]


@pytest.mark.parametrize("name", MESHES)
@pytest.mark.parametrize("prune", [0, 1])
def test_closest_hit_bit_exact_vs_reference_golden(ctx, name, prune):
    g = np.load(os.path.join(GOLDEN, "closest_%s.npz" % name))
    s = gpu_scene(ctx, name, R.SurfaceMaterial_Diffuse())
    s.set_prune(prune)
    hits, shape, tri = s.FindIntersectionWithScene(g["rays"])
    assert (shape == g["shape"]).all()
    hit = shape >= 0
    assert (tri[hit] == g["tri"][hit]).all()
    assert (bits(hits[hit]) == bits(g["hit"][hit])).all()


@pytest.mark.parametrize("name", MESHES)
def test_closest_hit_bit_exact_vs_oracle_large_batch(ctx, oracle_mod, name):
    """Fresh seeded rays (camera, interior, axis-parallel, near-epsilon, NaN/inf) vs the oracle."""
    from tests.golden.make_golden import make_rays
    os_ = oracle_mod.Scene()
    sh = os_.add_mesh_obj(asset(name + ".obj"))
    rng = np.random.default_rng(99)
    rays = make_rays(os_.shape_bounds(sh), 1500 if name == "unitychan" else 6000, rng)
    odd = rays[:8].copy()
    odd[0, 3] = np.nan
    odd[1, 5] = np.nan
    odd[2, 0] = np.inf
    odd[3, 3:6] = 0
    odd[4, 6] = 0
    odd[5, 6] = -1
    odd[6, 4] = 1e-8
    odd[7, 3] = np.inf
    s = gpu_scene(ctx, name, R.SurfaceMaterial_Diffuse())
    if name == "unitychan":
        # NaN rays reach RTexture::Sample with NaN uv: out-of-bounds reads in the reference (and in the oracle).
        # The GPU path must survive them; there is nothing defined to compare against.
        s.FindIntersectionWithScene(odd)
    else:
        rays = np.concatenate([rays, odd])
    of, oshape, otri = os_.trace_closest(rays)
    for prune in (0, 1):
        s.set_prune(prune)
        hits, shape, tri = s.FindIntersectionWithScene(rays)
        assert (shape == oshape).all()
        hit = shape >= 0
        assert (tri[hit] == otri[hit]).all()
        assert same(hits[hit], of[hit])


@pytest.mark.parametrize("mat", [0, 4])
def test_texture_sample_bit_exact(ctx, mat):
    g = np.load(os.path.join(GOLDEN, "texsample_unitychan_m%d.npz" % mat))
    s = gpu_scene(ctx, "unitychan", R.SurfaceMaterial_Diffuse())
    out = s.texture_sample(0, mat, g["uv"])
    assert (bits(out) == bits(g["rgba"])).all()


FRAMES = sorted(os.path.basename(p)[6:-4] for p in glob.glob(os.path.join(GOLDEN, "frame_*.npz")))
FUZZY = {"torus_blendfuzz_d6", "monkey_blendfuzz_d6"}


@pytest.mark.parametrize("tag", FRAMES)
def test_frame_vs_reference_golden(ctx, tag):
    g = np.load(os.path.join(GOLDEN, "frame_%s.npz" % tag))
    W, H, ns, depth, preview, seed, pass0, npass = [int(v) for v in g["params"]]
    s = gpu_scene(ctx, str(g["mesh"]), R.material_nodes_from_array(g["material"]))
    accum, argb = render_frame(ctx, s, W, H, ns, depth, preview, seed, pass0, npass)
    if tag in FUZZY:
        # fuzzy reflection: the device evaluates sin/cos/acos in double (not libm's float routines), so a direction may differ in its
        # last bit.  Stated tolerance: 1e-4 per channel.  OBSERVED on these fixtures (and asserted): every pixel within 2e-6, a handful
        # differ at all; a deviation beyond 1e-4 could only come from a hit turned into a miss (tests/test_oracle_golden.py::
        # test_fuzzy_reflection_modes_differ_only_where_the_hit_history_differs shows that with per-pixel hit histories), none occurs here
        d = np.abs(accum[:, :3] - g["accum"][:, :3]).max(axis=1)
        print("fuzzy frame %s vs the reference: %d of %d pixels differ, max deviation %.3g" % (tag, int((d > 0).sum()), len(d), d.max()))
        assert d.max() <= 2e-6
        assert (d > 0).mean() <= 0.01
        assert (accum[:, 3] == g["accum"][:, 3]).all()
        return
    assert (argb == g["argb"]).all()
    if not preview:
        assert (bits(accum) == bits(g["accum"])).all()


@pytest.mark.parametrize("tag", ["torus_blendfuzz_d6", "monkey_blendfuzz_d6"])
def test_fuzzy_frames_bit_exact_vs_oracle_f64_mode(ctx, oracle_mod, tag):
    """With the oracle's transcendental mode switched to the device's (double sin/cos/acos), even the
    fuzzy-reflection frames must agree bit for bit."""
    O = oracle_mod
    g = np.load(os.path.join(GOLDEN, "frame_%s.npz" % tag))
    W, H, ns, depth, preview, seed, pass0, npass = [int(v) for v in g["params"]]
    os_ = O.Scene()
    sh = os_.add_mesh_obj(asset(str(g["mesh"]) + ".obj"))
    os_.set_material(sh, g["material"])
    os_.set_unitvec_mode(O.UNITVEC_F64)
    ofb = O.Framebuffer(W, H)
    for p in range(pass0, pass0 + npass):
        os_.render_range(ofb, 0, W * H - 1, depth, False, p, ns, seed)
    oa, ob = ofb.read()
    s = gpu_scene(ctx, str(g["mesh"]), R.material_nodes_from_array(g["material"]))
    accum, argb = render_frame(ctx, s, W, H, ns, depth, preview, seed, pass0, npass)
    assert (bits(accum) == bits(oa)).all()
    assert (argb == ob).all()


@pytest.mark.parametrize("tag", ["unitychan_diffuse", "monkey_blendfuzz"])
def test_ray_trace_vs_reference_golden(ctx, tag):
    g = np.load(os.path.join(GOLDEN, "raytrace_%s.npz" % tag))
    depth, seed, W, H = [int(v) for v in g["params"]]
    s = gpu_scene(ctx, str(g["mesh"]), R.material_nodes_from_array(g["material"]))
    rgb = s.RayTrace(g["rays"], g["keys"], depth, None, seed, W, H)
    if tag == "monkey_blendfuzz":
        d = np.abs(rgb - g["rgb"]).max(axis=1)       # tolerance 1e-4 (double vs libm transcendentals); observed and asserted: 2e-6
        print("fuzzy rays vs the reference: %d of %d differ, max deviation %.3g" % (int((d > 0).sum()), len(d), d.max()))
        assert d.max() <= 2e-6
    else:
        assert (bits(rgb) == bits(g["rgb"])).all()


def test_config1_torus_256_depth1_vs_oracle(ctx, oracle_mod):
    """BASELINE configs[0]: TorusKnot 256x256, 1 spp, depth 1."""
    O = oracle_mod
    os_ = O.Scene()
    sh = os_.add_mesh_obj(asset("TorusKnot.obj"))
    os_.set_material(sh, [(O.MAT_DIFFUSE, (1, 1, 1), 0, 0, 0)])
    ofb = O.Framebuffer(256, 256)
    os_.render_pass_pool(ofb, 1, False, 0, 1, 12345, threads=0, task_rows=10)
    oa, ob = ofb.read()
    s = gpu_scene(ctx, "TorusKnot", R.SurfaceMaterial_Diffuse())
    accum, argb = render_frame(ctx, s, 256, 256, 1, 1, 0, 12345)
    assert (bits(accum) == bits(oa)).all() and (argb == ob).all()


def test_config2_band_vs_oracle_and_stats(ctx, oracle_mod):
    """BASELINE configs[1] (TorusKnot 1080p, 1 spp, depth 4): the centre band of rows against the
    oracle bit for bit, with reference-faithful work counters equal on both sides."""
    O = oracle_mod
    W, H = 1920, 1080
    r0, r1 = 500, 580
    os_ = O.Scene()
    sh = os_.add_mesh_obj(asset("TorusKnot.obj"))
    os_.set_material(sh, [(O.MAT_DIFFUSE, (1, 1, 1), 0, 0, 0)])
    ofb = O.Framebuffer(W, H)
    O.stats_reset()
    os_.render_range(ofb, r0 * W, r1 * W - 1, 4, False, 0, 1, 12345)
    ostats = O.stats_get()
    oa, ob = ofb.read()
    s = gpu_scene(ctx, "TorusKnot", R.SurfaceMaterial_Diffuse())
    s.set_prune(0)
    s.set_traversal(0)          # the binary preorder walk visits exactly the boxes the reference visits
    ctx.stats_enable(True)
    ctx.stats_reset()
    fb = R.Framebuffer(ctx, W, H)
    R.ThreadWorker_Render(s, fb, r0 * W, r1 * W - 1, 4, None, 0, 1, 12345)
    accum, argb = fb.read_float(), fb.resolve_argb()
    gstats = ctx.stats()
    ctx.stats_enable(False)
    sl = slice(r0 * W, r1 * W)
    assert (bits(accum[sl]) == bits(oa[sl])).all() and (argb[sl] == ob[sl]).all()
    assert (accum[:r0 * W] == 0).all() and (accum[r1 * W:] == 0).all()
    assert gstats == ostats


@pytest.mark.parametrize("mesh,mat,ns,depth", [("TorusKnot", "diffuse", 1, 4), ("BlenderMonkey", "blend", 4, 6),
                                               ("unitychan", "diffuse", 4, 4)])
def test_full_size_properties(ctx, mesh, mat, ns, depth):
    """1920x1080 (configs 2-4): (a) pruned traversal == reference-order traversal bit for bit;
    (b) any split into disjoint ThreadWorker_Render ranges == one full-frame call; (c) row tasks dealt
    to 1, 2, 3 ranks give the same image; (d) rendering twice is deterministic."""
    W, H = 1920, 1080
    material = R.SurfaceMaterial_Diffuse() if mat == "diffuse" else \
        R.SurfaceMaterial_Blend(R.SurfaceMaterial_Reflective((1, 1, 1), 0.2), R.SurfaceMaterial_Diffuse(), 0.5)
    s = gpu_scene(ctx, mesh, material)
    s.set_prune(1)
    a1, b1 = render_frame(ctx, s, W, H, ns, depth, 0, 777)
    a2, b2 = render_frame(ctx, s, W, H, ns, depth, 0, 777)
    assert (bits(a1) == bits(a2)).all() and (b1 == b2).all()
    s.set_prune(0)
    a0, b0 = render_frame(ctx, s, W, H, ns, depth, 0, 777)
    assert (bits(a1) == bits(a0)).all() and (b1 == b0).all()
    s.set_prune(1)
    fb = R.Framebuffer(ctx, W, H)
    cuts = [0, 1, 1919, 1920, 400000, 400001, 1234567, W * H]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        R.ThreadWorker_Render(s, fb, lo, hi - 1, depth, None, 0, ns, 777)
    assert (bits(fb.read_float()) == bits(a1)).all() and (fb.resolve_argb() == b1).all()
    for world in (2, 3):
        fbw = R.Framebuffer(ctx, W, H)
        for rank in range(world):
            s.render_tasks(fbw, 10, rank, world, depth, None, 0, ns, 777)
        assert (bits(fbw.read_float()) == bits(a1)).all() and (fbw.resolve_argb() == b1).all()
    assert (a1[:, 3] == 1).all()
    assert np.isfinite(a1[:, :3]).all()


def test_accumulation_over_passes_and_empty_range(ctx, oracle_mod):
    O = oracle_mod
    W, H = 160, 90
    os_ = O.Scene()
    sh = os_.add_mesh_obj(asset("BlenderMonkey.obj"))
    os_.set_material(sh, [(O.MAT_DIFFUSE, (0.9, 0.8, 0.7), 0, 0, 0)])
    ofb = O.Framebuffer(W, H)
    for p in range(4):
        os_.render_range(ofb, 0, W * H - 1, 8, False, p, 4, 5)
    oa, ob = ofb.read()
    s = gpu_scene(ctx, "BlenderMonkey", R.SurfaceMaterial_Diffuse((0.9, 0.8, 0.7)))
    fb = R.Framebuffer(ctx, W, H)
    for p in range(4):
        R.ThreadWorker_Render(s, fb, 0, W * H - 1, 8, None, p, 4, 5)
    R.ThreadWorker_Render(s, fb, 10, 9, 8, None, 0, 4, 5)        # begin > end: nothing, like the reference loop
    accum, argb = fb.read_float(), fb.resolve_argb()
    assert (bits(accum) == bits(oa)).all() and (argb == ob).all()
    with pytest.raises(R.RtwError):
        R.ThreadWorker_Render(s, fb, 0, W * H, 8, None, 0, 4, 5)
    with pytest.raises(R.RtwError):
        R.ThreadWorker_Render(s, fb, 0, 10, 17, None, 0, 4, 5)


def test_unitychan_textured_frame_vs_oracle(ctx, oracle_mod):
    """Config 4 shape at reduced size: textured shading with alpha pass-through, bit-exact."""
    O = oracle_mod
    W, H = 320, 180
    os_ = O.Scene()
    sh = os_.add_mesh_obj(asset("unitychan.obj"))
    os_.set_material(sh, [(O.MAT_DIFFUSE, (1, 1, 1), 0, 0, 0)])
    ofb = O.Framebuffer(W, H)
    O.stats_reset()
    os_.render_pass_pool(ofb, 4, False, 0, 4, 2024, threads=0, task_rows=10)
    ostats = O.stats_get()
    oa, ob = ofb.read()
    assert ostats["tex_samples"] > 100
    s = gpu_scene(ctx, "unitychan", R.SurfaceMaterial_Diffuse())
    accum, argb = render_frame(ctx, s, W, H, 4, 4, 0, 2024)
    assert (bits(accum) == bits(oa)).all() and (argb == ob).all()


def test_two_meshes_in_one_scene(ctx, oracle_mod):
    """FindIntersectionWithScene over shapes in insertion order (closest wins, later shape on ties)."""
    O = oracle_mod
    os_ = O.Scene()
    a = os_.add_mesh_obj(asset("TorusKnot.obj"))
    b = os_.add_mesh_obj(asset("BlenderMonkey.obj"))
    os_.set_material(a, [(O.MAT_DIFFUSE, (1, 0.5, 0.5), 0, 0, 0)])
    os_.set_material(b, [(O.MAT_REFLECTIVE, (0.5, 1, 0.5), 0, 0, 0)])
    ofb = O.Framebuffer(128, 128)
    os_.render_range(ofb, 0, 128 * 128 - 1, 5, False, 0, 4, 31)
    oa, ob = ofb.read()
    s = R.RayTracerScene(ctx)
    s.AddShape(R.RMeshShape.Create(asset("TorusKnot.obj")), R.SurfaceMaterial_Diffuse((1, 0.5, 0.5)))
    s.AddShape(R.RMeshShape.Create(asset("BlenderMonkey.obj")), R.SurfaceMaterial_Reflective((0.5, 1, 0.5)))
    accum, argb = render_frame(ctx, s, 128, 128, 4, 5, 0, 31)
    assert (bits(accum) == bits(oa)).all() and (argb == ob).all()


@pytest.mark.parametrize("mesh,ns,depth,preview", [("TorusKnot", 1, 4, 0), ("unitychan", 4, 4, 0), ("BlenderMonkey", 3, 0, 0),
                                                   ("TorusKnot", 4, 3, 1)])
def test_the_three_pipelines_give_the_same_bits_and_counters(ctx, mesh, ns, depth, preview):
    """The pass-batched default (4; also with every trace round forced through the ray-per-lane kernels, and through the wave-per-ray kernel), the one-pass
    bins + wave pipeline (3) and the one-thread-per-pixel kernel (0) give the same bits and the same per-ray work counters; 1280x720, two accumulated passes."""
    W, H = 1280, 720
    s = gpu_scene(ctx, mesh, R.SurfaceMaterial_Diffuse((0.9, 0.9, 0.8)))
    out = []
    for mode, wave_below in ((0, 80000), (3, 80000), (4, 80000), (4, 0), (4, 1 << 30)):
        ctx.set_option("pipeline", mode)
        ctx.set_option("wave_below", wave_below)
        ctx.stats_enable(True)
        ctx.stats_reset()
        a, b = render_frame(ctx, s, W, H, ns, depth, preview, 4321, 0, 2)
        out.append((a, b, ctx.stats()))
        ctx.stats_enable(False)
    ctx.set_option("pipeline", DEFAULT_PIPELINE)
    ctx.set_option("wave_below", 80000)
    keys = ("rays", "shaded_hits", "tex_samples", "camera_rays")     # box / triangle test counts depend on the walk
    for o in out[1:]:
        assert (bits(out[0][0]) == bits(o[0])).all() and (out[0][1] == o[1]).all()
        assert [out[0][2][k] for k in keys] == [o[2][k] for k in keys]


@pytest.mark.parametrize("mesh,ns,depth", [("TorusKnot", 1, 4), ("unitychan", 4, 4), ("BlenderMonkey", 4, 6)])
def test_accelerated_walks_equal_the_reference_order_walk(ctx, mesh, ns, depth):
    """traversal=1 (screen bins, link tree and flat hierarchy: candidates gathered, then triangle-tested in order) vs traversal=0 (the binary
    preorder walk in the reference's visit order, one thread per pixel), with and without pruning: same bits at 1920x1080."""
    W, H = 1920, 1080
    s = gpu_scene(ctx, mesh, R.SurfaceMaterial_Diffuse())
    ref = None
    for trav, prune in ((0, 0), (0, 1), (1, 0), (1, 1)):
        s.set_traversal(trav)
        s.set_prune(prune)
        a, b = render_frame(ctx, s, W, H, ns, depth, 0, 99)
        if ref is None:
            ref = (a, b)
        else:
            assert (bits(a) == bits(ref[0])).all() and (b == ref[1]).all(), (trav, prune)


@pytest.mark.gpu
def test_config5_shape_4k_depth8_pipelines_agree_and_tasks_compose(ctx):
    """BASELINE configs[4] at its full frame size (unitychan + textures, 3840x2160, depth 8; one 4-spp pass of the 16 spp):
    the pass-batched and the bins + wave pipelines give the bits of the single kernel, and the frame dealt as 10-row tasks
    over 8 ranks (rendered one after another into one buffer) equals the one-rank frame."""
    W, H = 3840, 2160
    s = gpu_scene(ctx, "unitychan", R.SurfaceMaterial_Diffuse((1, 1, 1)))
    ctx.set_option("pipeline", 0)
    a1, b1 = render_frame(ctx, s, W, H, 4, 8, 0, 2024, 3, 1)
    ctx.set_option("pipeline", 3)
    a3, b3 = render_frame(ctx, s, W, H, 4, 8, 0, 2024, 3, 1)
    assert (bits(a1) == bits(a3)).all() and (b1 == b3).all()
    ctx.set_option("pipeline", DEFAULT_PIPELINE)
    a4, b4 = render_frame(ctx, s, W, H, 4, 8, 0, 2024, 3, 1)
    assert (bits(a4) == bits(a3)).all() and (b4 == b3).all()
    fb = R.Framebuffer(ctx, W, H)
    for rank in range(8):
        s.render_tasks(fb, 10, rank, 8, 8, None, 3, 4, 2024)
    a8, b8 = fb.read_float(), fb.resolve_argb()
    assert (bits(a8) == bits(a3)).all() and (b8 == b3).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mesh,W,H,ns,depth", [("unitychan", 64, 200, 4, 5), ("BlenderMonkey", 64, 600, 4, 4), ("TorusKnot", 64, 450, 2, 2),
                                               ("BlenderMonkey", 48, 1000, 3, 3), ("TorusKnot", 1000, 48, 4, 3), ("unitychan", 37, 211, 4, 2)])
def test_tall_narrow_and_odd_frames_bins_equal_single_kernel(ctx, mesh, W, H, ns, depth):
    """The sub-sample jitter is 1 / (4 W) in camera units whatever the height (Src/RayTracerProgram.cpp:147-162), so on a tall
    narrow frame it spans several PIXELS; the screen bins' margin has to follow it.  Frames of such shapes (and sizes no tile
    shape divides) through the bins + wave pipeline equal the one-thread-per-pixel kernel."""
    s = gpu_scene(ctx, mesh, R.SurfaceMaterial_Diffuse((0.9, 0.8, 0.7)))
    ctx.set_option("pipeline", 0)
    a0, b0 = render_frame(ctx, s, W, H, ns, depth, 0, 555, 1, 2)
    ctx.set_option("pipeline", 3)
    a3, b3 = render_frame(ctx, s, W, H, ns, depth, 0, 555, 1, 2)
    assert (bits(a0) == bits(a3)).all() and (b0 == b3).all()
    ctx.set_option("pipeline", DEFAULT_PIPELINE)        # frames that do not tile run through the bins too (partial tiles), no fallback
    a4, b4 = render_frame(ctx, s, W, H, ns, depth, 0, 555, 1, 2)
    assert ctx.last_pass_pipeline() == 4
    assert (bits(a0) == bits(a4)).all() and (b0 == b4).all()


@pytest.mark.gpu
def test_randomised_scenes_frames_and_options_bins_equal_single_kernel(ctx):
    """tools/soak.py: 60 random cases (OBJ assets, two-mesh scenes, triangle soups with degenerate / axis-aligned / coincident /
    behind-the-camera triangles; random material trees, frame shapes, spp, depth, rank splits and pipeline options) through the
    one-thread-per-pixel kernel and through the bins + wave pipeline: same bits."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("rtw_soak", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "soak.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = []
    bad = mod.run(20261004, 60, ctx, lambda *a, **k: lines.append(" ".join(str(x) for x in a)))
    assert bad == 0, [ln for ln in lines if ln.endswith("DIFF")]


# ---- multi-shape scenes: RSphere / RPlane / RCapsule beside meshes (tests/scenes.py) ----------------------------------
import scenes as SC  # noqa: E402
from oracle import oracle as _O  # noqa: E402  (material node arrays only)


def multi_scene_gpu(ctx, tag):
    s = R.RayTracerScene(ctx)
    for sh in SC.SCENES[tag]():
        mat = None if sh[-1] is None else R.material_nodes_from_array(_O.materials(sh[-1]))
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


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["default_nofuzz", "quirk", "shapes", "room", "tris"])
def test_scene_closest_hit_vs_reference_golden(ctx, tag):
    """FindIntersectionWithScene over spheres, planes, capsules and meshes in insertion order, bit-exact against the reference's
    own outputs -- including (scene "quirk") the sampled colour an analytic hit inherits from an earlier textured-mesh hit."""
    g = np.load(os.path.join(GOLDEN, "sceneclosest_%s.npz" % tag))
    s = multi_scene_gpu(ctx, tag)
    hit, shape, tri = s.FindIntersectionWithScene(g["rays"])
    assert (shape == g["shape"]).all()
    m = shape >= 0
    assert same(hit[m], g["hit"][m])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["default_nofuzz_d5", "default_preview", "quirk_d4", "quirk_preview", "shapes_d6", "shapes_1spp_d2", "room_d8", "tris_d5"])
@pytest.mark.parametrize("pipeline", [4, 3, 0])
def test_scene_frame_vs_reference_golden(ctx, name, pipeline):
    """RayTracerProgram::SetupScene's scene (fuzziness zeroed, see tests/scenes.py), the texture-inheritance scene and the
    analytic-only scene: the frames the reference's RayTrace rendered, bit for bit, through the default pipeline and the single kernel."""
    g = np.load(os.path.join(GOLDEN, "sceneframe_%s.npz" % name))
    W, H, ns, depth, preview, seed, pass0, npass = [int(v) for v in g["params"]]
    s = multi_scene_gpu(ctx, str(g["scene"]))
    ctx.set_option("pipeline", pipeline)
    try:
        accum, argb = render_frame(ctx, s, W, H, ns, depth, preview, seed, pass0, npass)
        used = ctx.last_pass_pipeline()
    finally:
        ctx.set_option("pipeline", DEFAULT_PIPELINE)
    if pipeline == 4:       # no cliff: texel-inheritance scenes (a textured mesh before an analytic shape) stay in the pass-batched pipeline
        assert used == 4
    assert (argb == g["argb"]).all()
    if not preview:
        assert (bits(accum) == bits(g["accum"])).all()


@pytest.mark.gpu
def test_default_scene_with_fuzz_vs_oracle_f64_mode_and_close_to_reference(ctx, oracle_mod):
    """SetupScene as written (fuzzy reflections on the capsule, the ground and the mesh): bit-exact against the oracle in the
    device's transcendental mode, and within 1e-4 of the reference's frame on >= 99 % of the pixels."""
    O = oracle_mod
    g = np.load(os.path.join(GOLDEN, "sceneframe_default_d5.npz"))
    W, H, ns, depth, preview, seed, pass0, npass = [int(v) for v in g["params"]]
    os_ = O.Scene().add_shapes(SC.SCENES["default"](), lambda n: asset(n + ".obj"))
    os_.set_unitvec_mode(O.UNITVEC_F64)
    ofb = O.Framebuffer(W, H)
    for p in range(pass0, pass0 + npass):
        os_.render_range(ofb, 0, W * H - 1, depth, False, p, ns, seed)
    oa, ob = ofb.read()
    s = multi_scene_gpu(ctx, "default")
    accum, argb = render_frame(ctx, s, W, H, ns, depth, preview, seed, pass0, npass)
    assert (bits(accum) == bits(oa)).all() and (argb == ob).all()
    d = np.abs(accum[:, :3] - g["accum"][:, :3]).max(axis=1)
    print("SetupScene with fuzz vs the reference: %d of %d pixels differ, max deviation %.3g" % (int((d > 0).sum()), len(d), d.max()))
    assert d.max() <= 2e-6          # stated tolerance 1e-4; observed: 35 of 6 400 pixels differ, by at most 5e-7


@pytest.mark.gpu
@pytest.mark.parametrize("tag,W,H,ns,depth", [("default", 1280, 720, 4, 5), ("shapes", 800, 800, 4, 6), ("default_nofuzz", 333, 211, 3, 4),
                                              ("quirk", 640, 360, 4, 4), ("room", 800, 800, 4, 8), ("tris", 512, 512, 4, 5)])
def test_scene_pipelines_agree_at_size(ctx, tag, W, H, ns, depth):
    """larger frames than the oracle-sized fixtures: bins + wave pipeline == single kernel, whole frame and dealt as tasks over 3 ranks"""
    s = multi_scene_gpu(ctx, tag)
    ctx.set_option("pipeline", 0)
    a0, b0 = render_frame(ctx, s, W, H, ns, depth, 0, 31, 0, 2)
    ctx.set_option("pipeline", 3)
    a3, b3 = render_frame(ctx, s, W, H, ns, depth, 0, 31, 0, 2)
    assert (bits(a0) == bits(a3)).all() and (b0 == b3).all()
    ctx.set_option("pipeline", DEFAULT_PIPELINE)
    a4, b4 = render_frame(ctx, s, W, H, ns, depth, 0, 31, 0, 2)
    assert ctx.last_pass_pipeline() == 4
    assert (bits(a0) == bits(a4)).all() and (b0 == b4).all()
    fb = R.Framebuffer(ctx, W, H)
    for rank in range(3):
        s.render_passes(fb, 10, rank, 3, depth, None, 0, 2, ns, 31)
    assert (bits(fb.read_float()) == bits(a3)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mesh,ns,depth,world", [("TorusKnot", 1, 4, 1), ("unitychan", 4, 3, 2), ("BlenderMonkey", 2, 5, 3)])
def test_a_run_of_passes_with_one_fork_and_join_equals_pass_by_pass_calls(ctx, mesh, ns, depth, world):
    """rtw_render_passes launches its passes as one run (the second stream, which renders the sky-only tiles, is forked before the
    first pass and joined after the last): same bits as one rtw_render_tasks call per pass, also when other
    work follows the call at once (a read-back, another rank's run into the same buffer)."""
    W, H = 1280, 720
    s = gpu_scene(ctx, mesh, R.SurfaceMaterial_Diffuse((0.9, 0.8, 1.0)))
    ref = R.Framebuffer(ctx, W, H)
    for rank in range(world):
        for p in range(7):
            s.render_tasks(ref, 10, rank, world, depth, None, p, ns, 99)
    ra, rb = ref.read_float(), ref.resolve_argb()
    ctx.set_option("pipeline", 3)
    fb = R.Framebuffer(ctx, W, H)
    for rank in range(world):
        s.render_passes(fb, 10, rank, world, depth, None, 0, 4, ns, 99)
        s.render_passes(fb, 10, rank, world, depth, None, 4, 3, ns, 99)
    a, b = fb.read_float(), fb.resolve_argb()
    assert (bits(a) == bits(ra)).all() and (b == rb).all()
    ctx.set_option("pipeline", DEFAULT_PIPELINE)


# ---- the pass-batched pipeline (pipeline 4, the default): K passes share one set of launches ---------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("mesh,W,H,ns,depth,world", [("TorusKnot", 1920, 1080, 1, 4, 1), ("unitychan", 640, 360, 4, 4, 2), ("BlenderMonkey", 333, 217, 3, 6, 3),
                                                     ("TorusKnot", 640, 360, 2, 1, 1), ("BlenderMonkey", 200, 64, 4, 0, 1)])
def test_k_batched_passes_equal_pass_by_pass(ctx, mesh, W, H, ns, depth, world):
    """rtw_render_passes renders its passes in groups of K that share one set of launches (K x the rays per launch); a pixel's pass colours
    are added in pass order, its accumulator entry and ARGB word written once per group.  For every group size, first pass and split into calls the
    accumulator and the ARGB image are those of one rtw_render_tasks call per pass through the single kernel (pipeline 0)."""
    mat = R.SurfaceMaterial_Blend(R.SurfaceMaterial_Reflective((0.9, 0.9, 0.9), 0.0), R.SurfaceMaterial_Diffuse((1.0, 0.9, 0.8)), 0.5)
    s = gpu_scene(ctx, mesh, mat)
    npass, first = 11, 2
    ctx.set_option("pipeline", 0)
    ref = R.Framebuffer(ctx, W, H)
    for rank in range(world):
        for p in range(first, first + npass):
            s.render_tasks(ref, 10, rank, world, depth, None, p, ns, 4242)
    ra, rb = ref.read_float(), ref.resolve_argb()
    ctx.set_option("pipeline", DEFAULT_PIPELINE)
    try:
        for gmax, calls in ((64, (11,)), (4, (11,)), (2, (3, 8)), (1, (11,)), (8, (1, 1, 9)), (16, (5, 6))):
            ctx.set_option("group_max", gmax)
            fb = R.Framebuffer(ctx, W, H)
            for rank in range(world):
                p = first
                for n in calls:
                    s.render_passes(fb, 10, rank, world, depth, None, p, n, ns, 4242)
                    p += n
            assert ctx.last_pass_pipeline() == 4
            a, b = fb.read_float(), fb.resolve_argb()
            assert (bits(a) == bits(ra)).all() and (b == rb).all(), (gmax, calls)
            fb.close()
    finally:
        ctx.set_option("group_max", 256)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,W,H,ns,depth,npass,parts", [("mesh", 640, 360, 2, 4, 13, 2), ("default_nofuzz", 400, 400, 4, 5, 8, 2), ("quirk", 320, 180, 4, 4, 9, 2),
                                                          ("mesh", 640, 360, 2, 4, 13, 3), ("default_nofuzz", 400, 400, 4, 5, 11, 4)])
def test_group_as_parts_on_several_streams(ctx, tag, W, H, ns, depth, npass, parts):
    """A group of passes runs as two (option group_parts: up to four) parts on as many streams (own workspaces; the group's sky kernel, all its passes, is
    enqueued ahead of the parts; a part's resolve waits for the previous part's).  Forced here for small frames (split_min 2, split_paths 0; by default only
    groups of >= 8 passes with >= 400 k paths per part are split): accumulator and ARGB image equal one call per pass through the single kernel, for one call
    and for the same passes in two calls."""
    if tag == "mesh":
        s = gpu_scene(ctx, "TorusKnot", R.SurfaceMaterial_Blend(R.SurfaceMaterial_Reflective((0.9, 0.9, 0.9), 0.0), R.SurfaceMaterial_Diffuse((1.0, 0.9, 0.8)), 0.5))
    else:
        s = multi_scene_gpu(ctx, tag)
    ctx.set_option("pipeline", 0)
    ref = R.Framebuffer(ctx, W, H)
    for p in range(npass):
        R.ThreadWorker_Render(s, ref, 0, W * H - 1, depth, None, p, ns, 77)
    ra, rb = ref.read_float(), ref.resolve_argb()
    ctx.set_option("pipeline", DEFAULT_PIPELINE)
    ctx.set_option("split_min", 2)
    ctx.set_option("split_paths", 0)
    ctx.set_option("group_parts", parts)
    try:
        for calls in ((npass,), (5, npass - 5)):
            fb = R.Framebuffer(ctx, W, H)
            p = 0
            for n in calls:
                s.render_passes(fb, 10, 0, 1, depth, None, p, n, ns, 77)
                p += n
            assert ctx.last_pass_pipeline() == 4
            assert (bits(fb.read_float()) == bits(ra)).all() and (fb.resolve_argb() == rb).all(), calls
            fb.close()
    finally:
        ctx.set_option("split_min", 8)
        ctx.set_option("split_paths", 400000)
        ctx.set_option("group_parts", 2)


@pytest.mark.gpu
@pytest.mark.parametrize("mesh,W,H,ns,depth,npass", [("TorusKnot", 1920, 1080, 1, 4, 11), ("unitychan", 640, 360, 2, 3, 5), ("BlenderMonkey", 501, 283, 4, 4, 3),
                                                    ("BlenderMonkey", 333, 211, 3, 3, 4)])
def test_primary_kernel_walks_a_bin_once_for_several_rays(ctx, mesh, W, H, ns, depth, npass):
    """One-mesh scenes: a wave of the primary kernel takes up to four rays per lane through its tile's bin at once -- the pixel's sub-samples and / or the
    same pixel in several passes (option primary_passes: 0 = chosen per launch, 1, 2, 4; -1 = one ray set at a time, the order every other kernel is
    compared with).  Every setting gives the accumulator and the ARGB image of one call per pass through the single kernel (pass counts that do not
    divide by the passes per wave, ragged frames, 1 - 4 sub-samples)."""
    s = gpu_scene(ctx, mesh, R.SurfaceMaterial_Blend(R.SurfaceMaterial_Reflective((0.9, 0.9, 0.9), 0.0), R.SurfaceMaterial_Diffuse((1.0, 0.9, 0.8)), 0.5))
    ctx.set_option("pipeline", 0)
    ref = R.Framebuffer(ctx, W, H)
    for p in range(npass):
        R.ThreadWorker_Render(s, ref, 0, W * H - 1, depth, None, p, ns, 4242)
    ra, rb = ref.read_float(), ref.resolve_argb()
    ctx.set_option("pipeline", DEFAULT_PIPELINE)
    try:
        for pp in (0, -1, 1, 2, 4):
            ctx.set_option("primary_passes", pp)
            fb = R.Framebuffer(ctx, W, H)
            s.render_passes(fb, 10, 0, 1, depth, None, 0, npass, ns, 4242)
            assert ctx.last_pass_pipeline() == 4
            assert (bits(fb.read_float()) == bits(ra)).all() and (fb.resolve_argb() == rb).all(), pp
            fb.close()
    finally:
        ctx.set_option("primary_passes", 0)


@pytest.mark.gpu
def test_groups_of_more_than_64_passes(ctx):
    """The largest groups the default policy forms (256 passes: a rank's share of a frame at 8 ranks, small frames): 200 passes of a small frame
    in one group (8 bits of a slot hold the pass), against one call per pass through the single kernel."""
    s = gpu_scene(ctx, "TorusKnot", R.SurfaceMaterial_Reflective((0.9, 0.9, 0.9), 0.0))
    W, H, depth, npass = 320, 180, 4, 200
    ctx.set_option("pipeline", 0)
    ref = R.Framebuffer(ctx, W, H)
    for p in range(npass):
        R.ThreadWorker_Render(s, ref, 0, W * H - 1, depth, None, p, 1, 31)
    ra, rb = ref.read_float(), ref.resolve_argb()
    ctx.set_option("pipeline", DEFAULT_PIPELINE)
    fb = R.Framebuffer(ctx, W, H)
    s.render_passes(fb, 10, 0, 1, depth, None, 0, npass, 1, 31)
    assert ctx.last_pass_pipeline() == 4 and ctx.last_group_passes() == npass
    assert (bits(fb.read_float()) == bits(ra)).all() and (fb.resolve_argb() == rb).all()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,W,H,ns,depth", [("default_nofuzz", 400, 400, 4, 5), ("quirk", 320, 180, 4, 4), ("room", 256, 256, 2, 6), ("shapes", 333, 211, 4, 6)])
def test_k_batched_passes_multi_shape_scenes(ctx, tag, W, H, ns, depth):
    """The same for scenes with spheres / planes / capsules / triangles beside meshes, including the texel-inheritance scene (a textured mesh
    BEFORE an analytic shape, Src/RRay.cpp:53-58,75-80), which the pass-batched pipeline renders itself (hit records carry the mesh hit whose
    sampled colour the analytic hit keeps) -- no fallback to the single kernel."""
    s = multi_scene_gpu(ctx, tag)
    ctx.set_option("pipeline", 0)
    ref = R.Framebuffer(ctx, W, H)
    for p in range(6):
        R.ThreadWorker_Render(s, ref, 0, W * H - 1, depth, None, p, ns, 99)
    ra, rb = ref.read_float(), ref.resolve_argb()
    ctx.set_option("pipeline", DEFAULT_PIPELINE)
    for gmax in (64, 2):
        ctx.set_option("group_max", gmax)
        fb = R.Framebuffer(ctx, W, H)
        s.render_passes(fb, 10, 0, 1, depth, None, 0, 6, ns, 99)
        assert ctx.last_pass_pipeline() == 4
        assert (bits(fb.read_float()) == bits(ra)).all() and (fb.resolve_argb() == rb).all(), gmax
        fb.close()
    ctx.set_option("group_max", 256)


@pytest.mark.gpu
@pytest.mark.parametrize("mesh,ns,depth", [("TorusKnot", 1, 4), ("unitychan", 4, 4)])
def test_pass_batched_trace_variants_agree(ctx, mesh, ns, depth):
    """The trace round's variants -- a ray per lane in persistent waves that refill their lanes, a wave per ray for short lists, rays that exceed their
    visit budget handed to the wave-per-ray kernel -- give the same bits (1280x720, 5 passes)."""
    W, H = 1280, 720
    s = gpu_scene(ctx, mesh, R.SurfaceMaterial_Diffuse((0.9, 0.9, 0.8)))
    ref = None
    try:
        for below, budget in ((100000, 256), (0, 0), (0, 8), (100000000, 256), (0, 64)):
            for k, v in (("wave_below", below), ("visit_budget", budget)):
                ctx.set_option(k, v)
            fb = R.Framebuffer(ctx, W, H)
            s.render_passes(fb, 10, 0, 1, depth, None, 0, 5, ns, 2025)
            s.render_passes(fb, 10, 0, 1, depth, None, 5, 5, ns, 2025)       # (the second call knows the first one's list lengths: short lists take the wave-per-ray kernel)
            out = (fb.read_float(), fb.resolve_argb())
            fb.close()
            if ref is None:
                ref = out
            else:
                assert (bits(out[0]) == bits(ref[0])).all() and (out[1] == ref[1]).all(), (below, budget)
    finally:
        for k, v in (("wave_below", 80000), ("visit_budget", 384)):
            ctx.set_option(k, v)
    ctx.set_option("pipeline", 0)
    a0, b0 = R.Framebuffer(ctx, W, H), None
    for p in range(10):
        R.ThreadWorker_Render(s, a0, 0, W * H - 1, depth, None, p, ns, 2025)
    ctx.set_option("pipeline", DEFAULT_PIPELINE)
    assert (bits(a0.read_float()) == bits(ref[0])).all() and (a0.resolve_argb() == ref[1]).all()


WORKERS = sorted(os.path.basename(p)[7:-4] for p in glob.glob(os.path.join(GOLDEN, "worker_*.npz")))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", WORKERS)
def test_frame_vs_the_references_own_thread_worker_render(ctx, tag):
    """The GPU frame against what the reference's OWN ThreadWorker_Render wrote into its accuBuffer[] / bitcolor[] (800 x 800, 4 sub-samples,
    its compiled-in constants; fixtures worker_*.npz): ARGB for every pixel, the accumulator through its SHA-256 and a 64-row band, bit for bit.
    Passes rendered one call each and as one K-batched rtw_render_passes call."""
    from test_oracle_golden import worker_digest
    g = np.load(os.path.join(GOLDEN, "worker_%s.npz" % tag))
    W, H, ns, depth, preview, seed, pass0, npass = [int(v) for v in g["params"]]
    s = gpu_scene(ctx, str(g["mesh"]), R.material_nodes_from_array(g["material"]))
    for batched in (False, True):
        fb = R.Framebuffer(ctx, W, H)
        if batched:
            s.render_passes(fb, 10, 0, 1, depth, R.RenderOption(bool(preview)), pass0, npass, ns, seed)
        else:
            for p in range(pass0, pass0 + npass):
                R.ThreadWorker_Render(s, fb, 0, W * H - 1, depth, R.RenderOption(bool(preview)), p, ns, seed)
        accum, argb = fb.read_float(), fb.resolve_argb()
        fb.close()
        assert (argb == g["argb"]).all()
        if not preview:
            r0, r1 = [int(v) for v in g["band"]]
            assert (bits(accum[r0 * W:r1 * W]) == bits(g["accum_band"])).all()
            assert worker_digest(accum) == str(g["accum_sha256"])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,mesh,W,H,ns,depth,passes,blend", [("c2", "TorusKnot", 1920, 1080, 1, 4, 3, False), ("c3", "BlenderMonkey", 1920, 1080, 4, 6, 2, True),
                                                                ("c4", "unitychan", 1920, 1080, 4, 4, 2, False), ("c5", "unitychan", 3840, 2160, 4, 8, 4, False)])
def test_full_size_frames_bit_exact_vs_oracle(ctx, oracle_mod, cfg, mesh, W, H, ns, depth, passes, blend):
    """BASELINE configs[1..4] at their FULL frame sizes, every pass (C5: all four 4-spp passes of its 16 spp at 3840x2160, depth 8), the GPU's
    accumulator and ARGB image against the oracle's (its 10-row task pool on the box's host threads), bit for bit, every pixel.
    C3's material has a fuzzy Reflective: the oracle runs in the device's transcendental mode (see the fuzzy tests for the distance to libm)."""
    O = oracle_mod
    os_ = O.Scene()
    sh = os_.add_mesh_obj(asset(mesh + ".obj"))
    if blend:
        os_.set_material(sh, [(O.MAT_BLEND, (0, 0, 0), 0.5, 1, 2), (O.MAT_REFLECTIVE, (1, 1, 1), 0.2, 0, 0), (O.MAT_DIFFUSE, (1, 1, 1), 0, 0, 0)])
        os_.set_unitvec_mode(O.UNITVEC_F64)
        mat = R.SurfaceMaterial_Blend(R.SurfaceMaterial_Reflective((1, 1, 1), 0.2), R.SurfaceMaterial_Diffuse((1, 1, 1)), 0.5)
    else:
        os_.set_material(sh, [(O.MAT_DIFFUSE, (1, 1, 1), 0, 0, 0)])
        mat = R.SurfaceMaterial_Diffuse((1, 1, 1))
    ofb = O.Framebuffer(W, H)
    for p in range(passes):
        os_.render_pass_pool(ofb, depth, False, p, ns, 12345, threads=0, task_rows=10)
    oa, ob = ofb.read()
    s = gpu_scene(ctx, mesh, mat)
    fb = R.Framebuffer(ctx, W, H)
    s.render_passes(fb, 10, 0, 1, depth, None, 0, passes, ns, 12345)
    assert ctx.last_pass_pipeline() == DEFAULT_PIPELINE
    accum, argb = fb.read_float(), fb.resolve_argb()
    fb.close()
    assert (argb == ob).all()
    assert (bits(accum) == bits(oa)).all()


# ---- KdNode::Build on the device -------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", MESHES)
def test_device_built_tree_equals_the_references(ctx, name):
    """rtw_scene_commit builds the tree on the device (rtw_build_kernels.h: the reference's split decisions -- fp32 centroid mean in list order,
    strict `<`, the axis `>` cascade, the half / half fallback -- level by level, a wave per node).  Its preorder bounds and leaf order are those of
    the reference's own KdNode::Build (mesh_*.npz: dumped from the reference's pointer tree), bit for bit, and every derived layout (flat
    hierarchy levels, collapsed trees, screen bins) equals what the host build of the same mesh derives."""
    g = np.load(os.path.join(GOLDEN, "mesh_%s.npz" % name))
    out = {}
    for dev in (1, 0):
        ctx.set_option("device_build", dev)
        s = R.RayTracerScene(ctx)
        s.AddShape(R.RMeshShape.Create(asset(name + ".obj")), R.SurfaceMaterial_Diffuse())
        s.commit()
        b, skip, tri = s.mesh_nodes()
        assert (bits(b) == bits(g["tree_bounds"])).all() and (tri == g["tree_tri"]).all(), dev
        out[dev] = dict(skip=skip, depth=s.mesh_info()["max_depth"], flat=[s.mesh_flat(l) for l in range(3)],
                        bins=s.mesh_bins(1920, 1080, 16, 4), bins2=s.mesh_bins(333, 217, 16, 4))
        s.close()
    ctx.set_option("device_build", 1)
    d, h = out[1], out[0]
    assert (d["skip"] == h["skip"]).all() and d["depth"] == h["depth"]
    for l in range(3):
        assert (bits(d["flat"][l]) == bits(h["flat"][l])).all(), l
    for k in ("bins", "bins2"):
        assert (d[k][0] == h[k][0]).all() and (d[k][1] == h[k][1]).all()


@pytest.mark.gpu
def test_device_built_trees_of_awkward_meshes_render_like_host_built_ones(ctx):
    """triangle soups that exercise the split rules' corners (coincident centroids -> one-sided splits -> half / half, degenerate and axis-aligned
    triangles, a single triangle): same tree either way, same frames"""
    rng = np.random.default_rng(5)
    def soup(n, kind):
        if kind == "same":          # every centroid equal: every split is one-sided
            base = rng.random((1, 3, 3)).astype(np.float32)
            pts = np.repeat(base, n, 0)
        elif kind == "line":        # geometric spacing: very unbalanced splits, a deep tree
            pts = (rng.random((n, 3, 3)).astype(np.float32) * 0.01)
            pts[:, :, 0] += (1.5 ** -np.arange(n, dtype=np.float32))[:, None]
        else:
            pts = (rng.random((n, 3, 3)).astype(np.float32) - 0.5) * 2
            pts[::7, :, 2] = pts[::7, :1, 2]          # axis-aligned
            pts[::11, 2] = pts[::11, 1]               # degenerate
        p = pts.reshape(-1, 3)
        i = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
        nrm = np.tile(np.array([[0, 0, 1]], np.float32), (3 * n, 1))
        return R.RMeshShape.FromArrays(p, np.zeros_like(p), nrm, i, i, i)
    for n, kind in ((1, "rand"), (2, "same"), (37, "same"), (40, "line"), (300, "rand"), (1000, "rand")):
        shape = soup(n, kind)
        res = {}
        for dev in (1, 0):
            ctx.set_option("device_build", dev)
            s = R.RayTracerScene(ctx)
            s.AddShape(shape, R.SurfaceMaterial_Diffuse((0.9, 0.8, 0.7)))
            s.commit()
            b, skip, tri = s.mesh_nodes()
            a, argb = render_frame(ctx, s, 160, 90, 2, 3, 0, 11)
            res[dev] = (b, skip, tri, a, argb)
            s.close()
        ctx.set_option("device_build", 1)
        for x, y in zip(res[1], res[0]):
            assert (bits(x) == bits(y)).all() if x.dtype == np.float32 else (x == y).all(), (n, kind)


@pytest.mark.gpu
def test_native_gather_communicator_on_one_rank(ctx):
    """rtw_comm_create / rtw_gather_rows / rtw_comm_destroy with a world of one: librccl is found (the copy already in the process when PyTorch is
    loaded, else ROCm's), ncclCommInitRank succeeds on this GPU, and the gather of a one-rank world moves nothing and changes nothing.  (The
    exchange between ranks needs several GPUs: bench.py --gpus N checks it against a one-GPU replay.)"""
    import torch  # noqa: F401  (so that PyTorch's librccl is the one in the process, as in bench.py)
    s = gpu_scene(ctx, "TorusKnot", R.SurfaceMaterial_Diffuse())
    fb = R.Framebuffer(ctx, 320, 180)
    s.render_passes(fb, 10, 0, 1, 3, None, 0, 2, 2, 5)
    before = (fb.read_float(), fb.resolve_argb())
    assert len(R.Comm.unique_id()) == 128
    comm = R.Comm(ctx, 0, 1)
    comm.gather_rows(fb, 10)
    comm.gather_rows(fb, 10, argb_only=True)
    ctx.synchronize()
    after = (fb.read_float(), fb.resolve_argb())
    comm.close()
    assert (bits(before[0]) == bits(after[0])).all() and (before[1] == after[1]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("world,mode", [(2, "all"), (3, "argb")])
def test_gather_plan_between_ranks_sharing_one_gpu(ctx, tmp_path, world, mode):
    """rtw_gather_rows between real processes.  RCCL refuses two ranks on one GPU, so on this one-GPU box the transport underneath is the loopback
    stand-in of tests/support/loopback_rccl.cpp (named pipes + host copies, loaded through RTW_RCCL_LIBRARY); everything above it is the product:
    each rank renders its own 10-row tasks (the last task is ragged), the plan of rtw_gather_rows picks rows, offsets, peers and order, and
    rank 0 must end up holding, bit for bit, the frame one rank renders alone (accumulators too in mode "all"; in mode "argb" the accumulator
    rows of the other ranks stay zero on rank 0).  RCCL's own part is checked by bench.py --gpus N (gather_verified_bit_identical_to_1gpu)."""
    import subprocess
    import sys
    sup = os.path.join(os.path.dirname(os.path.abspath(__file__)), "support")
    lib = str(tmp_path / "libloopback_rccl.so")
    subprocess.run(["g++", "-O1", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(sup, "loopback_rccl.cpp"),
                    "-o", lib, "-L/opt/rocm/lib", "-lamdhip64"], check=True)
    env = dict(os.environ, RTW_RCCL_LIBRARY=lib, RTW_LOOPBACK_DIR=str(tmp_path))
    out = str(tmp_path / "rank0.npz")
    procs = [subprocess.Popen([sys.executable, os.path.join(sup, "gather_rank.py"), str(r), str(world), mode, out], env=env) for r in range(world)]
    codes = []
    for p in procs:
        try:
            codes.append(p.wait(timeout=180))
        except subprocess.TimeoutExpired:
            p.kill()
            codes.append(-9)
    assert codes == [0] * world
    msgs = [int(np.load(out + ".messages%d.npy" % r)[0]) for r in range(world)]
    assert msgs == [world - 1] + [1] * (world - 1), msgs      # ONE ncclSend per sender, one ncclRecv per peer on the root (round 2: one per 10-row task)
    got = np.load(out)
    W, H, ROWS = 200, 117, 10
    s = gpu_scene(ctx, "TorusKnot", R.SurfaceMaterial_Reflective())
    fb = R.Framebuffer(ctx, W, H)
    s.render_passes(fb, ROWS, 0, 1, 3, None, 0, 3, 2, 9)
    want_accum, want_argb = fb.read_float(), fb.resolve_argb()
    assert (got["argb"] == want_argb).all()
    if mode == "all":
        assert (bits(got["accum"]) == bits(want_accum)).all()
    else:
        row_task = (np.arange(W * H) // W) // ROWS
        mine = (row_task % world) == 0
        assert (bits(got["accum"][mine]) == bits(want_accum[mine])).all() and (got["accum"][~mine] == 0).all()


@pytest.mark.gpu
def test_bench_multirank_code_path_rehearsed_on_one_gpu(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (one process per rank under torch.distributed.run), rehearsed on this one-GPU box: every rank on
    GPU 0, gloo for the process group, the loopback stand-in for librccl under rtw_gather_rows (tools/rehearse_multirank.sh).  The line must come out
    with the gathered image verified against the one-GPU replay; RCCL's own part is the one thing this cannot show."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["bash", os.path.join(root, "tools", "rehearse_multirank.sh"), "2", "--steps", "6", "--warmup", "2"], capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and "rehearsal" in d
    assert d["gather_verified_bit_identical_to_1gpu"] is True
    assert d["gather"].startswith("rtw_gather_rows")


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_when_no_launcher_did(tmp_path):
    """`python3 bench.py --gpus 2` with WORLD_SIZE unset -- the way the driver issues its N = 1 command -- must start the two rank processes itself (before it
    touches a GPU), relay rank 0's line and exit with their status.  Rehearsed on this one-GPU box (--rehearse: every rank on GPU 0, gloo, loopback transport
    under rtw_gather_rows).  The line carries the C2 strong-scaling figures AND a `c5` block (BASELINE configs[4]) with its gather timed separately; both
    gathered images verified against the one-GPU replay."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--rehearse"], capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and "rehearsal" in d
    assert d["gather_verified_bit_identical_to_1gpu"] is True and d["gather"].startswith("rtw_gather_rows")
    assert d["gather_ms"] is not None and d["gather_ms"] >= 0
    c5 = d["c5"]
    assert "3840x2160" in c5["workload"] and c5["steps"] == 4 and c5["gather_verified_bit_identical_to_1gpu"] is True
    assert c5["gather_ms"] is not None and c5["ms_per_step"] > 0 and c5["value"] > 0


@pytest.mark.gpu
def test_workspace_follows_the_call_and_a_refusal_renders_in_smaller_groups(ctx):
    """The group workspace is sized for the groups a call actually forms (round 2 sized it for the largest group the policy could ever form, twice:
    10.6 GB to render a 1 200-triangle mesh).  C2's frame: one pass per call (rtw_render_range, the facade's default) stays under 200 MB of workspace, a
    20-pass call under 1.5 GB of device memory in all (unit-vector table included); rtw_render_reserve ahead of a call leaves the call nothing to allocate;
    and when a group's workspace is refused -- here by the context's limit, on a full device by hipMalloc -- the call renders the SAME image in smaller
    groups (rtw_context_fallbacks counts) instead of failing."""
    W, H, depth = 1920, 1080, 4
    c2 = R.Context(0)
    try:
        s = gpu_scene(c2, "TorusKnot", R.SurfaceMaterial_Diffuse())
        fb = R.Framebuffer(c2, W, H)
        assert c2.workspace_bytes() == 0
        R.ThreadWorker_Render(s, fb, 0, W * H - 1, depth, None, 0, 1, 12345)
        c2.synchronize()
        one = c2.workspace_bytes()
        assert 0 < one <= 200 << 20, one
        fb.clear()
        s.render_reserve(fb, 10, 0, 1, depth, 20, 1)
        reserved = c2.workspace_bytes()
        s.render_passes(fb, 10, 0, 1, depth, None, 0, 20, 1, 12345)
        c2.synchronize()
        assert c2.workspace_bytes() == reserved and c2.fallbacks() == 0
        assert c2.memory_bytes() <= 1500 << 20, c2.memory_bytes()
        want = (fb.read_float(), fb.resolve_argb())
        c2.trim()
        assert c2.workspace_bytes() == 0
        c2.set_option("workspace_limit_mb", 64)
        fb.clear()
        s.render_passes(fb, 10, 0, 1, depth, None, 0, 20, 1, 12345)
        c2.synchronize()
        got = (fb.read_float(), fb.resolve_argb())
        assert c2.fallbacks() > 0 and c2.workspace_bytes() <= 2 * (64 << 20), (c2.fallbacks(), c2.workspace_bytes())
        assert (bits(got[0]) == bits(want[0])).all() and (got[1] == want[1]).all()
        assert c2.last_pass_pipeline() == 4
        c2.set_option("workspace_limit_mb", 0)
    finally:
        c2.close()
