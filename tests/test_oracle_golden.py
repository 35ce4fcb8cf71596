"""The CPU oracle (oracle/rt_oracle.c) against the golden vectors produced by the REAL
reference (oracle/_ref/ref_harness, see tests/golden/make_golden.py).  Bit-exact."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, asset

MESHES = ["TorusKnot", "BlenderMonkey", "unitychan"]
_scenes = {}


def scene_for(O, name):
    if name not in _scenes:
        s = O.Scene()
        sh = s.add_mesh_obj(asset(name + ".obj"))
        _scenes[name] = (s, sh)
    return _scenes[name]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("name", MESHES)
def test_parser_and_tree_match_reference(oracle_mod, name):
    g = np.load(os.path.join(GOLDEN, "mesh_%s.npz" % name))
    s, sh = scene_for(oracle_mod, name)
    c = s.counts(sh)
    assert [c["points"], c["texcoords"], c["normals"], c["tris"]] == g["counts"][:4].tolist()
    assert c["nodes"] == g["counts"][5] == 2 * c["tris"] - 1
    a = s.mesh_arrays(sh)
    for k in ("points", "texcoords", "normals"):
        assert (bits(a[k]) == bits(g[k])).all(), k
    for k in ("pidx", "tidx", "nidx", "matid"):
        assert (a[k] == g[k]).all(), k
    b, t = s.tree_preorder(sh)
    assert (bits(b) == bits(g["tree_bounds"])).all()
    assert (t == g["tree_tri"]).all()
    assert (bits(s.shape_bounds(sh)) == bits(g["shape_bounds"])).all()


def test_topology_facts_from_survey(oracle_mod):
    # nodes 2399/1935/32111; material ids -1 / 0 / 0..8
    expect = {"TorusKnot": (2399, -1, -1), "BlenderMonkey": (1935, 0, 0), "unitychan": (32111, 0, 8)}
    for name, (nodes, lo, hi) in expect.items():
        s, sh = scene_for(oracle_mod, name)
        assert s.counts(sh)["nodes"] == nodes
        m = s.mesh_arrays(sh)["matid"]
        assert (m.min(), m.max()) == (lo, hi)


@pytest.mark.parametrize("name", MESHES)
def test_closest_hit_matches_reference(oracle_mod, name):
    g = np.load(os.path.join(GOLDEN, "closest_%s.npz" % name))
    s, sh = scene_for(oracle_mod, name)
    hf, hs, ht = s.trace_closest(g["rays"])
    assert (hs == g["shape"]).all()
    hit = hs >= 0
    assert hit.sum() > 100
    assert (ht[hit] == g["tri"][hit]).all()
    assert (bits(hf[hit]) == bits(g["hit"][hit])).all()


@pytest.mark.parametrize("mat", [0, 4])
def test_texture_sample_matches_reference(oracle_mod, mat):
    g = np.load(os.path.join(GOLDEN, "texsample_unitychan_m%d.npz" % mat))
    s, sh = scene_for(oracle_mod, "unitychan")
    out = s.texture_sample(sh, mat, g["uv"])
    assert (bits(out) == bits(g["rgba"])).all()


FRAMES = sorted(os.path.basename(p)[6:-4] for p in glob.glob(os.path.join(GOLDEN, "frame_*.npz")))


@pytest.mark.parametrize("tag", FRAMES)
def test_frame_matches_reference(oracle_mod, tag):
    O = oracle_mod
    g = np.load(os.path.join(GOLDEN, "frame_%s.npz" % tag))
    W, H, ns, depth, preview, seed, pass0, npass = [int(v) for v in g["params"]]
    s = O.Scene()
    sh = s.add_mesh_obj(asset(str(g["mesh"]) + ".obj"))
    s.set_material(sh, g["material"])
    fb = O.Framebuffer(W, H)
    for p in range(pass0, pass0 + npass):
        s.render_range(fb, 0, W * H - 1, depth, bool(preview), p, ns, seed)
    accum, argb = fb.read()
    assert (argb == g["argb"]).all()
    if not preview:
        assert (bits(accum) == bits(g["accum"])).all()


@pytest.mark.parametrize("tag", ["unitychan_diffuse", "monkey_blendfuzz"])
def test_ray_trace_matches_reference(oracle_mod, tag):
    O = oracle_mod
    g = np.load(os.path.join(GOLDEN, "raytrace_%s.npz" % tag))
    depth, seed, W, H = [int(v) for v in g["params"]]
    s = O.Scene()
    sh = s.add_mesh_obj(asset(str(g["mesh"]) + ".obj"))
    s.set_material(sh, g["material"])
    rgb = s.ray_trace(g["rays"], g["keys"], depth, False, seed, W, H)
    assert (bits(rgb) == bits(g["rgb"])).all()


def test_f64_unit_vector_mode_is_close_to_libm(oracle_mod):
    """The device evaluates the fuzzy-reflection direction with double transcendentals
    (ORC_UNITVEC_F64); it must stay within a few ulp of the reference's libm floats."""
    O = oracle_mod
    g = np.load(os.path.join(GOLDEN, "frame_torus_blendfuzz_d6.npz"))
    W, H, ns, depth, preview, seed, pass0, npass = [int(v) for v in g["params"]]
    s = O.Scene()
    sh = s.add_mesh_obj(asset("TorusKnot.obj"))
    s.set_material(sh, g["material"])
    s.set_unitvec_mode(O.UNITVEC_F64)
    fb = O.Framebuffer(W, H)
    s.render_range(fb, 0, W * H - 1, depth, False, pass0, ns, seed)
    accum, _ = fb.read()
    d = np.abs(accum[:, :3] - g["accum"][:, :3]).max(axis=1)
    assert (d <= 1e-4).mean() > 0.995


def test_pool_equals_serial(oracle_mod):
    O = oracle_mod
    s, sh = scene_for(O, "TorusKnot")
    s.set_material(sh, [(O.MAT_DIFFUSE, (1, 1, 1), 0, 0, 0)])
    a, b = O.Framebuffer(80, 45), O.Framebuffer(80, 45)
    s.render_range(a, 0, 80 * 45 - 1, 4, False, 0, 4, 11)
    s.render_pass_pool(b, 4, False, 0, 4, 11, threads=4, task_rows=10)
    (aa, ab), (ba, bb) = a.read(), b.read()
    assert (bits(aa) == bits(ba)).all() and (ab == bb).all()


# ---- multi-shape scenes: RSphere / RPlane / RCapsule beside meshes (tests/scenes.py) ----------------------------------
import scenes as SC  # noqa: E402

_multi = {}


def multi_scene(O, tag):
    if tag not in _multi:
        _multi[tag] = O.Scene().add_shapes(SC.SCENES[tag](), lambda n: asset(n + ".obj"))
    return _multi[tag]


@pytest.mark.parametrize("tag", ["default_nofuzz", "quirk", "shapes", "room", "tris"])
def test_scene_closest_hit_matches_reference(oracle_mod, tag):
    """FindIntersectionWithScene over spheres, planes, capsules and meshes in insertion order (Src/RayTracerScene.cpp:99-125),
    including the sampled colour a nearer analytic hit inherits from an earlier textured-mesh hit."""
    g = np.load(os.path.join(GOLDEN, "sceneclosest_%s.npz" % tag))
    hf, hs, _ = multi_scene(oracle_mod, tag).trace_closest(g["rays"])
    assert (hs == g["shape"]).all()
    hit = hs >= 0
    assert hit.sum() > 100 and len(set(hs[hit].tolist())) >= 3
    assert (bits(hf[hit]) == bits(g["hit"][hit])).all()
    # misses too: whatever a failed query left in the result
    assert (bits(hf) == bits(g["hit"])).all()
    if tag == "quirk":      # the quirk fires: analytic-shape hits whose sampled colour is not the default white
        analytic = hit & (hs > 0)
        assert (np.abs(hf[analytic, 7:10] - 1.0).max(axis=1) > 1e-3).sum() > 20


@pytest.mark.parametrize("name", ["default_d5", "default_nofuzz_d5", "default_preview", "quirk_d4", "quirk_preview", "shapes_d6",
                                  "shapes_1spp_d2", "room_d8", "tris_d5"])
def test_scene_frame_matches_reference(oracle_mod, name):
    """RayTracerProgram::SetupScene's scene (Src/RayTracerProgram.cpp:467-552) and two synthetic multi-shape scenes, rendered by
    the reference's RayTrace through the harness: the oracle gives the same accumulator and ARGB bits."""
    g = np.load(os.path.join(GOLDEN, "sceneframe_%s.npz" % name))
    W, H, ns, depth, preview, seed, pass0, npass = [int(v) for v in g["params"]]
    s = multi_scene(oracle_mod, str(g["scene"]))
    fb = oracle_mod.Framebuffer(W, H)
    for p in range(pass0, pass0 + npass):
        s.render_range(fb, 0, W * H - 1, depth, bool(preview), p, ns, seed)
    accum, argb = fb.read()
    assert (argb == g["argb"]).all()
    if not preview:         # a preview pass writes the picture only
        assert (bits(accum) == bits(g["accum"])).all()


WORKERS = sorted(os.path.basename(p)[7:-4] for p in glob.glob(os.path.join(GOLDEN, "worker_*.npz")))


def worker_digest(accum):
    """SHA-256 of an accumulator as the reference holds it: three floats + the int32 count per pixel"""
    import hashlib
    ints = np.ascontiguousarray(accum, np.float32).copy()
    ints[:, 3] = ints[:, 3].astype(np.int32).view(np.float32)
    return hashlib.sha256(ints.tobytes()).hexdigest()


@pytest.mark.parametrize("tag", WORKERS)
def test_oracle_vs_the_references_own_thread_worker_render(oracle_mod, tag):
    """worker_*.npz hold what the reference's OWN ThreadWorker_Render (Src/RayTracerProgram.cpp:131-188, not the harness's restatement of
    it) left in its accuBuffer[] / bitcolor[] at its compiled-in 800 x 800: the oracle's pixel loop, accumulation, gamma and packing
    (rows a1-a3 of SURVEY.md 8a) against that, bit for bit."""
    O = oracle_mod
    g = np.load(os.path.join(GOLDEN, "worker_%s.npz" % tag))
    W, H, ns, depth, preview, seed, pass0, npass = [int(v) for v in g["params"]]
    assert (W, H, ns) == (800, 800, 4)
    s = O.Scene()
    sh = s.add_mesh_obj(asset(str(g["mesh"]) + ".obj"))
    s.set_material(sh, g["material"])
    fb = O.Framebuffer(W, H)
    for p in range(pass0, pass0 + npass):
        s.render_pass_pool(fb, depth, bool(preview), p, ns, seed, threads=0, task_rows=10)
    accum, argb = fb.read()
    assert (argb == g["argb"]).all()
    if not preview:
        r0, r1 = [int(v) for v in g["band"]]
        assert (bits(accum[r0 * W:r1 * W]) == bits(g["accum_band"])).all()
        assert worker_digest(accum) == str(g["accum_sha256"])


@pytest.mark.parametrize("mesh,W,H,ns,depth", [("TorusKnot", 64, 64, 4, 6), ("BlenderMonkey", 480, 270, 4, 6)])
def test_fuzzy_reflection_modes_differ_only_where_the_hit_history_differs(oracle_mod, mesh, W, H, ns, depth):
    """The one bounded approximation of the device path: fuzzy Reflective evaluates sin / cos / acos in double (the reference: glibc's float
    routines).  The oracle has both modes (LIBM == the reference bit for bit, F64 == the GPU bit for bit).  Rendered with a per-pixel signature of
    the hit / miss history of the pixel's paths: wherever the two modes took the same branches the colours agree to 2e-6 per channel, and a
    pixel that deviates by more than the stated tolerance of 1e-4 is one whose paths took a different branch (a last-bit difference of a
    direction turned a hit into a miss) -- never a shading difference.  Observed on the fixtures' frames: no such pixel at all."""
    O = oracle_mod
    mat = [(O.MAT_BLEND, (0, 0, 0), 0.5, 1, 2), (O.MAT_REFLECTIVE, (1, 1, 1), 0.2, 0, 0), (O.MAT_DIFFUSE, (1, 1, 1), 0, 0, 0)]
    out = {}
    for mode in (O.UNITVEC_LIBM, O.UNITVEC_F64):
        s = O.Scene()
        sh = s.add_mesh_obj(asset(mesh + ".obj"))
        s.set_material(sh, mat)
        s.set_unitvec_mode(mode)
        fb = O.Framebuffer(W, H)
        hist = np.zeros(W * H, np.uint32)
        s.render_range(fb, 0, W * H - 1, depth, False, 0, ns, 99, history=hist)
        out[mode] = (fb.read()[0], hist)
    (a, ha), (b, hb) = out[O.UNITVEC_LIBM], out[O.UNITVEC_F64]
    d = np.abs(a[:, :3] - b[:, :3]).max(axis=1)
    same = ha == hb
    assert d[same].max() <= 2e-6
    assert same[d > 1e-4].sum() == 0            # every deviation beyond the tolerance is a hit / miss flip
    print("fuzzy %s %dx%d: %d of %d pixels differ at all, %d beyond 1e-4 (all with a different hit history), max %.3g" %
          (mesh, W, H, int((d > 0).sum()), W * H, int((d > 1e-4).sum()), d.max()))
