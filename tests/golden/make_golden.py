#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference.

Runs oracle/_ref/ref_harness, i.e. the reference's own translation units compiled in
place from /root/reference (oracle/Makefile), with rand() interposed so that every
stochastic decision is replayed from the counter-based generator of the oracle.  The
fixtures hold inputs + the reference's outputs only (no reference text).  Run it in
the build container only; the GPU box never regenerates fixtures.

    python tests/golden/make_golden.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes as SC  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
ASSETS = os.path.join(ROOT, "assets")
H = O.REF_HARNESS
D, DC, RF, EM, BL, CO, NU = range(7)

MATERIALS = {
    "diffuse": [(D, (1, 1, 1), 0, 0, 0)],
    "mirror": [(RF, (0.9, 0.8, 0.7), 0, 0, 0)],
    "blendfuzz": [(BL, (0, 0, 0), 0.5, 1, 2), (RF, (1, 1, 1), 0.2, 0, 0), (D, (1, 1, 1), 0, 0, 0)],
    "checker": [(DC, (0.8, 0.9, 1.0), 0.3, 0, 0)],
    "combine": [(CO, (0, 0, 0), 0, 1, 4), (BL, (0, 0, 0), 0.5, 2, 3), (RF, (0.95, 0.75, 0.1), 0, 0, 0),
                (D, (0.95, 0.75, 0.1), 0, 0, 0), (EM, (0.475, 0.375, 0.05), 0, 0, 0)],
    "null": [(NU, (0, 0, 0), 0, 0, 0)],
    "partial": [(D, (0.5, 0.0, 0.2), 0, 0, 0)],
}


def obj_path(name):
    return os.path.join(ASSETS, name + ".obj")


def run(args):
    subprocess.check_call([H] + [str(a) for a in args], stdout=subprocess.DEVNULL)


def make_rays(bounds, n, rng):
    lo, hi = bounds[:3].astype(np.float64), bounds[3:].astype(np.float64)
    ext = hi - lo
    rays = []
    tgt = lo + rng.random((n, 3)) * ext                      # camera rays through the mesh box
    o = np.tile(np.array([0, 0, 7.0]), (n, 1))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays.append(np.c_[o, d, np.full(n, 1000.0)])
    o = lo + rng.random((n, 3)) * ext                        # interior origins, random directions (secondary-like)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays.append(np.c_[o, d, rng.uniform(0.5, 1000, n)])
    k = max(n // 4, 1)
    o = lo + rng.random((k, 3)) * ext                        # axis-parallel (-z, +x): slab axes skipped
    o[:, 2] = hi[2] + 1
    rays.append(np.c_[o, np.tile([0, 0, -1.0], (k, 1)), np.full(k, 50.0)])
    o = lo + rng.random((k, 3)) * ext
    o[:, 0] = lo[0] - 1
    rays.append(np.c_[o, np.tile([1.0, 0, 0], (k, 1)), np.full(k, 50.0)])
    o = lo + rng.random((k, 3)) * ext                        # near-zero components around FLT_EPSILON
    o[:, 1] = hi[1] + 0.5
    d = np.c_[rng.uniform(-1e-6, 1e-6, k), -np.ones(k), rng.uniform(-2e-7, 2e-7, k)]
    rays.append(np.c_[o, d, np.full(k, 10.0)])
    o = lo + rng.random((k, 3)) * ext                        # short segments that end inside the mesh
    d = rng.normal(size=(k, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays.append(np.c_[o, d, rng.uniform(0.01, 0.3, k)])
    return np.concatenate(rays).astype(np.float32)


def golden_mesh(name, tmp):
    pre = os.path.join(tmp, name)
    run(["dump_mesh", obj_path(name), pre])
    f32 = lambda s: np.fromfile(pre + s, np.float32)  # noqa: E731
    i32 = lambda s: np.fromfile(pre + s, np.int32)  # noqa: E731
    np.savez_compressed(os.path.join(OUT, "mesh_%s.npz" % name),
                        counts=i32(".counts.i32"), points=f32(".points.f32").reshape(-1, 3),
                        texcoords=f32(".texcoords.f32").reshape(-1, 3), normals=f32(".normals.f32").reshape(-1, 3),
                        pidx=i32(".pidx.i32").reshape(-1, 3), tidx=i32(".tidx.i32").reshape(-1, 3),
                        nidx=i32(".nidx.i32").reshape(-1, 3), matid=i32(".matid.i32"),
                        tree_bounds=f32(".tree_bounds.f32").reshape(-1, 6), tree_tri=i32(".tree_tri.i32"),
                        shape_bounds=f32(".shape_bounds.f32"), texinfo=i32(".texinfo.i32").reshape(-1, 2))
    return f32(".shape_bounds.f32")


def golden_closest(name, bounds, n, tmp, rng):
    rays = make_rays(bounds, n, rng)
    rp, hp = os.path.join(tmp, "rays.bin"), os.path.join(tmp, "hits.bin")
    rays.tofile(rp)
    run(["closest", obj_path(name), rp, len(rays), hp])
    r = np.fromfile(hp, np.float32).reshape(-1, 13)
    np.savez_compressed(os.path.join(OUT, "closest_%s.npz" % name), rays=rays, hit=r[:, :11].copy(),
                        shape=r[:, 11].copy().view(np.int32), tri=r[:, 12].copy().view(np.int32))


def golden_texsample(name, mat, n, tmp, rng):
    uv = rng.uniform(-2.5, 2.5, (n, 2)).astype(np.float32)
    edge = np.array([[0, 0], [1, 1], [0.5, 0.5], [-1e-9, 0.25], [0.25, -1e-9], [1.0, 0.0], [0.999999, 0.999999],
                     [2.0, -3.0], [1 / 511.0, 2 / 511.0], [255 / 255.0, 17 / 255.0]], np.float32)
    uv = np.concatenate([edge, uv])
    up, op = os.path.join(tmp, "uv.bin"), os.path.join(tmp, "tex.bin")
    uv.tofile(up)
    run(["texsample", obj_path(name), mat, up, len(uv), op])
    np.savez_compressed(os.path.join(OUT, "texsample_%s_m%d.npz" % (name, mat)), uv=uv, material=np.int32(mat),
                        rgba=np.fromfile(op, np.float32).reshape(-1, 4))


def golden_frame(tag, name, mat, W, Hh, ns, depth, preview, seed, pass0, npass, tmp):
    arr = O.materials(MATERIALS[mat])
    mp, fp = os.path.join(tmp, "mat.bin"), os.path.join(tmp, "frame")
    arr.tofile(mp)
    run(["frame", obj_path(name), mp, W, Hh, ns, depth, preview, seed, pass0, npass, 0, W * Hh - 1, fp])
    np.savez_compressed(os.path.join(OUT, "frame_%s.npz" % tag), mesh=name, material=arr,
                        params=np.array([W, Hh, ns, depth, preview, seed, pass0, npass], np.int64),
                        accum=np.fromfile(fp + ".accum.f32", np.float32).reshape(-1, 4),
                        argb=np.fromfile(fp + ".argb.u32", np.uint32))


def golden_worker(tag, name, mat, depth, preview, seed, pass0, npass, tmp):
    """The reference's OWN ThreadWorker_Render (Src/RayTracerProgram.cpp:131-188; 800 x 800, 4 sub-samples) writing its own accuBuffer[] /
    bitcolor[] (harness command `worker`).  Kept: the whole ARGB image, the SHA-256 of the accumulator's bytes (count as int32) and the
    accumulator of a 64-row band."""
    import hashlib
    W = Hh = 800
    arr = O.materials(MATERIALS[mat])
    mp, fp = os.path.join(tmp, "mat.bin"), os.path.join(tmp, "worker")
    arr.tofile(mp)
    run(["worker", obj_path(name), mp, depth, preview, seed, pass0, npass, 0, W * Hh - 1, fp])
    accum = np.fromfile(fp + ".accum.f32", np.float32).reshape(-1, 4)
    ints = accum.copy()
    ints[:, 3] = accum[:, 3].astype(np.int32).view(np.float32)          # the count as the int the reference holds
    band = slice(368 * W, 432 * W)
    np.savez_compressed(os.path.join(OUT, "worker_%s.npz" % tag), mesh=name, material=arr,
                        params=np.array([W, Hh, 4, depth, preview, seed, pass0, npass], np.int64),
                        argb=np.fromfile(fp + ".argb.u32", np.uint32), accum_sha256=hashlib.sha256(ints.tobytes()).hexdigest(),
                        band=np.array([368, 432], np.int64), accum_band=accum[band])


def golden_misc(tmp):
    """Small facts straight from the reference's own functions: FormatTimeString's renderings (Src/RayTracerProgram.cpp:242-268) and a PNG
    written by RTexture::SaveBufferToPNG (Src/Texture.cpp:201-283) with its decoded pixels."""
    import json
    from PIL import Image
    ms = [0, 1, 999, 1000, 1001, 59999, 60000, 61000, 125000, 3599999, 3600000, 3661000, 3725000, 86400000 + 62000]
    out = subprocess.check_output([H, "timestring"] + [str(v) for v in ms]).decode().strip().splitlines()
    table = {int(ln.split(" ", 1)[0]): ln.split(" ", 1)[1] for ln in out}
    json.dump(table, open(os.path.join(OUT, "timestrings.json"), "w"), indent=0, sort_keys=True)
    g = np.load(os.path.join(OUT, "frame_unitychan_diffuse_d4.npz"))
    W, Hh = int(g["params"][0]), int(g["params"][1])
    argb = g["argb"].astype(np.uint32)
    ap, pp = os.path.join(tmp, "img.argb"), os.path.join(OUT, "png_written_by_reference.png")
    argb.tofile(ap)
    run(["savepng", ap, W, Hh, pp])
    np.savez_compressed(os.path.join(OUT, "png_reference_decode.npz"), argb=argb, size=np.array([W, Hh]), rgb=np.asarray(Image.open(pp).convert("RGB")))


def main_worker(tmp):
    golden_worker("torus_mirror_d4", "TorusKnot", "mirror", 4, 0, 31, 0, 2, tmp)
    golden_worker("unitychan_preview", "unitychan", "diffuse", 4, 1, 31, 0, 1, tmp)
    golden_worker("unitychan_mirror_d3", "unitychan", "mirror", 3, 0, 77, 1, 1, tmp)


def golden_raytrace(tag, name, mat, n, depth, seed, W, Hh, tmp, rng, bounds):
    rays = make_rays(bounds, n, rng)
    keys = np.c_[rng.integers(0, W * Hh, len(rays)), rng.integers(0, 8, len(rays))].astype(np.uint32)
    arr = O.materials(MATERIALS[mat])
    mp, rp, kp, op = [os.path.join(tmp, f) for f in ("mat.bin", "rays.bin", "keys.bin", "rt.bin")]
    arr.tofile(mp)
    rays.tofile(rp)
    keys.tofile(kp)
    run(["raytrace", obj_path(name), mp, rp, kp, len(rays), depth, 0, seed, W, Hh, op])
    np.savez_compressed(os.path.join(OUT, "raytrace_%s.npz" % tag), mesh=name, material=arr, rays=rays, keys=keys,
                        params=np.array([depth, seed, W, Hh], np.int64),
                        rgb=np.fromfile(op, np.float32).reshape(-1, 3))


def write_scene_file(scene, tmp, tag):
    """the harness's .scene text: shapes in insertion order, numbers as exact decimal renderings of the float32 values"""
    num = lambda v: " ".join("%.17g" % float(np.float32(x)) for x in (v if isinstance(v, (tuple, list)) else [v]))  # noqa: E731
    lines = []
    for k, sh in enumerate(scene):
        if sh[-1] is None:
            mat = "none"
        else:
            mat = os.path.join(tmp, "%s_mat%d.bin" % (tag, k))
            O.materials(sh[-1]).tofile(mat)
        if sh[0] == "sphere":
            lines.append("sphere %s %s %s" % (num(sh[1]), num(sh[2]), mat))
        elif sh[0] == "plane":
            lines.append("plane %s %s %s" % (num(sh[1]), num(sh[2]), mat))
        elif sh[0] == "capsule":
            lines.append("capsule %s %s %s %s" % (num(sh[1]), num(sh[2]), num(sh[3]), mat))
        elif sh[0] == "triangle":
            lines.append("triangle %s %s %s %s" % (num(sh[1]), num(sh[2]), num(sh[3]), mat))
        else:
            lines.append("mesh %s %s" % (obj_path(sh[1]), mat))
    path = os.path.join(tmp, tag + ".scene")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path


def golden_scene_closest(tag, n, tmp, rng, mesh_bounds):
    scene = SC.SCENES[tag]()
    sp = write_scene_file(scene, tmp, tag)
    rays = make_rays(SC.scene_bounds(scene, mesh_bounds), n, rng)
    rp, hp = os.path.join(tmp, "rays.bin"), os.path.join(tmp, "hits.bin")
    rays.tofile(rp)
    run(["closest", sp, rp, len(rays), hp])
    r = np.fromfile(hp, np.float32).reshape(-1, 13)
    np.savez_compressed(os.path.join(OUT, "sceneclosest_%s.npz" % tag), scene=tag, rays=rays, hit=r[:, :11].copy(),
                        shape=r[:, 11].copy().view(np.int32))


def golden_scene_frame(name, tag, W, Hh, ns, depth, preview, seed, pass0, npass, tmp):
    sp, fp = write_scene_file(SC.SCENES[tag](), tmp, tag), os.path.join(tmp, "frame")
    run(["frame", sp, "-", W, Hh, ns, depth, preview, seed, pass0, npass, 0, W * Hh - 1, fp])
    np.savez_compressed(os.path.join(OUT, "sceneframe_%s.npz" % name), scene=tag,
                        params=np.array([W, Hh, ns, depth, preview, seed, pass0, npass], np.int64),
                        accum=np.fromfile(fp + ".accum.f32", np.float32).reshape(-1, 4),
                        argb=np.fromfile(fp + ".argb.u32", np.uint32))


def main_scenes(tmp, rng, bounds):
    """multi-shape scenes: RSphere / RPlane / RCapsule beside meshes (tests/scenes.py)"""
    for tag in ("default_nofuzz", "quirk", "shapes", "room", "tris"):
        golden_scene_closest(tag, 700, tmp, rng, bounds)
    S = golden_scene_frame
    S("default_d5", "default", 80, 80, 4, 5, 0, 12345, 0, 1, tmp)
    S("default_nofuzz_d5", "default_nofuzz", 80, 80, 4, 5, 0, 12345, 0, 2, tmp)
    S("default_preview", "default", 80, 80, 4, 5, 1, 12345, 0, 1, tmp)
    S("quirk_d4", "quirk", 96, 96, 4, 4, 0, 77, 0, 1, tmp)
    S("quirk_preview", "quirk", 96, 96, 4, 4, 1, 77, 0, 1, tmp)
    S("shapes_d6", "shapes", 96, 64, 4, 6, 0, 9, 1, 2, tmp)
    S("shapes_1spp_d2", "shapes", 50, 70, 1, 2, 0, 9, 0, 1, tmp)
    S("room_d8", "room", 64, 64, 4, 8, 0, 21, 0, 1, tmp)
    S("tris_d5", "tris", 96, 96, 4, 5, 0, 4, 0, 1, tmp)


def main():
    O.build()
    if not os.path.exists(H):
        sys.exit("oracle/_ref/ref_harness is not built (no /root/reference here?)")
    rng = np.random.default_rng(20261004)
    if "--worker-only" in sys.argv:        # add the ThreadWorker_Render fixtures without regenerating the others
        with tempfile.TemporaryDirectory() as tmp:
            main_worker(tmp)
        return
    if "--misc-only" in sys.argv:
        with tempfile.TemporaryDirectory() as tmp:
            golden_misc(tmp)
        return
    if "--scenes-only" in sys.argv:        # add the multi-shape fixtures without regenerating the others
        rng = np.random.default_rng(20261005)
        with tempfile.TemporaryDirectory() as tmp:
            bounds = {n: np.load(os.path.join(OUT, "mesh_%s.npz" % n))["shape_bounds"] for n in ("TorusKnot", "BlenderMonkey", "unitychan")}
            main_scenes(tmp, rng, bounds)
        return
    with tempfile.TemporaryDirectory() as tmp:
        bounds = {}
        for name in ("TorusKnot", "BlenderMonkey", "unitychan"):
            bounds[name] = golden_mesh(name, tmp)
            golden_closest(name, bounds[name], 600 if name == "unitychan" else 900, tmp, rng)
        golden_texsample("unitychan", 4, 600, tmp, rng)     # 256x256 RGBA (cheek)
        golden_texsample("unitychan", 0, 600, tmp, rng)     # 512x512 RGB (skin)
        F = golden_frame
        F("torus_diffuse_4spp_d4", "TorusKnot", "diffuse", 64, 64, 4, 4, 0, 12345, 0, 1, tmp)
        F("torus_diffuse_1spp_d1", "TorusKnot", "diffuse", 64, 48, 1, 1, 0, 7, 0, 1, tmp)
        F("torus_diffuse_3pass_d10", "TorusKnot", "diffuse", 64, 64, 4, 10, 0, 12345, 0, 3, tmp)
        F("torus_preview", "TorusKnot", "diffuse", 64, 64, 4, 4, 1, 12345, 0, 1, tmp)
        F("torus_mirror_d6", "TorusKnot", "mirror", 64, 64, 4, 6, 0, 99, 0, 1, tmp)
        F("torus_blendfuzz_d6", "TorusKnot", "blendfuzz", 64, 64, 4, 6, 0, 99, 0, 1, tmp)
        F("monkey_blendfuzz_d6", "BlenderMonkey", "blendfuzz", 96, 54, 4, 6, 0, 5, 1, 2, tmp)
        F("torus_checker", "TorusKnot", "checker", 64, 64, 2, 4, 0, 3, 0, 1, tmp)
        F("torus_checker_preview", "TorusKnot", "checker", 64, 64, 4, 4, 1, 3, 0, 1, tmp)
        F("torus_combine", "TorusKnot", "combine", 64, 64, 4, 5, 0, 3, 0, 1, tmp)
        F("torus_null", "TorusKnot", "null", 64, 64, 4, 5, 0, 3, 0, 1, tmp)
        F("torus_partial_albedo", "TorusKnot", "partial", 64, 64, 4, 5, 0, 3, 0, 1, tmp)
        F("unitychan_diffuse_d4", "unitychan", "diffuse", 96, 96, 4, 4, 0, 12345, 0, 1, tmp)
        F("unitychan_preview", "unitychan", "diffuse", 96, 96, 4, 4, 1, 12345, 0, 1, tmp)
        golden_raytrace("unitychan_diffuse", "unitychan", "diffuse", 500, 6, 4242, 1920, 1080, tmp, rng, bounds["unitychan"])
        golden_raytrace("monkey_blendfuzz", "BlenderMonkey", "blendfuzz", 900, 6, 4242, 1920, 1080, tmp, rng, bounds["BlenderMonkey"])
        main_scenes(tmp, np.random.default_rng(20261005), bounds)
        main_worker(tmp)
        golden_misc(tmp)
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.endswith(".npz"))
    print("golden fixtures written: %.2f MB" % (total / 1e6))


if __name__ == "__main__":
    main()
