#!/usr/bin/env python3
"""Randomised cross-check of the render pipelines (test infrastructure, needs a GPU).

  python tools/soak.py [--seed S] [--cases N]

Every case draws a scene (one of the three OBJ assets, two of them together, random spheres / planes / capsules alone or around a
mesh, or a random triangle soup with degenerate, axis-aligned, coincident and behind-the-camera triangles), a material tree, a frame shape (including tall / narrow / odd sizes),
spp, depth, preview flag, passes, a rank split and random pipeline options, renders it through the one-thread-per-pixel kernel
(pipeline 0, the most literal reading of ThreadWorker_Render, Src/RayTracerProgram.cpp:130-189) and through the default bins + wave
pipeline, and compares accumulator and ARGB bits.  Prints one line per case and the number of mismatches; exit code 1 on any."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracerwin_amd as R  # noqa: E402


def soup(rng, kind):
    n = int(rng.choice([1, 2, 7, 33, 200, 1500]))
    ctr = rng.uniform(-2.0, 2.0, (n, 1, 3)).astype(np.float32)
    if kind == "behind":                 # some triangles at or behind the camera plane z = 7: no screen bins for the shape
        ctr[:, 0, 2] = rng.uniform(-2.0, 9.0, n)
    tri = ctr + rng.normal(0, rng.choice([0.05, 0.4, 2.0]), (n, 3, 3)).astype(np.float32)
    if kind == "flat":                   # axis-aligned sheets: zero-thickness leaf boxes
        tri[:, :, int(rng.integers(0, 3))] = np.round(ctr[:, :, int(rng.integers(0, 3))] * 2) / 2
    if kind == "degenerate" and n > 2:
        tri[::3, 2] = tri[::3, 1]        # zero-area
        tri[1::5] = tri[0]               # coincident copies
    pts = tri.reshape(-1, 3).astype(np.float32)
    idx = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    nrm = rng.normal(0, 1, (3 * n, 3)).astype(np.float32)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-6)
    tcs = rng.uniform(0, 1, (3 * n, 3)).astype(np.float32)
    return R.RMeshShape.FromArrays(pts, tcs, nrm, idx, idx, idx)


def analytic(rng):
    k = int(rng.integers(0, 4))
    v = lambda lo, hi: tuple(float(x) for x in rng.uniform(lo, hi, 3))  # noqa: E731
    if k == 3:
        return R.RTriangle.Create(v(-3, 3), v(-3, 3), v(-3, 3))
    if k == 0:
        return R.RSphere.Create(v(-3, 3), float(rng.choice([0.05, 0.5, 1.5, 8.0])))       # 8.0: the camera may sit inside
    if k == 1:
        n = np.asarray(v(-1, 1)); n /= max(np.linalg.norm(n), 1e-6)
        return R.RPlane.Create(tuple(float(x) for x in (n if rng.random() < 0.7 else n * 3)), v(-3, 3))
    a = v(-3, 3)
    return R.RCapsule.Create(a, a if rng.random() < 0.1 else v(-3, 3), float(rng.choice([0.1, 0.4, 1.0])))


def material(rng, depth=0):
    k = int(rng.integers(0, 7 if depth < 2 else 4))
    col = tuple(float(v) for v in rng.uniform(0.2, 1.0, 3))
    if k == 0:
        return R.SurfaceMaterial_Diffuse(col)
    if k == 1:
        return R.SurfaceMaterial_DiffuseChecker(col, float(rng.choice([0.5, 2.0, 5.0])))
    if k == 2:
        return R.SurfaceMaterial_Reflective(col, 0.0)
    if k == 3:
        return R.SurfaceMaterial_Emissive(col)
    if k == 4:
        return R.SurfaceMaterial_Blend(material(rng, depth + 1), material(rng, depth + 1), float(rng.choice([0.0, 0.3, 0.5, 1.0])))
    if k == 5:
        return R.SurfaceMaterial_Combine(material(rng, depth + 1), material(rng, depth + 1))
    return R.SurfaceMaterial_Null()


DEFAULTS = dict(pipeline=4, group_max=256, wave_below=80000, visit_budget=384, group_split=1, split_min=8, split_paths=400000, workspace_limit_mb=0, primary_passes=0)


def run(seed_arg, cases, ctx=None, log=print, only=-1, pipelines=(0, 3, 4), keep=None, override=None):
    rng = np.random.default_rng(seed_arg)
    ctx = ctx or R.Context(0)
    objs = ("TorusKnot", "BlenderMonkey", "unitychan")
    bad = 0
    for it in range(cases):
        kind = str(rng.choice(["obj", "obj", "obj", "two", "soup", "flat", "degenerate", "behind", "shapes", "shapes", "mixed", "mixed"]))
        s = R.RayTracerScene(ctx)
        if kind in ("shapes", "mixed"):       # spheres / planes / capsules, alone or around a mesh (before it: no texel is inherited;
            order = ["a"] * int(rng.integers(1, 6))     # after a textured one: the library takes the single kernel for both runs)
            if kind == "mixed":
                order.insert(int(rng.integers(0, len(order) + 1)), "m")
            for o in order:
                if o == "a":
                    s.AddShape(analytic(rng), material(rng) if rng.random() < 0.9 else None)
                else:
                    s.AddShape(R.RMeshShape.Create(os.path.join(ROOT, "assets", "%s.obj" % rng.choice(objs))), material(rng))
        elif kind in ("obj", "two"):
            for m in rng.choice(objs, 2 if kind == "two" else 1, replace=False):
                s.AddShape(R.RMeshShape.Create(os.path.join(ROOT, "assets", "%s.obj" % m)), material(rng) if rng.random() < 0.8 else None)
        else:
            s.AddShape(soup(rng, kind), material(rng))
            if rng.random() < 0.3:
                s.AddShape(soup(rng, "soup"), material(rng))
        s.commit()
        W = int(rng.choice([1, 7, 16, 37, 48, 64, 96, 100, 128, 160, 200, 256, 333, 384, 512, 640, 1024, 1920]))
        H = int(rng.choice([1, 5, 36, 54, 64, 90, 100, 108, 128, 211, 256, 360, 450, 600, 1080]))
        spp, depth = int(rng.integers(1, 5)), int(rng.integers(0, 9))
        prev, seed, npass = int(rng.random() < 0.15), int(rng.integers(1, 1 << 30)), int(rng.integers(1, 9))
        world, rows = int(rng.choice([1, 1, 2, 3, 8])), int(rng.choice([10, 10, 7, 16, 1]))
        opts = dict(batch=int(rng.random() < 0.7))        # pipeline 3: its passes as one rtw_render_passes run, or call by call
        # the pass-batched pipeline's switches: passes per group, the wave-per-ray threshold, the visit budget, split groups, a workspace limit that forces smaller groups
        gopts = dict(group_max=int(rng.choice([1, 2, 4, 64, 256])), wave_below=int(rng.choice([0, 2000, 100000, 10000000])), visit_budget=int(rng.choice([0, 16, 384])),
                     group_split=int(rng.random() < 0.8), split_min=int(rng.choice([2, 2, 4, 8])), split_paths=int(rng.choice([0, 0, 400000])),
                     workspace_limit_mb=int(rng.choice([0, 0, 0, 1, 8])),
                     primary_passes=[0, -1, 2, 4][(seed >> 3) & 3])       # (from the case's seed: no extra draw, so the cases of a soak seed stay what they were)
        if override:
            gopts.update(override)
        if only >= 0 and it != only:        # replay of one case: the others only advance the generator
            for rank in range(world):
                rng.random()
            s.close()
            continue
        if os.environ.get("RTW_SOAK_VERBOSE"):
            log("case", it, kind, W, H, "spp", spp, "d", depth, "prev", prev, "passes", npass, "world", world, "rows", rows, opts, gopts, flush=True)
        res = []
        for pl in pipelines:
            ctx.set_option("pipeline", pl)
            for k, v in gopts.items():
                ctx.set_option(k, v if pl == 4 else DEFAULTS[k])
            fb = R.Framebuffer(ctx, W, H)
            for rank in range(world):
                if (pl == 3 and opts["batch"]) or (pl == 4 and rng.random() < 0.8):
                    s.render_passes(fb, rows, rank, world, depth, R.RenderOption(bool(prev)), 0, npass, spp, seed)
                else:
                    for p in range(npass):
                        s.render_tasks(fb, rows, rank, world, depth, R.RenderOption(bool(prev)), p, spp, seed)
            res.append((fb.read_float().view(np.uint32).copy(), fb.resolve_argb().copy()))
            if keep is not None:
                keep[pl] = res[-1]
                keep["case"] = dict(W=W, H=H, spp=spp, depth=depth, preview=prev, seed=seed, passes=npass, world=world, rows=rows)
            fb.close()
        s.close()
        def same_acc(a, b):         # bit-identical, except that any NaN equals any NaN (sign and payload of a NaN are not specified; the tests' rule)
            fa, fb = a.view(np.float32), b.view(np.float32)
            return bool(((a == b) | (np.isnan(fa) & np.isnan(fb))).all())
        ok = all(same_acc(res[0][0], r[0]) and bool((res[0][1] == r[1]).all()) for r in res[1:])
        bad += not ok
        log(it, kind, W, H, "spp", spp, "d", depth, "prev", prev, "passes", npass, "world", world, "rows", rows, opts, gopts, "OK" if ok else "DIFF", flush=True)
        if not ok:       # which pipeline, how many pixels, where
            for pl, r in zip((3, 4), res[1:]):
                da = ((res[0][0] != r[0]) & ~(np.isnan(res[0][0].view(np.float32)) & np.isnan(r[0].view(np.float32)))).reshape(-1, 4).any(axis=1); db = res[0][1] != r[1]
                first = int(np.flatnonzero(da | db)[0]) if (da | db).any() else -1
                log("   pipeline", pl, "vs 0: accumulator pixels", int(da.sum()), "ARGB pixels", int(db.sum()), "first pixel", first,
                    "(x %d y %d)" % (first % W, first // W) if first >= 0 else "", flush=True)
    for k, v in DEFAULTS.items():
        ctx.set_option(k, v)
    log("soak done, seed", seed_arg, "mismatches:", bad)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--only", type=int, default=-1, help="replay this case alone (the others only advance the generator)")
    a = ap.parse_args()
    return 1 if run(a.seed, a.cases, only=a.only) else 0


if __name__ == "__main__":
    sys.exit(main())
