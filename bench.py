#!/usr/bin/env python3
"""bench.py -- throughput of the ray-trace hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one full frame: every camera ray of the frame through
ThreadWorker_Render's GPU equivalent (ray generation, tree traversal, triangle tests, shading,
bounce loop, accumulation, gamma/ARGB resolve).  Workload at N=1 = BASELINE.json configs[1]:
TorusKnot.obj, 1920x1080, 1 spp, depth 4, Diffuse(1,1,1), one mesh in the scene, reference camera.
For N>1 the SAME frame is split into the reference's 10-row tasks dealt round-robin over the ranks
(strong scaling); each rank accumulates its own rows and the rows are gathered to rank 0 over
RCCL once, inside the timed region, after the K passes (the reference also writes its image once,
after all passes).  Inputs are resident in HBM before the timed region starts.

Prints ONE JSON line (rank 0).  `roofline` prices the render kernel against HBM with ALGORITHMIC
bytes from reference-faithful visit counters; `cpu_baseline` times the reference's CPU path
(oracle/_ref, the reference's own sources) or the oracle port on the host cores.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (mesh, width, height, spp, depth, material)
    "c2": ("TorusKnot", 1920, 1080, 1, 4, "diffuse"),
    "c3": ("BlenderMonkey", 1920, 1080, 4, 6, "blend"),
    "c4": ("unitychan", 1920, 1080, 4, 4, "diffuse"),
    # RayTracerProgram::SetupScene (4 spheres, capsule, ground plane, unitychan) at the reference's window size and bounce limit
    "setup": ("unitychan", 800, 800, 4, 10, "setup"),
}
SEED = 12345
TASK_ROWS = 10          # NumTaskRows, Src/RayTracerProgram.cpp:282
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def procedural_torus_knot(path, segs=100, sides=6):
    """(2,3) torus knot tube: 600 vertices / 1200 triangles, same topology class as TorusKnot.obj."""
    t = np.linspace(0, 2 * np.pi, segs, endpoint=False)
    c = np.stack([(0.6 + 0.25 * np.cos(3 * t)) * np.cos(2 * t), (0.6 + 0.25 * np.cos(3 * t)) * np.sin(2 * t),
                  0.25 * np.sin(3 * t)], 1)
    tan = np.roll(c, -1, 0) - np.roll(c, 1, 0)
    tan /= np.linalg.norm(tan, axis=1, keepdims=True)
    up = np.cross(tan, np.array([0, 0, 1.0]))
    up /= np.linalg.norm(up, axis=1, keepdims=True)
    bi = np.cross(tan, up)
    with open(path, "w") as f:
        for i in range(segs):
            for j in range(sides):
                a = 2 * np.pi * j / sides
                n = np.cos(a) * up[i] + np.sin(a) * bi[i]
                p = c[i] + 0.09 * n
                f.write("v %.4f %.4f %.4f\nvn %.4f %.4f %.4f\nvt %.4f %.4f\n" % (*p, *n, i / segs, j / sides))
        for i in range(segs):
            for j in range(sides):
                a, b = i * sides + j, i * sides + (j + 1) % sides
                c2, d = ((i + 1) % segs) * sides + j, ((i + 1) % segs) * sides + (j + 1) % sides
                for tri in ((a, c2, b), (b, c2, d)):
                    f.write("f " + " ".join("%d/%d/%d" % (v + 1, v + 1, v + 1) for v in tri) + "\n")


def make_material(R, kind):
    if kind == "blend":
        return R.SurfaceMaterial_Blend(R.SurfaceMaterial_Reflective((1, 1, 1), 0.2), R.SurfaceMaterial_Diffuse((1, 1, 1)), 0.5)
    return R.SurfaceMaterial_Diffuse((1, 1, 1))


def oracle_material(O, kind):
    if kind == "blend":
        return [(O.MAT_BLEND, (0, 0, 0), 0.5, 1, 2), (O.MAT_REFLECTIVE, (1, 1, 1), 0.2, 0, 0), (O.MAT_DIFFUSE, (1, 1, 1), 0, 0, 0)]
    return [(O.MAT_DIFFUSE, (1, 1, 1), 0, 0, 0)]


def algorithmic_bytes(st, pixels):
    """Bytes one launch must touch with the layout the kernel fetches ('Roofline'):
    32 B per node visit, 64 B per triangle record, 140 B per shaded hit (64 B shading record + 64 B
    triangle record re-read + 12 B unit-vector entry), 16 B per bilinear texture sample (4 RGBA8
    texels), 36 B per pixel (16 B accumulator read + 16 B write + 4 B ARGB)."""
    return 32 * st["box_tests"] + 64 * st["tri_tests"] + 140 * st["shaded_hits"] + 16 * st["tex_samples"] + 36 * pixels


def oracle_nodes(O, m):
    """nested material tuple of raytracerwin_amd.setup_scene -> the oracle's preorder node list"""
    k = m[0]
    if k == "diffuse":
        return [(O.MAT_DIFFUSE, m[1], 0, 0, 0)]
    if k == "checker":
        return [(O.MAT_DIFFUSE_CHECKER, m[1], m[2], 0, 0)]
    if k == "reflective":
        return [(O.MAT_REFLECTIVE, m[1], m[2], 0, 0)]
    if k == "emissive":
        return [(O.MAT_EMISSIVE, m[1], 0, 0, 0)]
    a, b = oracle_nodes(O, m[1]), oracle_nodes(O, m[2])
    pair = (O.MAT_BLEND, O.MAT_COMBINE)
    shift = lambda nodes, off: [(t, c, q, x + off if t in pair else x, y + off if t in pair else y) for t, c, q, x, y in nodes]  # noqa: E731
    return [(O.MAT_BLEND if k == "blend" else O.MAT_COMBINE, (0, 0, 0), m[3] if k == "blend" else 0, 1, 1 + len(a))] + shift(a, 1) + shift(b, 1 + len(a))


def cpu_baseline_setup(mesh_path, W, H, spp, depth, rays_per_frame, budget_s):
    """SetupScene on the host cores: the reference's own translation units when oracle/_ref is there (the scene handed to the
    harness as a .scene file), else the oracle port."""
    from oracle import oracle as O
    from raytracerwin_amd.setup_scene import SHAPES
    cores = O.hw_threads()
    num = lambda v: " ".join("%.17g" % float(np.float32(x)) for x in (v if isinstance(v, (tuple, list)) else [v]))  # noqa: E731
    if os.path.exists(O.REF_HARNESS):
        with tempfile.TemporaryDirectory() as tmp:
            lines = []
            for k, sh in enumerate(SHAPES):
                mp = os.path.join(tmp, "mat%d.bin" % k)
                O.materials(oracle_nodes(O, sh[-1])).tofile(mp)
                lines.append("mesh %s %s" % (mesh_path, mp) if sh[0] == "mesh" else "%s %s %s" % (sh[0], " ".join(num(v) for v in sh[1:-1]), mp))
            sp = os.path.join(tmp, "setup.scene")
            open(sp, "w").write("\n".join(lines) + "\n")
            run = lambda passes: json.loads(subprocess.check_output([O.REF_HARNESS, "time", sp, "-", str(W), str(H), str(spp), str(depth), str(cores),  # noqa: E731
                                                                     str(passes), "tl"], stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1])
            probe = run(1)
            passes = int(max(2, min(200, budget_s / max(probe["mean_s"], 1e-3) * 0.5)))
            res = run(passes)
        return {"value": rays_per_frame / res["mean_s"] / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "reference", "ms_per_frame": res["mean_s"] * 1e3,
                "sample": "%d full %dx%d passes of the reference's own translation units (oracle/_ref) over SetupScene's shapes, 10-row tasks, "
                          "per-thread rand(); rays/frame taken from the GPU counters of the same frame" % (passes, W, H)}
    s = O.Scene()
    for sh in SHAPES:
        i = {"sphere": lambda: s.add_sphere(sh[1], sh[2]), "plane": lambda: s.add_plane(sh[1], sh[2]),
             "capsule": lambda: s.add_capsule(sh[1], sh[2], sh[3]), "mesh": lambda: s.add_mesh_obj(mesh_path)}[sh[0]]()
        s.set_material(i, oracle_nodes(O, sh[-1]))
    s.set_unitvec_mode(O.UNITVEC_F64)
    fb = O.Framebuffer(W, H)
    t = s.render_pass_pool(fb, depth, False, 0, spp, SEED, threads=cores, task_rows=TASK_ROWS)
    passes = int(max(1, min(100, budget_s / max(t, 1e-3))))
    O.stats_reset()
    tt = 0.0
    for p in range(passes):
        tt += s.render_pass_pool(fb, depth, False, p + 1, spp, SEED, threads=cores, task_rows=TASK_ROWS)
    st = O.stats_get()
    return {"value": st["rays"] / tt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port", "ms_per_frame": tt / passes * 1e3,
            "sample": "%d full %dx%d passes of the oracle port over SetupScene's shapes (10-row task pool)" % (passes, W, H)}


def cpu_baseline(mesh_path, W, H, spp, depth, kind, rays_per_frame, budget_s=20.0):
    """The reference's CPU path on the host cores, bounded to roughly `budget_s` seconds of wall time."""
    from oracle import oracle as O
    if kind == "setup":
        return cpu_baseline_setup(mesh_path, W, H, spp, depth, rays_per_frame, budget_s)
    cores = O.hw_threads()
    harness = O.REF_HARNESS
    if os.path.exists(harness) and kind == "diffuse":
        with tempfile.TemporaryDirectory() as tmp:
            mp = os.path.join(tmp, "mat.bin")
            O.materials(oracle_material(O, kind)).tofile(mp)
            probe = json.loads(subprocess.check_output([harness, "time", mesh_path, mp, str(W), str(H), str(spp), str(depth),
                                                        str(cores), "1", "tl"], stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1])
            passes = int(max(2, min(200, budget_s / max(probe["mean_s"] + 1.5 / max(1, 200), 1e-3) * 0.5)))
            # the harness spends ~1.3 s filling the unit-vector table before timing; bounded passes after it
            res = json.loads(subprocess.check_output([harness, "time", mesh_path, mp, str(W), str(H), str(spp), str(depth),
                                                      str(cores), str(passes), "tl"], stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1])
        return {"value": rays_per_frame / res["mean_s"] / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "reference",
                "ms_per_frame": res["mean_s"] * 1e3,
                "sample": "%d full %dx%d passes of the reference's own translation units (oracle/_ref: RayTrace + "
                          "ThreadTaskQueue.h, 10-row tasks, per-thread rand()); rays/frame taken from the GPU counters "
                          "of the same frame" % (passes, W, H)}
    s = O.Scene()
    sh = s.add_mesh_obj(mesh_path)
    s.set_material(sh, oracle_material(O, kind))
    s.set_unitvec_mode(O.UNITVEC_F64)
    fb = O.Framebuffer(W, H)
    t = s.render_pass_pool(fb, depth, False, 0, spp, SEED, threads=cores, task_rows=TASK_ROWS)
    passes = int(max(1, min(100, budget_s / max(t, 1e-3))))
    O.stats_reset()
    tt = 0.0
    for p in range(passes):
        tt += s.render_pass_pool(fb, depth, False, p + 1, spp, SEED, threads=cores, task_rows=TASK_ROWS)
    st = O.stats_get()
    return {"value": st["rays"] / tt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "ms_per_frame": tt / passes * 1e3,
            "sample": "%d full %dx%d passes of the oracle port (recursive pointer tree, un-pruned DFS, 10-row task pool)" % (passes, W, H)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--prune", type=int, default=1)
    ap.add_argument("--pipeline", type=int, default=3)
    ap.add_argument("--traversal", type=int, default=1)
    ap.add_argument("--packets", type=int, default=1)
    ap.add_argument("--path-lanes", type=int, default=16)
    ap.add_argument("--path-variant", type=int, default=2)
    ap.add_argument("--depth", type=int, default=0, help="override the config's depth (experiments only)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist
    import raytracerwin_amd as R

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    mesh, W, H, spp, depth, kind = CONFIGS[args.config]
    if args.depth > 0:
        depth = args.depth
    mesh_path = os.path.join(ROOT, "assets", mesh + ".obj")
    data = "synthetic frame of assets/%s.obj (byte copy of the reference's Data file)" % mesh
    if not os.path.exists(mesh_path):
        mesh_path = os.path.join(tempfile.gettempdir(), "rtw_procedural_knot_%d.obj" % rank)
        procedural_torus_knot(mesh_path)
        data = "synthetic frame of a procedural (2,3) torus knot, 600 vertices / 1200 triangles (asset missing)"
    npix = W * H

    stream = torch.cuda.Stream(device=dev)
    ctx = R.Context(local_rank, stream=stream.cuda_stream)
    ctx.set_option("pipeline", args.pipeline)
    ctx.set_option("packets", args.packets)
    ctx.set_option("path_lanes", args.path_lanes)
    ctx.set_option("path_variant", args.path_variant)
    scene = R.RayTracerScene(ctx)
    if kind == "setup":
        from raytracerwin_amd.setup_scene import SetupScene
        SetupScene(scene, mesh_path)
    else:
        scene.AddShape(R.RMeshShape.Create(mesh_path), make_material(R, kind))
    scene.set_prune(args.prune)
    scene.set_traversal(args.traversal)
    scene.commit()

    with torch.cuda.stream(stream):
        accum = torch.zeros(npix * 4, dtype=torch.float32, device=dev)
        argb = torch.zeros(npix, dtype=torch.int32, device=dev)
    fb = R.Framebuffer(ctx, W, H, accum.data_ptr(), argb.data_ptr())

    from raytracerwin_amd import sharding

    def steps(first, n):
        # n steps = n passes of the reference's sample loop (UpdateBitmapPixels, Src/RayTracerProgram.cpp:317-361) over this rank's
        # 10-row tasks: one rtw_render_passes call, pass indices first .. first + n - 1
        scene.render_passes(fb, TASK_ROWS, rank, world, depth, None, first, n, spp, SEED)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    K, Wm = args.steps, args.warmup
    with torch.cuda.stream(stream):
        steps(0, Wm)
    barrier()
    accum.zero_()
    argb.zero_()
    barrier()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gathered = None
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        ev0.record(stream)
        steps(0, K)
        ev1.record(stream)
        if world > 1:
            # the one exchange of the path: every rank's rows of (accumulator, ARGB) to rank 0 over RCCL
            sharding.gather_rows([accum.view(H, W * 4), argb.view(H, W)], H, TASK_ROWS, rank, world, dist, dev)
            gathered = True
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / K           # HIP events on the launch stream: average render-kernel duration
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    result = None
    if rank == 0:
        final_accum = accum.cpu().numpy().view(np.uint32).copy()
        final_argb = argb.cpu().numpy().copy()
        # untimed replay on one GPU: exact work counters of the K timed passes and the reference image for N>1
        ctx.stats_enable(True)
        ctx.stats_reset()
        fb2 = R.Framebuffer(ctx, W, H)
        for i in range(K):
            R.ThreadWorker_Render(scene, fb2, 0, npix - 1, depth, None, i, spp, SEED)
        st = ctx.stats()
        verified = None
        if world > 1:
            a2 = fb2.read_float()
            a2[:, 3] = a2[:, 3].astype(np.int32).view(np.float32)      # count back to int bits
            verified = bool((a2.view(np.uint32).ravel() == final_accum).all() and (fb2.resolve_argb().view(np.int32) == final_argb).all())
        # per-kernel durations: HIP events recorded by the library on the launch stream around the three kernels of
        # a pass (untimed extra passes on the scratch framebuffer; the timed region above stays exactly K steps)
        ctx.stats_enable(False)
        ctx.set_option("kernel_timing", 1)
        kms = []
        if args.pipeline >= 1:
            for i in range(30):
                R.ThreadWorker_Render(scene, fb2, 0, npix - 1, depth, None, K + i, spp, SEED)
                if i >= 5:
                    kms.append(ctx.last_pass_kernel_ms())
        ctx.set_option("kernel_timing", 0)
        ctx.stats_enable(True)
        kernel_parts = [float(np.mean([k[j] for k in kms])) for j in range(3)] if kms else None
        # reference-faithful visit counts (un-pruned order) of ONE pass for the algorithmic byte count
        scene.set_prune(0)
        scene.set_traversal(0)
        ctx.stats_reset()
        fb2.clear()
        R.ThreadWorker_Render(scene, fb2, 0, npix - 1, depth, None, 0, spp, SEED)
        st_ref = ctx.stats()
        scene.set_prune(args.prune)
        scene.set_traversal(args.traversal)
        ctx.stats_reset()
        R.ThreadWorker_Render(scene, fb2, 0, npix - 1, depth, None, 0, spp, SEED)
        st_run = ctx.stats()
        ctx.stats_enable(False)
        fb2.close()

        rays_total = st["rays"]
        value = rays_total / elapsed / 1e6
        alg_bytes = algorithmic_bytes(st_ref, npix) / world
        run_bytes = algorithmic_bytes(st_run, npix) / world
        # the three launches of one rtw_render_* call are priced together ("render pass"); at N=1 their event
        # durations are summed, for N>1 the whole-loop event time per step is used
        # "launch duration" of a render pass = HIP events on the launch stream around the K timed steps, divided by K
        # (the pass is several dependent launches; the per-stage event durations below are reported beside it -- recording
        # events between the stages perturbs them, so their sum is not used)
        pass_ms = kernel_ms
        achieved = alg_bytes / (pass_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
        if os.path.exists(tp) and world == 1:
            try:
                traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "Mrays/s (rays = closest-hit scene queries, primary + secondary) at %dx%d depth %d" % (W, H, depth),
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": data,
            "config": {"workload": ("RayTracerProgram::SetupScene (4 spheres, capsule, ground plane, %s.obj) %dx%d %d spp depth %d, reference camera "
                                    "(the reference's default scene and window; not a BASELINE config)" % (mesh, W, H, spp, depth)) if kind == "setup" else
                                   "%s.obj %dx%d %d spp depth %d, %s, 1 mesh, reference camera (BASELINE configs[%s])"
                                   % (mesh, W, H, spp, depth, kind, {"c2": 1, "c3": 2, "c4": 3}[args.config]),
                       "sharding": "10-row tasks round-robin over ranks, one RCCL gather of the rows after the K passes",
                       "seed": SEED, "prune": args.prune, "pipeline": args.pipeline, "traversal": args.traversal, "packets": args.packets, "path_lanes": args.path_lanes},
            "camera_Mrays_per_s": st["camera_rays"] / elapsed / 1e6,
            "rays_per_frame": rays_total / K,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "kernel": ("render pass = primary_bins_kernel + per-bounce shade_kernel / trace_wave_kernel rounds + resolve_kernel (one rtw_render_tasks call)" if args.pipeline == 3
                                    else "render pass = primary_kernel + path kernels + resolve_kernel (one rtw_render_tasks call)" if args.pipeline >= 1 else "render_kernel"),
                         "kernel_ms": pass_ms,
                         "stage_ms_with_events_between": dict(zip(("primary", "bounce_rounds", "resolve"), kernel_parts)) if kernel_parts else None,
                         "loop_ms_per_step_hip_events": kernel_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "executed_bytes_per_launch": run_bytes,
                         "counters_per_frame_reference_order": st_ref, "counters_per_frame_as_run": st_run},
        }
        if verified is not None:
            result["gather_verified_bit_identical_to_1gpu"] = verified
        if world == 1 and not args.no_cpu:
            try:
                result["cpu_baseline"] = cpu_baseline(mesh_path, W, H, spp, depth, kind, rays_total / K, args.cpu_budget)
            except Exception as e:      # the baseline is a reported extra, never the thing measured
                result["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": None, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    fb.close()
    scene.close()
    ctx.close()


if __name__ == "__main__":
    main()
