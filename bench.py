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

`--config c5` is BASELINE configs[4] (unitychan, 3840x2160, depth 8; a step = one 4-spp pass of its 16 spp).

`python bench.py --gpus N` WITHOUT a launcher (WORLD_SIZE unset) starts its own N rank processes (python -m
torch.distributed.run ... bench.py, before this process has touched a GPU), relays rank 0's line and exits with
their status; launched under torch.distributed.run it is one rank.  With N > 1 and the default config the line also
carries a `c5` block (BASELINE configs[4]: the frame whose tiles can pay for 8 GPUs; its gather timed separately).
`--rehearse`: every rank on GPU 0, gloo + a loopback stand-in under rtw_gather_rows -- a check of the N > 1 code path
on a one-GPU box, not a measurement.

Prints ONE JSON line (rank 0).  `roofline`: the path is LATENCY-bound (bound = "latency"); `frac` is the share of the
8 TB/s HBM roof that rocprofv3's counters saw (profiles/, stamped with the kernels' hash: printed only when the
profile was taken on the build being timed), the algorithmic figures of SURVEY.md 8(d) are labelled extras; every
fraction is computed from `ms_per_step`.  `cpu_baseline` times the reference's CPU path (oracle/_ref, the reference's
own sources) on the host cores.
"""
import argparse
import gc
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
    # BASELINE configs[4]: 16 spp = four passes of four sub-samples; `--gpus 8` splits the frame over 8 ranks + one gather
    "c5": ("unitychan", 3840, 2160, 4, 8, "diffuse"),
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


def algorithmic_bytes(st, pixel_passes, texel_bytes=4):
    """B = 32 V + 48 T + 64 H + 4 * texel_bytes * S + 36 P -- V box tests (32 B node), T triangle tests (three 12-B positions
    padded to 48), H shaded hits (three normals + three uv + material id ~ 64 B), S bilinear texture samples (four texels each; RGBA8 here),
    P pixel-passes (16 B accumulator read + 16 B write + 4 B ARGB)."""
    return 32 * st["box_tests"] + 48 * st["tri_tests"] + 64 * st["shaded_hits"] + 4 * texel_bytes * st["tex_samples"] + 36 * pixel_passes


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
    """The reference's CPU path on the host cores, bounded to roughly `budget_s` seconds of wall time: the reference's own translation
    units (oracle/_ref) through its ThreadTaskQueue with a per-thread rand() ("CPU ref"), and once more with a process-wide lock per rand()
    call like glibc's ("reference as shipped")."""
    from oracle import oracle as O
    if kind == "setup":
        return cpu_baseline_setup(mesh_path, W, H, spp, depth, rays_per_frame, budget_s)
    cores = O.hw_threads()
    harness = O.REF_HARNESS
    if os.path.exists(harness):
        with tempfile.TemporaryDirectory() as tmp:
            mp = os.path.join(tmp, "mat.bin")
            O.materials(oracle_material(O, kind)).tofile(mp)
            run = lambda passes, mode: json.loads(subprocess.check_output([harness, "time", mesh_path, mp, str(W), str(H), str(spp), str(depth),  # noqa: E731
                                                                           str(cores), str(passes), mode], stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1])
            probe = run(1, "tl")
            # the harness spends ~1.3 s filling the unit-vector table before timing; bounded passes after it
            passes = int(max(2, min(200, budget_s * 0.7 / max(probe["mean_s"], 1e-3) * 0.5)))
            res = run(passes, "tl")
            shipped = None
            try:
                sp = run(1, "glibc")
                shipped = {"ms_per_frame": sp["mean_s"] * 1e3, "Mrays_per_s": rays_per_frame / sp["mean_s"] / 1e6,
                           "sample": "1 full %dx%d pass, one process-wide lock per rand() call like glibc's rand() (the reference as shipped does not scale with cores)" % (W, H)}
            except Exception:
                shipped = None
        return {"value": rays_per_frame / res["mean_s"] / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "reference",
                "ms_per_frame": res["mean_s"] * 1e3, "as_shipped_glibc_rand": shipped,
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


def rays_through_shape_boxes(scene, W, H, spp):
    """How many camera rays (pixel-centre directions; the sub-sample jitter moves a ray by less than half a pixel) meet the culling box of
    some shape of the scene: the 'non-trivial' camera rays.  Host arithmetic on the shapes' bounds, outside the timed region."""
    xs = np.arange(W, dtype=np.float64)
    ys = np.arange(H, dtype=np.float64)
    dx = -(xs - W // 2) / (W * 2) * (W / H)
    dy = -(ys - H // 2) / (H * 2)
    d = np.stack(np.broadcast_arrays(dx[None, :], dy[:, None], np.full((H, W), -0.5)), -1).reshape(-1, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.array([0.0, 0.0, 7.0])
    hit = np.zeros(len(d), bool)
    for k in range(scene.n_shapes):
        b = scene.mesh_info(k)["bounds"].astype(np.float64)
        if not np.isfinite(b).all() or (b[3:] < b[:3]).any():
            return W * H * spp            # a shape without a culling box (a plane): every ray is tested against it
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (b[:3] - o) / d
            t2 = (b[3:] - o) / d
        hit |= np.minimum(t1, t2).max(1) < np.maximum(t1, t2).min(1)
    return int(hit.sum()) * spp


def kernels_sha(R):
    """hash of the device sources the loaded librtwin.so was built from (rtw_version(): '... kernels <sha>')"""
    v = R.library().rtw_version().decode()
    return v.split("kernels ")[1].split(")")[0].strip() if "kernels " in v else None, v


def profile_field(name, sha):
    """a figure measured under rocprofv3 by tools/profile_round.sh and kept under profiles/: returned only when that profile was taken on the very
    kernels being timed now (its recorded hash == the loaded library's); else None and the reason"""
    p = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(p):
        return None, "no %s" % os.path.relpath(p, ROOT)
    try:
        d = json.load(open(p))
    except Exception as e:
        return None, "unreadable %s: %r" % (os.path.relpath(p, ROOT), e)
    if not sha or d.get("kernels_sha") != sha:
        return None, "%s was taken on kernels %s, this library is %s: not printed" % (os.path.relpath(p, ROOT), d.get("kernels_sha"), sha)
    return d, "profiles/%s (rocprofv3 run of this build's kernels %s; an earlier run, not this one)" % (name, sha)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, BEFORE this process touches a GPU (it never does), the way the
    driver would (one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1); relay rank 0's JSON line; exit with the children's status."""
    import socket
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    tmp = None
    if args.rehearse:
        tmp = tempfile.mkdtemp(prefix="rtw_rehearse_")
        lib = os.path.join(tmp, "libloopback_rccl.so")
        subprocess.check_call(["g++", "-O1", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "support", "loopback_rccl.cpp"),
                               "-o", lib, "-L/opt/rocm/lib", "-lamdhip64"])
        env.update(RTW_RCCL_LIBRARY=lib, RTW_LOOPBACK_DIR=tmp, RTW_BENCH_REHEARSE="1")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--rehearse"]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith('{"metric"'):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if tmp:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    if line:
        print(line, flush=True)
    sys.exit(rc if rc != 0 else (0 if line else 1))


class Bench:
    """one rank's device state: context, stream, process group, communicator"""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        import raytracerwin_amd as R
        self.torch, self.dist, self.R, self.args = torch, dist, R, args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU (the product has no CPU fallback)")
        # REHEARSAL of the N > 1 code path on a box with ONE GPU (--rehearse / tools/rehearse_multirank.sh): every rank on GPU 0, gloo for the process group and the
        # loopback stand-in for librccl under rtw_gather_rows (RCCL refuses two ranks on one device).  The line says so in `rehearsal`; its throughput means nothing.
        self.rehearse = os.environ.get("RTW_BENCH_REHEARSE", "") == "1"
        if self.rehearse:
            local_rank = 0
        if local_rank >= torch.cuda.device_count():
            sys.exit("bench.py: rank %d has no GPU (%d visible); --rehearse runs every rank on GPU 0" % (self.rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        self.local_rank = local_rank
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo" if self.rehearse else "nccl", **({} if self.rehearse else {"device_id": self.dev}))
        self.stream = torch.cuda.Stream(device=self.dev)
        self.ctx = R.Context(local_rank, stream=self.stream.cuda_stream)
        self.ctx.set_option("pipeline", args.pipeline)
        if args.group_max > 0:
            self.ctx.set_option("group_max", args.group_max)
        for kv in args.option:
            k, v = kv.split("=")
            self.ctx.set_option(k, int(v))
        self.comm, self.gather_kind = None, None
        if self.world > 1 and args.gather == "native":
            self.make_comm()
        elif self.world > 1:
            self.gather_kind = "torch.distributed.gather with staging copies"

    def make_comm(self):
        # rank 0's ncclUniqueId to every rank through the process group the launcher set up; every rank takes the same decision at every step, so that a
        # failure on one of them (librccl not loadable, communicator not created) sends ALL ranks to the torch.distributed gather together
        torch, dist, R, dev = self.torch, self.dist, self.R, self.dev
        note = torch.zeros(129, dtype=torch.uint8, device=dev)
        if self.rank == 0:
            try:
                note[1:] = torch.tensor(list(R.Comm.unique_id()), dtype=torch.uint8, device=dev)
                note[0] = 1
            except Exception:
                note[0] = 0
        dist.broadcast(note, src=0)
        made = torch.zeros(1, dtype=torch.int32, device=dev)
        comm = None
        if int(note[0].item()) == 1:
            try:
                comm = R.Comm(self.ctx, self.rank, self.world, bytes(note[1:].cpu().tolist()))
                made[0] = 1
            except Exception:
                comm = None
        dist.all_reduce(made, op=dist.ReduceOp.MIN)
        if int(made.item()) == 1:
            self.comm = comm
            self.gather_kind = "rtw_gather_rows: every rank packs its task rows into one block, ONE ncclSend / ncclRecv per peer (RCCL over xGMI), rank 0 unpacks them with one launch"
        else:
            if comm is not None:
                comm.close()
            self.gather_kind = "torch.distributed.gather with staging copies (the native communicator could not be created on every rank)"

    def barrier(self):
        self.torch.cuda.synchronize(self.dev)
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize(self.dev)

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()
        if self.comm is not None:
            self.comm.close()
        self.ctx.close()


def measure(B, cfg, K, Wm, primary):
    """W warm-up steps, then EXACTLY K steps (ONE rtw_render_passes call; with N > 1 followed by the one gather) between barriers, max over ranks;
    rank 0 verifies the timed buffers against a pass-by-pass single-kernel replay and returns the block of figures (None on the other ranks).
    primary: the config the line is about (all extras); else a compact block."""
    torch, dist, R, args, ctx, dev, stream, rank, world = B.torch, B.dist, B.R, B.args, B.ctx, B.dev, B.stream, B.rank, B.world
    from raytracerwin_amd import sharding
    mesh, W, H, spp, depth, kind = CONFIGS[cfg]
    if args.depth > 0 and primary:
        depth = args.depth
    mesh_path = os.path.join(ROOT, "assets", mesh + ".obj")
    data = "synthetic frame of assets/%s.obj (byte copy of the reference's Data file)" % mesh
    if not os.path.exists(mesh_path):
        mesh_path = os.path.join(tempfile.gettempdir(), "rtw_procedural_knot_%d.obj" % rank)
        procedural_torus_knot(mesh_path)
        data = "synthetic frame of a procedural (2,3) torus knot, 600 vertices / 1200 triangles (asset missing)"
    npix = W * H

    scene = R.RayTracerScene(ctx)
    if kind == "setup":
        from raytracerwin_amd.setup_scene import SetupScene
        SetupScene(scene, mesh_path)
    else:
        scene.AddShape(R.RMeshShape.Create(mesh_path), make_material(R, kind))
    scene.set_prune(args.prune)
    t0 = time.perf_counter()
    scene.commit()              # OBJ already parsed: the reference's tree (KdNode::Build's decisions), its flat / explicit-link copies, upload
    commit_ms = (time.perf_counter() - t0) * 1e3

    with torch.cuda.stream(stream):
        accum = torch.zeros(npix * 4, dtype=torch.float32, device=dev)
        argb = torch.zeros(npix, dtype=torch.int32, device=dev)
    fb = R.Framebuffer(ctx, W, H, accum.data_ptr(), argb.data_ptr())
    argb_only = not args.gather_accum

    def steps(first, n):
        # n steps = n passes of the reference's sample loop (UpdateBitmapPixels, Src/RayTracerProgram.cpp:317-361) over this rank's
        # 10-row tasks: ONE rtw_render_passes call, pass indices first .. first + n - 1 (rendered in groups that share launches)
        scene.render_passes(fb, TASK_ROWS, rank, world, depth, None, first, n, spp, SEED)

    def gather():
        if B.comm is not None:
            B.comm.gather_rows(fb, TASK_ROWS, argb_only)
        else:
            bufs = [argb.view(H, W)] if argb_only else [accum.view(H, W * 4), argb.view(H, W)]
            sharding.gather_rows(bufs, H, TASK_ROWS, rank, world, dist, dev)

    with torch.cuda.stream(stream):
        t0 = time.perf_counter()
        steps(0, 1)             # first use of this frame shape: screen bins, tile / job tables, the workspace of a one-pass group
        torch.cuda.synchronize(dev)
        first_call_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        steps(1, 1)
        torch.cuda.synchronize(dev)
        second_call_ms = (time.perf_counter() - t0) * 1e3
        scene.render_reserve(fb, TASK_ROWS, rank, world, depth, K, spp)     # the workspace of the K-pass call's groups, outside the timed region
        if Wm > 0:
            steps(2, Wm)
        if world > 1:
            gather()            # RCCL sets its channels up on first use; the staging blocks are allocated here
        # ---- a frame that can be SHOWN after every pass (the reference's window blits bitcolor[] as the passes come, Src/RayTracerProgram.cpp:184-185,346-360):
        # separate rtw_render_passes calls of ONE pass each, launch-size hints primed, each between two HIP events on the launch stream.  Measured HERE, ahead of
        # the timed region, on every rank: a few milliseconds of the same kernels right before it.  (A stand-alone script with bench.py's sequence measured the
        # first 20-pass call after the set-up at 0.043 - 0.045 ms per pass by HIP events, after 40 such one-pass calls at 0.038 - 0.042, the calls that follow at
        # 0.037 - 0.039 either way; inside bench.py the one-shot figure still varies 0.039 - 0.044 from run to run -- see repeat_call_ms_per_step.)
        single = None
        if primary:
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
            for i in range(4):
                steps(Wm + 2 + i, 1)
            torch.cuda.synchronize(dev)
            for i in range(20):
                evs[i].record(stream)
                steps(Wm + 6 + i, 1)
            evs[20].record(stream)
            torch.cuda.synchronize(dev)
            per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(20))
            if world == 1:
                single = {"median": per[10], "min": per[0], "max": per[-1], "calls": 20}
    B.barrier()
    accum.zero_()
    argb.zero_()
    B.barrier()

    ev0, ev1, ev2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    gc.disable()        # (a collection of the interpreter's object graph -- parsed meshes, fixtures -- inside the timed region would stall the thread that enqueues the launches; no gc.collect() here: a full collection right before costs the call 80 us of cold caches)
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        ev0.record(stream)
        t_enq = time.perf_counter()
        steps(0, K)
        enqueue_ms = (time.perf_counter() - t_enq) * 1e3      # host time inside the one rtw_render_passes call (its launches are asynchronous)
        ev1.record(stream)
        if world > 1:
            gather()            # the one exchange of the path: every rank's rows to rank 0, once, after the K passes
        ev2.record(stream)
    B.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    render_ms = ev0.elapsed_time(ev1)               # HIP events on the launch stream around the K steps (this rank)
    gather_ms = ev1.elapsed_time(ev2) if world > 1 else None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / K * 1e3
    group_passes = ctx.last_group_passes()          # passes the timed call's last group held (both halves of a split group together)
    memory_bytes = ctx.memory_bytes()               # workspaces of the timed call + the unit-vector table, before the untimed replays below grow anything

    res = None
    if rank == 0:
        final_accum = accum.cpu().numpy().view(np.uint32).reshape(-1, 4).copy()
        final_argb = argb.cpu().numpy().view(np.uint32).copy()
        # ---- untimed diagnostic (N = 1): the same call three more times, host clock around call + synchronise -- what a caller's loop sees from its second
        # call on (the timed call above is the FIRST of its shape after a shorter warm-up call: launch-size hints scaled from it, clocks just back from idle)
        repeat_ms_per_step = None
        if world == 1:
            reps = []
            with torch.cuda.stream(stream):
                for i in range(3):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    steps(K * (i + 1), K)
                    torch.cuda.synchronize(dev)
                    reps.append((time.perf_counter() - t0) * 1e3 / K)
            repeat_ms_per_step = min(reps)
        # ---- untimed: the same K passes once more on this GPU alone, pass by pass through the single kernel (pipeline 0, one thread per pixel):
        # the timed buffers must hold exactly these bits (N = 1 and N > 1 alike)
        fb2 = R.Framebuffer(ctx, W, H)
        ctx.set_option("pipeline", 0)
        for i in range(K):
            R.ThreadWorker_Render(scene, fb2, 0, npix - 1, depth, None, i, spp, SEED)
        a2 = fb2.read_float()
        a2[:, 3] = a2[:, 3].astype(np.int32).view(np.float32)      # count back to int bits
        b2 = fb2.resolve_argb()
        ctx.set_option("pipeline", args.pipeline)
        same_argb = bool((b2 == final_argb).all())
        same_accum = bool((a2.view(np.uint32) == final_accum).all()) if not (world > 1 and argb_only) else None
        verified = bool(same_argb and same_accum is not False)
        # ---- work counters of the K timed passes as the timed pipeline runs them (one GPU, same grouping)
        ctx.stats_enable(True)
        ctx.stats_reset()
        fb2.clear()
        scene.render_passes(fb2, TASK_ROWS, 0, 1, depth, None, 0, K, spp, SEED)
        st = ctx.stats()
        ctx.stats_enable(False)
        rays_total, cam_total = st["rays"], st["camera_rays"]
        res = {"workload": ("RayTracerProgram::SetupScene (4 spheres, capsule, ground plane, %s.obj) %dx%d %d spp depth %d, reference camera "
                            "(the reference's default scene and window; not a BASELINE config)" % (mesh, W, H, spp, depth)) if kind == "setup" else
                           "%s.obj %dx%d %d spp depth %d, %s, 1 mesh, reference camera (BASELINE configs[%s]%s)"
                           % (mesh, W, H, spp, depth, kind, {"c2": 1, "c3": 2, "c4": 3, "c5": 4}[cfg],
                              "; a step is one 4-spp pass, four of them make the config's 16 spp" if cfg == "c5" else ""),
               "value": rays_total / elapsed / 1e6, "steps": K, "warmup": Wm, "ms_per_step": ms_per_step, "render_ms_per_step_rank0": render_ms / K, "enqueue_ms": enqueue_ms,
               "gather_ms": gather_ms, "rays_per_frame": rays_total / K, "camera_Mrays_per_s": cam_total / elapsed / 1e6,
               "verified": verified, "data": data, "depth": depth, "W": W, "H": H, "spp": spp, "kind": kind, "mesh_path": mesh_path,
               "commit_ms": commit_ms, "first_call_ms": first_call_ms, "second_call_ms": second_call_ms, "pipeline_run": ctx.last_pass_pipeline(),
               "memory_bytes": memory_bytes, "group_passes": group_passes, "repeat_ms_per_step": repeat_ms_per_step}
        if primary:
            st_run_pass = {k: v / K for k, v in st.items()}
            # reference-faithful visit counts (un-pruned DFS order: what KdNode::TestRayIntersection visits) of ONE pass
            ctx.stats_enable(True)
            scene.set_prune(0)
            scene.set_traversal(0)
            ctx.stats_reset()
            fb2.clear()
            R.ThreadWorker_Render(scene, fb2, 0, npix - 1, depth, None, 0, spp, SEED)
            st_ref = ctx.stats()
            scene.set_prune(args.prune)
            scene.set_traversal(1)
            ctx.stats_enable(False)
            # ---- per-stage durations: HIP events recorded by the library on the launch stream around the stages of one group (extra untimed passes)
            stage_ms = None
            if args.pipeline >= 3:
                ctx.set_option("kernel_timing", 1)
                acc = []
                for i in range(3):
                    fb2.clear()
                    scene.render_passes(fb2, TASK_ROWS, 0, 1, depth, None, 0, min(K, 64), spp, SEED)
                    if i > 0:
                        acc.append(ctx.last_pass_kernel_ms())
                ctx.set_option("kernel_timing", 0)
                g = float(max(1, ctx.last_group_passes()))
                stage_ms = {"passes_in_the_timed_group": int(g), "primary_per_pass": float(np.mean([k[0] for k in acc])) / g,
                            "bounce_rounds_per_pass": float(np.mean([k[1] for k in acc])) / g, "resolve_per_pass": float(np.mean([k[2] for k in acc])) / g,
                            "note": "HIP events between the stages of the LAST group of an untimed call (a call's passes are rendered in groups; the group's size follows the paths per pass; "
                                    "timed groups are not split into two halves)"}
            # ---- host vs device construction of the tree / layouts / screen bins (untimed; two fresh scenes)
            build_ms = {}
            for label, dev_build in (("device", 1), ("host", 0), ("device", 1), ("host", 0)):       # twice each, alternating; the faster run of each is kept
                ctx.set_option("device_build", dev_build)
                sc2 = R.RayTracerScene(ctx)
                if kind == "setup":
                    from raytracerwin_amd.setup_scene import SetupScene
                    SetupScene(sc2, mesh_path)
                else:
                    sc2.AddShape(R.RMeshShape.Create(mesh_path), make_material(R, kind))
                t0 = time.perf_counter()
                sc2.commit()
                t1 = time.perf_counter()
                sc2.mesh_bins(W, H, 16, 4, shape=sc2.n_shapes - 1)
                t2 = time.perf_counter()
                got = {"commit_ms": (t1 - t0) * 1e3, "screen_bins_ms": (t2 - t1) * 1e3}
                build_ms[label] = got if label not in build_ms else {k: min(got[k], build_ms[label][k]) for k in got}
                sc2.close()
            ctx.set_option("device_build", 1)
            nontrivial_cam = rays_through_shape_boxes(scene, W, H, spp)
            res.update(st_run_pass=st_run_pass, st_ref=st_ref, single=single, stage_ms=stage_ms, build_ms=build_ms, nontrivial_cam=nontrivial_cam,
                       nontrivial=nontrivial_cam * K + (rays_total - cam_total))          # every secondary ray starts on a surface, inside its shape's box
        fb2.close()
    fb.close()
    scene.close()
    del accum, argb
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=192)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--prune", type=int, default=1)
    ap.add_argument("--pipeline", type=int, default=4)
    ap.add_argument("--group-max", type=int, default=0, help="passes per group of the pass-batched pipeline (0 = the library's choice)")
    ap.add_argument("--option", action="append", default=[], help="name=value context option (experiments)")
    ap.add_argument("--depth", type=int, default=0, help="override the config's depth (experiments only)")
    ap.add_argument("--gather", default="native", choices=("native", "torch"), help="N>1: rtw_gather_rows (one RCCL message per peer) or torch.distributed.gather")
    ap.add_argument("--gather-accum", action="store_true", help="N>1: gather the float accumulator (16 B/pixel) as well as the ARGB image (4 B/pixel); the default moves what "
                                                              "the reference writes out after its passes, the image")
    ap.add_argument("--no-c5", action="store_true", help="N>1 with the default config: leave out the `c5` block")
    ap.add_argument("--rehearse", action="store_true", help="N>1 on a ONE-GPU box: every rank on GPU 0, gloo + loopback transport (a check of the code path, not a measurement)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)       # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world

    B = Bench(args)
    R, rank = B.R, B.rank
    K, Wm = args.steps, args.warmup
    m = measure(B, args.config, K, Wm, True)
    c5 = None
    if world > 1 and args.config == "c2" and not args.no_c5:
        k5 = 4 * max(1, min(K, 8) // 4)             # whole 16-spp frames (four 4-spp passes each)
        c5 = measure(B, "c5", k5, min(Wm, 4), False)

    if rank == 0:
        sha, version = kernels_sha(R)
        npix, spp, depth = m["W"] * m["H"], m["spp"], m["depth"]
        ms = m["ms_per_step"]
        sec = ms * 1e-3
        # bytes per pass: SURVEY.md 8(d)'s weights on (a) the visits the reference's un-pruned walk makes, (b) the tests the timed kernels execute.
        # The framebuffer term as EXECUTED: a group of passes reads and writes a pixel's accumulator once and stores its ARGB word once (the per-pass
        # accumulate / divide / gamma arithmetic stays in registers), so 36 B per pixel and GROUP, i.e. 36 / passes-per-group per pixel-pass
        group = min(max(1, m["group_passes"]), K) if args.pipeline == 4 else 1
        alg_ref = algorithmic_bytes(m["st_ref"], npix)
        alg_run = algorithmic_bytes(m["st_run_pass"], npix / float(group))
        tr, tr_src = profile_field("r03_traffic_%s.json" % args.config, sha) if world == 1 else (None, "N > 1: no per-rank profile")
        ins, ins_src = profile_field("r03_%s_insts.json" % args.config, sha) if world == 1 else (None, "N > 1: no per-rank profile")
        traffic = tr.get("hbm_bytes_per_pass") if tr else None
        valu_per_pass = ins.get("valu_instructions_per_pass") if ins else None
        hbm_gbs = (traffic / sec / 1e9) if traffic else None
        result = {
            "metric": "Mrays/s (rays = closest-hit scene queries, primary + secondary) at %dx%d depth %d" % (m["W"], m["H"], depth),
            "value": m["value"], "unit": "Mrays/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": m["data"],
            "config": {"workload": m["workload"],
                       "step": "one pass of the reference's sample loop over the whole frame; the K steps are ONE rtw_render_passes call, whose passes are rendered in groups "
                               "that share one set of launches: a pixel's pass colours are added to its accumulator one by one in pass order (in registers); its accumulator entry is written, and its "
                               "ARGB word (divide + gamma of the accumulator after the group's last pass) computed and written, ONCE PER GROUP -- intermediate images are not materialised inside a call (single_pass_ms is "
                               "the rate at which a caller can show every pass)",
                       "sharding": "10-row tasks round-robin over ranks, one gather of the rows to rank 0 after the K passes",
                       "seed": SEED, "prune": args.prune, "pipeline": args.pipeline, "pipeline_run": m["pipeline_run"], "passes_per_group": group},
            "library": version,
            "camera_Mrays_per_s": m["camera_Mrays_per_s"],
            "nontrivial_Mrays_per_s": m["nontrivial"] / (ms * K * 1e-3) / 1e6,
            "nontrivial_rays_note": "rays that meet a shape's culling box: camera rays by pixel-centre direction (%d of %d per pass) + all secondary rays" % (m["nontrivial_cam"], npix * spp),
            "rays_per_frame": m["rays_per_frame"],
            "verified_bit_identical_to_single_kernel_replay": m["verified"],
            "repeat_call_ms_per_step": m["repeat_ms_per_step"],
            "repeat_calls": 3 if m["repeat_ms_per_step"] else 0,
            "timed_call_ms": {"wall_between_barriers": ms * K, "hip_events_around_the_call": m["render_ms_per_step_rank0"] * K, "host_inside_the_call": m["enqueue_ms"],
                              "note": "the K steps are one asynchronous call: `host_inside_the_call` is how long the host took to enqueue its launches; a stall of that thread shows here and in both other figures"},
            "repeat_call_note": "the timed call again, three more times (best; host clock around call + synchronise): the timed figure above is the first call of its shape after a shorter warm-up",
            "single_pass_ms": m["single"]["median"] if m["single"] else None,
            "single_pass_note": ("median of 20 separate rtw_render_passes(n = 1) calls (min %.4f, max %.4f ms; HIP events around each call; launch-size hints primed by 4 such calls): "
                                 "a frame that can be shown after EVERY pass, as the reference's window does" % (m["single"]["min"], m["single"]["max"])) if m["single"] else None,
            "cold_first_call_ms": m["first_call_ms"],
            "cold_first_call_note": "wall clock of the first one-pass call on this frame shape (screen bins built on the device, tile / job tables, workspace hipMalloc); the second such call: %.3f ms" % m["second_call_ms"],
            "device_memory_bytes": m["memory_bytes"],
            "scene_commit_ms": m["commit_ms"],
            "bins_and_tables_build_ms": max(0.0, m["first_call_ms"] - m["second_call_ms"]),
            "tree_and_bins_build_ms": m["build_ms"],
            "host_setup_note": "scene_commit_ms = tree build (on the device by default, the reference's split decisions) + derived layouts + upload / read-back; bins_and_tables_build_ms = "
                               "first render call of this frame shape minus the second; tree_and_bins_build_ms times both constructions, device and host, on fresh scenes (two runs each, "
                               "alternating, the faster kept; commit_ms includes the texture atlas upload, the same either way); none of it is in the timed region",
            "roofline": {"bound": "latency",
                         "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (hbm_gbs / HBM_PEAK_GBS) if hbm_gbs else None,
                         "achieved_note": "HBM bytes per pass that rocprofv3's FETCH_SIZE / WRITE_SIZE counted (`traffic`) / ms_per_step: the rocprof-evidenced share of the 8 TB/s roof; "
                                          "null when no profile of THIS build's kernels is on file",
                         "traffic": traffic, "traffic_source": tr_src,
                         "frac_algorithmic_executed": alg_run / world / sec / 1e9 / HBM_PEAK_GBS,
                         "frac_algorithmic_reference_order": alg_ref / world / sec / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_note": "SURVEY.md 8(d)'s model -- 32 B/box test + 48 B/triangle test + 64 B/shaded hit + 16 B/texture sample + 36 B per pixel and GROUP of passes (the framebuffer "
                                             "traffic as executed) -- prices every record as if fetched from HBM; here the tree is staged in LDS and the records come out of L2 / Infinity "
                                             "Cache, so these are NOT HBM fractions: `executed` counts the tests the timed kernels run, `reference_order` the reference's own un-pruned visit "
                                             "list (bins and pruning provably skip visits the reference makes and rejects), and either can exceed 1",
                         "valu_issue_frac": (valu_per_pass * 2.0 / (256 * 4) / 2.4e9 / sec) if valu_per_pass else None,
                         "valu_issue_source": ins_src,
                         "valu_issue_note": "wave-level VALU instructions per pass (SQ_INSTS_VALU) x 2 cycles (a wave64 instruction on a SIMD-32) / (256 CUs x 4 SIMDs x 2.4 GHz) / ms_per_step",
                         "trace_kernels_valu_per_secondary_ray": ins.get("trace_kernels_valu_instructions_per_secondary_ray") if ins else None,
                         "limiter": "latency, not a throughput roof: HBM carries about a tenth of its peak (the tree, bins and triangle records stay in LDS / L2 / Infinity Cache) and the "
                                    "vector units issue well under their peak; a ray's walk is a chain of dependent steps (node record from LDS -> box test -> next index), a launch lasts "
                                    "as long as its slowest wave, and shading gathers scattered path records",
                         "kernel": "render pass = 1/K of a group: gprimary (+ gsky beside it) + per-bounce gtrace / gshade rounds + gresolve (rtw_group_kernels.h)" if args.pipeline == 4
                                   else "render pass (one rtw_render_tasks call)",
                         "kernel_ms": m["render_ms_per_step_rank0"],
                         "stage_ms": m["stage_ms"],
                         "algorithmic_bytes_per_pass_executed": alg_run,
                         "algorithmic_bytes_per_pass_reference_order": alg_ref,
                         "counters_per_pass_reference_order": m["st_ref"], "counters_per_pass_as_run": m["st_run_pass"]},
        }
        if B.rehearse:
            result["rehearsal"] = "RTW_BENCH_REHEARSE=1: all %d ranks on GPU 0, gloo process group, loopback transport under rtw_gather_rows -- a check of the code path, not a measurement" % world
        if world > 1:
            result["gather_ms"] = m["gather_ms"]
            result["gather"] = B.gather_kind + ("; ARGB only" if not args.gather_accum else "; accumulator + ARGB")
            result["gather_verified_bit_identical_to_1gpu"] = m["verified"]
            if c5 is not None:
                result["c5"] = {"workload": c5["workload"], "value": c5["value"], "unit": "Mrays/s", "steps": c5["steps"], "warmup": c5["warmup"], "ms_per_step": c5["ms_per_step"],
                                "render_ms_per_step_rank0": c5["render_ms_per_step_rank0"], "gather_ms": c5["gather_ms"], "rays_per_frame": c5["rays_per_frame"],
                                "camera_Mrays_per_s": c5["camera_Mrays_per_s"], "gather_verified_bit_identical_to_1gpu": c5["verified"], "scaling": "strong",
                                "note": "the same ranks, the same timing rules (barrier + synchronize on both sides, max over ranks), %d steps = %d frame(s) of 16 spp; C2's per-rank share is "
                                        "small against each launch's latency floor, this is the frame whose tiles can pay" % (c5["steps"], c5["steps"] // 4)}
        if world == 1 and not args.no_cpu:
            try:
                result["cpu_baseline"] = cpu_baseline(m["mesh_path"], m["W"], m["H"], spp, depth, m["kind"], m["rays_per_frame"], args.cpu_budget)
            except Exception as e:      # the baseline is a reported extra, never the thing measured
                result["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": None, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(result), flush=True)
    B.close()


if __name__ == "__main__":
    main()
