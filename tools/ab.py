#!/usr/bin/env python3
"""tools/ab.py CONFIG K "name=value,name=value" ["..." ...]  -- ON THE GPU BOX: same-session A/B of context options: for each option set, 3 warm-up calls
and 12 timed rtw_render_passes calls of K passes (HIP-synchronised wall clock per call); prints median / min ms per pass.  A set named like
"@lib.so:opts" loads nothing new (one library per process): run the script once per library (RTW_LIB)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracerwin_amd as R  # noqa: E402
import bench  # noqa: E402

cfg, K = sys.argv[1], int(sys.argv[2])
sets = sys.argv[3:] or [""]
mesh, W, H, spp, depth, kind = bench.CONFIGS[cfg]
ctx = R.Context(0)
s = R.RayTracerScene(ctx)
if kind == "setup":
    from raytracerwin_amd.setup_scene import SetupScene
    SetupScene(s, os.path.join(ROOT, "assets", mesh + ".obj"))
else:
    s.AddShape(R.RMeshShape.Create(os.path.join(ROOT, "assets", mesh + ".obj")), bench.make_material(R, kind))
s.commit()
fb = R.Framebuffer(ctx, W, H)
DEFAULT = {}
rounds = 2
res = {o: [] for o in sets}
for rnd in range(rounds):          # the sets interleaved, twice: drift of the box's clock shows as a difference between the rounds
    for o in sets:
        kv = [x.split("=") for x in o.split(",") if x]
        for k, v in kv:
            ctx.set_option(k, int(v))
        s.render_reserve(fb, 10, 0, 1, depth, K, spp)
        p = 0
        for i in range(3):
            s.render_passes(fb, 10, 0, 1, depth, None, p, K, spp, 12345); p += K
        ctx.synchronize()
        for i in range(12):
            t0 = time.perf_counter()
            s.render_passes(fb, 10, 0, 1, depth, None, p, K, spp, 12345); p += K
            ctx.synchronize()
            res[o].append((time.perf_counter() - t0) * 1e3 / K)
        for k, v in kv:     # back to the library's defaults for the next set
            ctx.set_option(k, {"backface_filter": 1, "group_parts": 2, "group_split": 1, "split_paths": 400000, "split_min": 8, "visit_budget": 384, "wave_below": 80000,
                               "group_max": 256, "primary_passes": 0, "sky_blocks": 4, "group_paths": 32 << 20, "workspace_limit_mb": 24 << 10, "device_build": 1, "pipeline": 4}.get(k, 0))
for o in sets:
    a = np.array(res[o])
    print("%-4s K=%-3d %-60s median %.4f  min %.4f  (rounds %.4f / %.4f) ms per pass" % (cfg, K, o or "(defaults)", np.median(a), a.min(), np.median(a[:12]), np.median(a[12:])))
