#!/usr/bin/env python3
"""tools/wave_stats.py CONFIG [K] [ROUND] -- ON THE GPU BOX, with the timing build of the library (RTW_LIB=raytracerwin_amd/librtwin_t<ROUND>.so, built by
`make -C raytracerwin_amd/csrc EXTRA=-DRTW_TIMING=<ROUND> OBJ=$PWD/raytracerwin_amd/csrc/_obj_t OUT=$PWD/raytracerwin_amd/librtwin_t<ROUND>.so`):
per-wave clocks and lane-occupancy counters of the persistent trace kernel of trace round ROUND, for K-pass calls of a bench config.
What it answers: of a launch's span, how much do its waves spend walking / triangle-testing / refilling, and how full are the lanes while they do."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracerwin_amd as R  # noqa: E402
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mesh, W, H, spp, depth, kind = bench.CONFIGS[cfg]
ctx = R.Context(0)
ctx.set_option("group_split", 0)         # one launch per round: the clocks of two overlapping halves would mix
s = R.RayTracerScene(ctx)
s.AddShape(R.RMeshShape.Create(os.path.join(ROOT, "assets", mesh + ".obj")), bench.make_material(R, kind))
s.commit()
fb = R.Framebuffer(ctx, W, H)
NW = 16384


def stats(tag):
    buf = np.zeros(NW * 12, np.uint64)
    R.library().rtw_debug_read_timing(buf.ctypes.data_as(C.c_void_p), C.c_int(NW * 12))
    t = buf.reshape(NW, 12)
    t = t[t[:, 1] > 0].astype(np.float64)
    if len(t) == 0:
        print(tag, "no waves filed"); return
    dur = (t[:, 1] - t[:, 0]) / 100.0          # us (100 MHz wall clock)
    span = (t[:, 1].max() - t[:, 0].min()) / 100.0
    cyc = t[:, 11]
    mhz = np.median(cyc / np.maximum(dur, 1e-3))
    wt, wl, tt, tl, ev, cw, ctri, cr, rays = (t[:, k] for k in (2, 3, 4, 5, 6, 7, 8, 9, 10))
    print("%s: %d waves, span %.0f us, wave mean %.0f max %.0f us (clock ~%.0f MHz); rays/wave %.0f" % (tag, len(t), span, dur.mean(), dur.max(), mhz, rays.mean()))
    print("   time in walk %.0f%%  triangles %.0f%%  refill %.0f%%  (of the waves' own cycles); launch span used by the mean wave %.0f%%" %
          (100 * cw.sum() / cyc.sum(), 100 * ctri.sum() / cyc.sum(), 100 * cr.sum() / cyc.sum(), 100 * dur.mean() / span))
    print("   walk: %.0f visit slots/wave, lanes busy %.1f%% -> %.1f visits/ray, %.0f cycles per visit slot; triangles: %.0f iterations/wave, lanes busy %.1f%% -> %.2f tests/ray, %.0f cycles per iteration; %.0f events/wave, %.0f cycles per refill" %
          (wt.mean(), 100 * wl.sum() / (64 * wt.sum()), wl.sum() / max(rays.sum(), 1), cw.sum() / max(wt.sum(), 1), tt.mean(), 100 * tl.sum() / (64 * max(tt.sum(), 1)), tl.sum() / max(rays.sum(), 1),
           ctri.sum() / max(tt.sum(), 1), ev.mean(), cr.sum() / max(ev.sum(), 1)))


def call(first, k, tag):
    ctx.synchronize()
    t0 = time.perf_counter()
    s.render_passes(fb, 10, 0, 1, depth, None, first, k, spp, 12345)
    ctx.synchronize()
    stats("%s (%.3f ms)" % (tag, (time.perf_counter() - t0) * 1e3))


call(0, 1, "1 pass (cold)")
call(1, K, "%d passes (hints from 1)" % K)
p = 1 + K
for i in range(3):
    call(p, K, "%d passes #%d" % (K, i + 1))
    p += K
