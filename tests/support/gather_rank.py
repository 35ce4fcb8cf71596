"""One rank of test_gather_plan_between_ranks_sharing_one_gpu: renders its tasks of a small frame, takes part in rtw_gather_rows (over the
loopback transport of loopback_rccl.cpp, named by RTW_RCCL_LIBRARY) and, on rank 0, writes what the gathered framebuffer holds.
usage: gather_rank.py RANK WORLD MODE OUT.npz [W H]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import raytracerwin_amd as R  # noqa: E402
from conftest import asset  # noqa: E402

rank, world, mode, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
W, H, ROWS, SPP, DEPTH, PASSES, SEED = 200, 117, 10, 2, 3, 3, 9      # 12 tasks, the last one of 7 rows
if len(sys.argv) > 6:
    W, H = int(sys.argv[5]), int(sys.argv[6])
ctx = R.Context(0)
s = R.RayTracerScene(ctx)
s.AddShape(R.RMeshShape.Create(asset("TorusKnot.obj")), R.SurfaceMaterial_Reflective())
s.commit()
fb = R.Framebuffer(ctx, W, H)
s.render_passes(fb, ROWS, rank, world, DEPTH, None, 0, PASSES, SPP, SEED)
ident = R.Comm.unique_id() if world > 1 else None      # the loopback id is the same constant on every rank
comm = R.Comm(ctx, rank, world, ident)
comm.gather_rows(fb, ROWS, argb_only=(mode == "argb"))
ctx.synchronize()
msgs = comm.messages()
comm.gather_rows(fb, ROWS, argb_only=(mode == "argb"))      # a second gather (the staging blocks exist now) must leave the same frame
ctx.synchronize()
assert comm.messages() == 2 * msgs
np.save(out + ".messages%d.npy" % rank, np.array([msgs]))
if rank == 0:
    np.savez(out, accum=fb.read_float(), argb=fb.resolve_argb())
comm.close()
