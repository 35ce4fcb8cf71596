#!/usr/bin/env python3
"""Long accumulation runs (test infrastructure, needs a GPU): hundreds of passes of C2 (1 and 4 samples) and of SetupScene rendered
as one rtw_render_passes run, pass by pass through rtw_render_tasks, and dealt over 3 ranks -- the three accumulators and pictures must
be bit-identical (the sky-only tiles run on a second stream that a run forks and joins once; this is the check that nothing races).

  python tools/longrun.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracerwin_amd as R
from raytracerwin_amd.setup_scene import SetupScene
ctx=R.Context(0)
def scene(kind):
    s=R.RayTracerScene(ctx)
    if kind=='setup': SetupScene(s, ROOT+'/assets/unitychan.obj')
    else: s.AddShape(R.RMeshShape.Create(ROOT+'/assets/TorusKnot.obj'), R.SurfaceMaterial_Diffuse())
    s.commit(); return s
for kind,W,H,d,spp,N in (('c2',1920,1080,4,1,600),('setup',800,800,10,4,150),('c2',1920,1080,4,4,200)):
    s=scene(kind)
    res=[]
    for mode in ('run','single','run_world3'):
        fb=R.Framebuffer(ctx,W,H)
        if mode=='run': s.render_passes(fb,10,0,1,d,None,0,N,spp,777)
        elif mode=='single':
            for p in range(N): s.render_tasks(fb,10,0,1,d,None,p,spp,777)
        else:
            for r in range(3): s.render_passes(fb,10,r,3,d,None,0,N,spp,777)
        res.append((fb.read_float().view(np.uint32).copy(), fb.resolve_argb().copy()))
    ok=all((res[0][0]==r[0]).all() and (res[0][1]==r[1]).all() for r in res[1:])
    print(kind,W,H,'spp',spp,'passes',N,'run == pass-by-pass == 3 ranks:',ok,flush=True)
