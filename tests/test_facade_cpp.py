"""The C++ facade (include/RayTracerWin.hpp): SetupScene-style code written with the reference's names compiles with
g++ against the C ABI and links to librtwin.so; without a GPU it fails loudly, on the GPU it renders the same image
as the Python mirror."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, asset


def build_example(tmp_path):
    exe = str(tmp_path / "render_mesh")
    lib_dir = os.path.join(ROOT, "raytracerwin_amd")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "render_mesh.cpp"), "-L" + lib_dir, "-lrtwin",
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def test_facade_compiles_links_and_refuses_to_run_without_a_gpu(tmp_path):
    import torch
    exe = build_example(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, asset("TorusKnot.obj"), "32", "32", "1", "4", str(tmp_path / "o.png")], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_facade_renders_like_the_python_mirror(tmp_path):
    import raytracerwin_amd as R
    exe = build_example(tmp_path)
    W, H = 200, 120
    png, raw = str(tmp_path / "o.png"), str(tmp_path / "o.argb")
    subprocess.check_call([exe, asset("BlenderMonkey.obj"), str(W), str(H), "2", "6", png, raw])
    cpp = np.fromfile(raw, np.uint32)
    ctx = R.Context(0)
    s = R.RayTracerScene(ctx)
    s.AddShape(R.RMeshShape.Create(asset("BlenderMonkey.obj")),
               R.SurfaceMaterial_Blend(R.SurfaceMaterial_Reflective((1, 1, 1), 0.2), R.SurfaceMaterial_Diffuse((1, 1, 1)), 1.0))
    fb = R.Framebuffer(ctx, W, H)
    for p in range(2):
        R.ThreadWorker_Render(s, fb, 0, W * H - 1, 6, None, p, 4, 12345)
    assert (fb.resolve_argb() == cpp).all()
    from PIL import Image
    im = np.asarray(Image.open(png))
    assert im.shape == (H, W, 3) and (im[..., 0].ravel() == (cpp >> 16) & 255).all()
    ctx.close()
