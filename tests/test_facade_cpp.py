"""The C++ facade (include/RayTracerWin.hpp): SetupScene-style code written with the reference's names compiles with
g++ against the C ABI and links to librtwin.so; without a GPU it fails loudly, on the GPU it renders the same image
as the Python mirror."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, asset


def build_example(tmp_path, name="render_mesh"):
    exe = str(tmp_path / name)
    lib_dir = os.path.join(ROOT, "raytracerwin_amd")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".cpp"), "-L" + lib_dir, "-lrtwin",
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


def test_progressive_example_compiles_and_refuses_to_run_without_a_gpu(tmp_path):
    import torch
    exe = build_example(tmp_path, "progressive")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, asset("TorusKnot.obj"), "32", "32", "2", "4"], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_update_bitmap_pixels_loop_names_its_png_and_accumulates_like_pass_by_pass_calls(tmp_path):
    """UpdateBitmapPixels of the facade (Src/RayTracerProgram.cpp:270-422): preview pass, N accumulated 4-sub-sample passes,
    the reference's progress line, Output_<spp>spp_<date>.png in the SavedImages folder that holds Output.txt."""
    import re
    import raytracerwin_amd as R
    exe = build_example(tmp_path, "progressive")
    (tmp_path / "SavedImages").mkdir()
    (tmp_path / "SavedImages" / "Output.txt").write_text("")
    run_dir = tmp_path / "Build"
    run_dir.mkdir()
    W, H, N = 160, 90, 6
    raw = str(tmp_path / "o.argb")
    out = subprocess.run([exe, asset("TorusKnot.obj"), str(W), str(H), str(N), "4", raw], capture_output=True, text=True, cwd=str(run_dir), check=True).stdout
    assert len(re.findall(r"RayTracer - S: \[\d+/%d\] \| T: \[.* / .*\] \| F: \[\d+ms\]" % N, out)) == N
    pngs = [f for f in os.listdir(str(tmp_path / "SavedImages")) if re.fullmatch(r"Output_%dspp_\d{4}-\d\d-\d\d_\d\d-\d\d-\d\d\.png" % N, f)]
    assert len(pngs) == 1
    cpp = np.fromfile(raw, np.uint32)
    ctx = R.Context(0)
    s = R.RayTracerScene(ctx)
    s.AddShape(R.RMeshShape.Create(asset("TorusKnot.obj")), R.SurfaceMaterial_Diffuse((1, 1, 1)))
    fb = R.Framebuffer(ctx, W, H)
    for p in range(N):
        R.ThreadWorker_Render(s, fb, 0, W * H - 1, 4, None, p, 4, 12345)
    assert (fb.resolve_argb() == cpp).all()
    from PIL import Image
    im = np.asarray(Image.open(os.path.join(str(tmp_path / "SavedImages"), pngs[0])))
    assert im.shape == (H, W, 3) and (im[..., 2].ravel() == cpp & 255).all()
    ctx.close()


def test_default_scene_example_compiles_and_refuses_to_run_without_a_gpu(tmp_path):
    import torch
    exe = build_example(tmp_path, "default_scene")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, asset("unitychan.obj"), "1", "2", "32", "32"], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_default_scene_example_equals_the_scene_built_through_the_python_mirror(tmp_path):
    """examples/default_scene.cpp states RayTracerProgram::SetupScene (Src/RayTracerProgram.cpp:467-552) with the facade's RSphere /
    RCapsule / RPlane / RMeshShape and renders it with UpdateBitmapPixels; the picture equals the same scene built from
    tests/scenes.py (the description the reference-pinned fixtures were rendered from), accumulated pass by pass."""
    import raytracerwin_amd as R
    import scenes as SC
    from oracle import oracle as O
    exe = build_example(tmp_path, "default_scene")
    W, H, N, D = 200, 200, 3, 6
    raw = str(tmp_path / "o.argb")
    subprocess.run([exe, asset("unitychan.obj"), str(N), str(D), str(W), str(H), raw], capture_output=True, text=True, cwd=str(tmp_path), check=True)
    cpp = np.fromfile(raw, np.uint32)
    ctx = R.Context(0)
    s = R.RayTracerScene(ctx)
    for sh in SC.default_scene():
        mat = R.material_nodes_from_array(O.materials(sh[-1]))
        shape = {"sphere": lambda: R.RSphere.Create(sh[1], sh[2]), "plane": lambda: R.RPlane.Create(sh[1], sh[2]),
                 "capsule": lambda: R.RCapsule.Create(sh[1], sh[2], sh[3]), "mesh": lambda: R.RMeshShape.Create(asset(sh[1] + ".obj"))}[sh[0]]()
        s.AddShape(shape, mat)
    fb = R.Framebuffer(ctx, W, H)
    for p in range(N):
        R.ThreadWorker_Render(s, fb, 0, W * H - 1, D, None, p, 4, 12345)
    assert (fb.resolve_argb() == cpp).all()
    ctx.close()
