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


def test_format_time_string_equals_the_references_own(tmp_path):
    """tests/golden/timestrings.json holds what the reference's OWN FormatTimeString printed (harness command `timestring`)."""
    import json
    from conftest import GOLDEN
    table = json.load(open(os.path.join(GOLDEN, "timestrings.json")))
    exe = build_example(tmp_path, "format_time")
    out = subprocess.check_output([exe] + list(table.keys())).decode().strip().splitlines()
    got = {ln.split(" ", 1)[0]: ln.split(" ", 1)[1] for ln in out}
    assert got == table


def test_png_writer_decodes_to_what_the_references_writer_decodes_to(tmp_path):
    """png_reference_decode.npz: an ARGB buffer, and the RGB pixels decoded from the PNG the reference's own RTexture::SaveBufferToPNG
    (Src/Texture.cpp:201-283) wrote for it; rtw_png_save_argb's file must decode to the same pixels, all three channels, and both files
    are 8-bit RGB without alpha."""
    import raytracerwin_amd as R
    from conftest import GOLDEN
    from PIL import Image
    g = np.load(os.path.join(GOLDEN, "png_reference_decode.npz"))
    W, H = [int(v) for v in g["size"]]
    argb = np.ascontiguousarray(g["argb"], np.uint32)
    out = str(tmp_path / "mine.png")
    import ctypes as C
    assert R.library().rtw_png_save_argb(out.encode(), argb.ctypes.data_as(C.c_void_p), W, H) == 0
    mine, ref = Image.open(out), Image.open(os.path.join(GOLDEN, "png_written_by_reference.png"))
    assert mine.mode == ref.mode == "RGB" and mine.size == ref.size == (W, H)
    assert (np.asarray(mine) == g["rgb"]).all()
    assert (np.asarray(ref) == g["rgb"]).all()
    # and the library's reader reads the reference's file back to the same pixels
    px = R.png_load(os.path.join(GOLDEN, "png_written_by_reference.png"))
    assert px.shape == (H, W, 3) and (px == g["rgb"]).all()


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
    from oracle import oracle as O
    W, H, N = 160, 90, 6
    # the ORACLE's image of the same run (CPU restatement of the reference, pinned by the reference's own outputs)
    os_ = O.Scene()
    sh = os_.add_mesh_obj(asset("TorusKnot.obj"))
    os_.set_material(sh, [(O.MAT_DIFFUSE, (1, 1, 1), 0, 0, 0)])
    ofb = O.Framebuffer(W, H)
    for p in range(N):
        os_.render_pass_pool(ofb, 4, False, p, 4, 12345, threads=0, task_rows=10)
    want = ofb.read()[1]
    time_re = r"(\d+ms|\d+s|\d+m:\d+s|\d+h:\d+m:\d+s)"
    for per_update in (1, 4):       # the reference's rhythm (a line per pass), and passes batched four to an update
        for f in os.listdir(str(tmp_path / "SavedImages")):
            if f.endswith(".png"):
                os.remove(os.path.join(str(tmp_path / "SavedImages"), f))
        raw = str(tmp_path / ("o%d.argb" % per_update))
        out = subprocess.run([exe, asset("TorusKnot.obj"), str(W), str(H), str(N), "4", raw, str(per_update)], capture_output=True, text=True, cwd=str(run_dir), check=True).stdout
        lines = re.findall(r"^RayTracer - S: \[(\d+)/%d\] \| T: \[%s / %s\] \| F: \[\d+ms\]$" % (N, time_re, time_re), out, re.M)
        assert [int(ln[0]) for ln in lines] == list(range(per_update, N, per_update)) + [N]
        shown = re.search(r"window: (\d+) frames presented, (\d+) titles, last frame equals the final image; title: RayTracer - S: \[%d/%d\]" % (N, N), out)
        assert shown and int(shown.group(1)) == len(lines) + 1 and int(shown.group(2)) == len(lines)      # + the preview
        pngs = [f for f in os.listdir(str(tmp_path / "SavedImages")) if re.fullmatch(r"Output_%dspp_\d{4}-\d\d-\d\d_\d\d-\d\d-\d\d\.png" % N, f)]
        assert len(pngs) == 1
        cpp = np.fromfile(raw, np.uint32)
        assert (cpp == want).all()
        from PIL import Image
        im = np.asarray(Image.open(os.path.join(str(tmp_path / "SavedImages"), pngs[0])))
        assert im.shape == (H, W, 3)
        assert (im[..., 0].ravel() == (cpp >> 16) & 255).all() and (im[..., 1].ravel() == (cpp >> 8) & 255).all() and (im[..., 2].ravel() == cpp & 255).all()


@pytest.mark.gpu
def test_update_bitmap_pixels_on_three_ranks_equals_the_one_rank_image(tmp_path):
    """The facade's UpdateBitmapPixels with Rank / World / Comm: three processes of examples/progressive.cpp, each rendering its
    10-row tasks, rank 0 gathering through rtw_gather_rows before every present and before it saves the PNG.  On this one-GPU box the three ranks share
    the GPU, so the transport under the gather is the loopback stand-in for librccl (tests/support/loopback_rccl.cpp, loaded through
    RTW_RCCL_LIBRARY); the id still travels from rank 0 to the others as it would with RCCL (a file).  Rank 0's final image, the frames its window
    was shown and its PNG equal the oracle's image of the whole frame."""
    import re
    from oracle import oracle as O
    exe = build_example(tmp_path, "progressive")
    lib = str(tmp_path / "libloopback_rccl.so")
    subprocess.check_call(["g++", "-O1", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "support", "loopback_rccl.cpp"),
                           "-o", lib, "-L/opt/rocm/lib", "-lamdhip64"])
    (tmp_path / "SavedImages").mkdir()
    (tmp_path / "SavedImages" / "Output.txt").write_text("")
    run_dir = tmp_path / "Build"
    run_dir.mkdir()
    W, H, N, world = 200, 117, 5, 3             # 12 tasks, the last one ragged
    os_ = O.Scene()
    sh = os_.add_mesh_obj(asset("TorusKnot.obj"))
    os_.set_material(sh, [(O.MAT_DIFFUSE, (1, 1, 1), 0, 0, 0)])
    ofb = O.Framebuffer(W, H)
    for p in range(N):
        os_.render_pass_pool(ofb, 4, False, p, 4, 12345, threads=0, task_rows=10)
    want = ofb.read()[1]
    env = dict(os.environ, RTW_RCCL_LIBRARY=lib, RTW_LOOPBACK_DIR=str(tmp_path))
    raw = str(tmp_path / "rank0.argb")
    procs = [subprocess.Popen([exe, asset("TorusKnot.obj"), str(W), str(H), str(N), "4", raw if r == 0 else str(tmp_path / ("rank%d.argb" % r)), "2",
                               str(r), str(world), str(tmp_path / "comm.id")], cwd=str(run_dir), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240))
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append(("", "timeout"))
    assert [p.returncode for p in procs] == [0] * world, outs
    got = np.fromfile(raw, np.uint32)
    assert (got == want).all()
    assert re.search(r"last frame equals the final image", outs[0][0]), outs[0][0]
    pngs = [f for f in os.listdir(str(tmp_path / "SavedImages")) if f.endswith(".png")]
    assert len(pngs) == 1                           # rank 0 alone saves
    from PIL import Image
    im = np.asarray(Image.open(os.path.join(str(tmp_path / "SavedImages"), pngs[0])))
    assert (im[..., 0].ravel() == (got >> 16) & 255).all() and (im[..., 2].ravel() == got & 255).all()


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


@pytest.mark.gpu
def test_progressive_loop_at_1080p_presents_the_device_image_after_every_pass(tmp_path):
    """The reference shows every pass (Src/RayTracerProgram.cpp:184-185,346-360; the window blits bitcolor[]).  The facade's loop with ONE pass per update and a
    window that takes the DEVICE image (RenderWindow::SetDeviceSink: no 8.3 MB host copy per present): 1920 x 1080, the preview + 40 passes of four sub-samples,
    a present after each.  Checked: every update is presented, the final image equals the pass-by-pass image of the Python mirror (bit for bit), and the loop's
    time per update, end to end (launches, synchronise, title, present), stays within 1.6 x the device time of a one-pass call + 0.15 ms of host work."""
    import re
    import time
    import raytracerwin_amd as R
    exe = build_example(tmp_path, "progressive")
    W, H, N = 1920, 1080, 40
    raw = str(tmp_path / "p.argb")
    env = dict(os.environ, RTW_EXAMPLE_DEVICE_SINK="1", RTW_EXAMPLE_QUIET="1")
    out = subprocess.run([exe, asset("TorusKnot.obj"), str(W), str(H), str(N), "4", raw, "1"], capture_output=True, text=True, cwd=str(tmp_path), check=True, env=env).stdout
    shown = re.search(r"window: (\d+) frames presented, (\d+) titles, last frame stayed on the device", out)
    assert shown and int(shown.group(1)) == N + 1 and int(shown.group(2)) == N, out[-600:]
    per_update = float(re.search(r"([\d.]+) ms per update end to end", out).group(1))
    ctx = R.Context(0)
    s = R.RayTracerScene(ctx)
    s.AddShape(R.RMeshShape.Create(asset("TorusKnot.obj")), R.SurfaceMaterial_Diffuse((1, 1, 1)))
    fb = R.Framebuffer(ctx, W, H)
    for p in range(N):
        s.render_passes(fb, 10, 0, 1, 4, None, p, 1, 4, 12345)
    want = fb.resolve_argb()
    assert (np.fromfile(raw, np.uint32) == want).all()
    for p in range(8):      # device time of a one-pass call, hints primed
        s.render_passes(fb, 10, 0, 1, 4, None, N + p, 1, 4, 12345)
    ctx.synchronize()
    t0 = time.perf_counter()
    for p in range(20):
        s.render_passes(fb, 10, 0, 1, 4, None, N + 8 + p, 1, 4, 12345)
    ctx.synchronize()
    one_pass = (time.perf_counter() - t0) * 1e3 / 20
    ctx.close()
    assert per_update <= 1.6 * one_pass + 0.15, (per_update, one_pass)
