"""Host-side logic of the product library, runnable without a GPU: the C-ABI library loads and
exports every symbol include/rtwin.h declares, refuses to run without a device (no CPU
fallback), and its parser / tree builder / tables agree with the reference goldens and the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, asset

import raytracerwin_amd as R
from raytracerwin_amd import api


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rtwin.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(rtw_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 30
    L = R.library()
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


def test_no_device_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(R.RtwError) as e:
        R.Context(0)
    assert e.value.code == -2
    s = R.RayTracerScene(None)
    s.AddShape(R.RMeshShape.Create(asset("TorusKnot.obj")), R.SurfaceMaterial_Diffuse())
    with pytest.raises(R.RtwError) as e:
        s.FindIntersectionWithScene(np.zeros((1, 7), np.float32))
    assert e.value.code == -2


def test_product_path_does_not_touch_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raytracerwin_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt and \
                    "rt_oracle" not in txt, f


@pytest.mark.parametrize("name", ["TorusKnot", "BlenderMonkey", "unitychan"])
def test_parser_and_flattened_tree_match_reference(name):
    g = np.load(os.path.join(GOLDEN, "mesh_%s.npz" % name))
    s = R.RayTracerScene(None)
    s.AddShape(R.RMeshShape.Create(asset(name + ".obj")))
    info = s.mesh_info()
    assert [info["points"], info["texcoords"], info["normals"], info["tris"]] == g["counts"][:4].tolist()
    assert (bits(info["bounds"]) == bits(g["shape_bounds"])).all()
    b, skip, tri = s.mesh_nodes()
    assert len(b) == 2 * info["tris"] - 1
    assert (bits(b) == bits(g["tree_bounds"])).all()
    assert (tri == g["tree_tri"]).all()
    # skip links: subtree sizes of a full binary tree in preorder
    n = len(skip)
    assert skip[0] == n
    leaf = tri >= 0
    assert (skip[leaf] == np.arange(n)[leaf] + 1).all()
    internal = np.where(~leaf)[0]
    assert (skip[skip[internal + 1]] == skip[internal]).all()       # right child ends where the parent ends
    ntex = int((g["texinfo"][:, 0] > 0).sum())
    assert info["textures"] == ntex


def test_tables_match_oracle(oracle_mod):
    O = oracle_mod
    rng = np.random.default_rng(0)
    for _ in range(2000):
        a = [int(v) for v in rng.integers(0, 2 ** 32, 4, dtype=np.uint64)]
        assert api.rand31(*a) == O.rand31(*a)
    for i in list(range(64)) + [0xFFFFFE, 12345678, 7777777]:
        assert (bits(api.unit_table_entry(i)) == bits(O.unit_table_entry(i))).all()
    assert (bits(api.gamma_thresholds()) == bits(O.gamma_thresholds())).all()
    assert (bits(api.texel_lut()) == bits(O.texel_lut())).all()


def test_gamma_threshold_staircase_equals_powf_formula(oracle_mod):
    """MakePixelColor(LinearToGamma(c)) (libm powf, via the oracle) == number of thresholds <= c,
    which is what the device evaluates instead of calling a device powf."""
    thr = api.gamma_thresholds().astype(np.float32)
    assert (np.diff(thr) > 0).all()
    rng = np.random.default_rng(1)
    c = np.concatenate([rng.random(300000, dtype=np.float32), thr, np.nextafter(thr, np.float32(0)),
                        np.nextafter(thr, np.float32(2)), np.float32([0, 1, 1.5, -0.5, 1e-30, np.inf, 3e38, -0.0])])
    c = np.resize(c, (len(c) // 3) * 3)
    direct = oracle_mod.make_pixel_colors(c.reshape(-1, 3))
    stair = np.searchsorted(thr, c, side="right") - 1
    stair = np.where(c > 0, np.clip(stair, 0, 255), 0).astype(np.uint32).reshape(-1, 3)
    packed = (np.uint32(255) << 24) | (stair[:, 0] << 16) | (stair[:, 1] << 8) | stair[:, 2]
    assert (direct == packed).all()


def test_png_reader_matches_pillow_and_writer_round_trips(tmp_path):
    from PIL import Image
    for rel in ("unitychan/cheek_00.png", "unitychan/skin_01.png"):
        px = api.png_load(asset(rel))
        ref = np.asarray(Image.open(asset(rel)))
        assert px.shape == ref.shape and (px == ref).all()
    rng = np.random.default_rng(3)
    argb = rng.integers(0, 2 ** 32, 37 * 21, dtype=np.uint64).astype(np.uint32)
    out = str(tmp_path / "o.png")
    api._check(R.library().rtw_png_save_argb(out.encode(), argb.ctypes.data_as(C.c_void_p), 37, 21))
    im = np.asarray(Image.open(out))
    assert im.shape == (21, 37, 3)
    assert (im[..., 0].ravel() == (argb >> 16) & 255).all() and (im[..., 2].ravel() == argb & 255).all()
    with pytest.raises(R.RtwError):
        api.png_load(asset("TorusKnot.obj"))


def test_bad_arguments_are_rejected():
    s = R.RayTracerScene(None)
    with pytest.raises(R.RtwError):
        s.AddShape(R.RMeshShape.Create("/nonexistent/mesh.obj"))
    pts = np.zeros((3, 3), np.float32)
    with pytest.raises(R.RtwError):     # index out of range: the reference would read out of bounds
        s.AddShape(R.RMeshShape.FromArrays(pts, pts, pts, [[0, 1, 5]], [[0, 1, 2]], [[0, 1, 2]]))
    idx = s.AddShape(R.RMeshShape.FromArrays(pts, pts, pts, [[0, 1, 2]], [[0, 1, 2]], [[0, 1, 2]]))
    bad = np.zeros(1, dtype=api.MATERIAL_DTYPE)
    bad[0] = (api.MAT_BLEND, 0, 0, 0, 0.5, 0, 0, 0)     # children must follow the parent
    with pytest.raises(R.RtwError):
        api._check(R.library().rtw_scene_set_material(s.h, idx, bad.ctypes.data_as(C.c_void_p), 1))


def test_empty_mesh_and_ragged_inputs():
    s = R.RayTracerScene(None)
    e = np.zeros((0, 3), np.float32)
    s.AddShape(R.RMeshShape.FromArrays(e, e, e, np.zeros((0, 3), np.int32), np.zeros((0, 3), np.int32), np.zeros((0, 3), np.int32)))
    s.commit()
    assert s.mesh_info()["nodes"] == 0


@pytest.mark.parametrize("name", ["TorusKnot", "BlenderMonkey", "unitychan"])
def test_quad_tree_is_an_order_preserving_collapse_of_the_reference_tree(name):
    """Depth-first over the 4-wide tree, slots left to right, must meet the leaves in the binary tree's
    preorder, each with the reference's own leaf box."""
    g = np.load(os.path.join(GOLDEN, "mesh_%s.npz" % name))
    s = R.RayTracerScene(None)
    s.AddShape(R.RMeshShape.Create(asset(name + ".obj")))
    b, c = s.mesh_quads()
    EMPTY = -2 ** 31
    order, boxes = [], []

    def walk(q):
        for k in range(4):
            ch = int(c[q, k])
            if ch == EMPTY:
                continue
            if ch < 0:
                order.append(-1 - ch)
                boxes.append(b[q, :, k])
            else:
                assert ch > q           # breadth-first numbering: children after parents
                walk(ch)
    import sys
    sys.setrecursionlimit(10000)
    walk(0)
    leaf = g["tree_tri"] >= 0
    assert order == g["tree_tri"][leaf].tolist()
    assert (bits(np.array(boxes)) == bits(g["tree_bounds"][leaf])).all()
