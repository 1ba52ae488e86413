"""OBJ + MTL ingest (host/solstrale_obj.cpp = src/loader/obj.rs:38-136 restated) at the sizes BASELINE.json names: configs 3 - 5 are OBJ
files of 262 k and 1.1 M triangles, the reference's own largest asset is a 1 368-face spider. tests/tools/export_obj.py writes the stand-in
meshes as OBJ + MTL + texture files; here they go through the loader - the material table with `map_Kd` and a `map_bump` height map, the host
reference-BVH build that defines the dfs order - and, under -m gpu, through sol_scene_create and a 128x128 crop against the oracle.
Measured (MI355X box, 16 host cores): profiles/r05_obj_ingest.txt."""
import os
import sys
import time

import numpy as np
import pytest

import orc
import parity_util as pu
from solstrale_amd import CameraConfig, DeviceScene, RenderConfig, SceneBuilder, _abi, scenes

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
import export_obj  # noqa: E402

ATRIUM_CAMERA = ((-13.0, 2.2, 0.6), (6.0, 4.5, -0.4), 55.)
ATRIUM_LIGHT = ((-6., scenes.ATRIUM_HEIGHT + 1.5, -2.0), (12., 0, 0), (0, 0, 4.0), (18., 17., 15.))


def test_exported_atrium_comes_back_through_the_loader(tmp_path):
    """A reduced atrium (the regular mesh at 20 000 triangles, 64^2 textures): every triangle back; 24 MTL materials + the
    default + the light's; 8 image albedos, one of them with the normal map the loader made out of the `map_bump` HEIGHT map."""
    n = 20000
    path = export_obj.export_atrium(str(tmp_path), n, 64, mesh="regular")
    sc = scenes.obj_file_scene(path, RenderConfig(64, 36, 1), camera=ATRIUM_CAMERA, light=ATRIUM_LIGHT)
    d = sc.desc
    assert d.n_triangles == n and d.n_quads == 1
    tri, _, _ = scenes.atrium_mesh(n, 24, "regular")
    want = tri.astype(np.float32).astype(np.float64).reshape(n, 9)  # (tobj parses `v` as f32, obj.rs:139-145 widens)
    got = np.array([np.concatenate([t.v0[:], np.array(t.v0[:]) + np.array(t.v0v1[:]), np.array(t.v0[:]) + np.array(t.v0v2[:])]) for t in (d.triangles[i] for i in range(n))])
    # the same SET of triangles (the flattened scene lists them in the order of the host's reference tree, bvh.rs:84-162)
    key = lambda a: a[np.lexsort(np.round(a, 6).T[::-1])]
    assert np.allclose(key(got), key(want), rtol=0, atol=1e-6)
    mats = [d.materials[i] for i in range(d.n_materials)]
    assert all(m.kind in (_abi.MAT_LAMBERTIAN, _abi.MAT_DIFFUSE_LIGHT) for m in mats)  # obj.rs:57-76: every OBJ material is Lambertian
    lamb = [m for m in mats if m.kind == _abi.MAT_LAMBERTIAN]
    image_albedo = [m for m in lamb if d.textures[m.albedo_tex].kind == _abi.TEX_IMAGE]
    assert len(image_albedo) == 8 and sum(1 for m in lamb if m.normal_tex >= 0) == 1
    bumped = [m for m in lamb if m.normal_tex >= 0][0]
    assert d.textures[bumped.normal_tex].kind == _abi.TEX_IMAGE and d.textures[bumped.normal_tex].width == 256


def test_material_ids_wrap_like_the_reference(tmp_path):
    """obj.rs:75,115: material keys are `i as i8` on both sides of the lookup. RESTATED, not widened: with more than 128 materials the keys wrap,
    material 256 + k REPLACES material k in the map (HashMap::insert), and material 255 replaces the default material (key -1). A file with 300
    one-triangle materials: triangle k (k < 300) carries the colour of the LAST material whose index is congruent to k modulo 256."""
    n = 300
    with open(tmp_path / "many.mtl", "w") as f:
        for k in range(n):
            f.write(f"newmtl m{k}\nKd {k / 1000.0:.3f} 0.5 0.25\n")
    with open(tmp_path / "many.obj", "w") as f:
        f.write("mtllib many.mtl\n")
        for k in range(n):
            f.write(f"v {k} 0 0\nv {k + 0.5} 0 0\nv {k} 0.5 0\n")
        for k in range(n):
            f.write(f"usemtl m{k}\nf {3 * k + 1} {3 * k + 2} {3 * k + 3}\n")
    b = SceneBuilder()
    model = b.load_obj(str(tmp_path) + os.sep, "many.obj", None, b.Lambertian(b.SolidColor(.9, .9, .9)))
    light = b.Sphere((0., 50., 0.), 1., b.DiffuseLight(1., 1., 1.))
    sc = b.finish(b.Bvh([light, model]), CameraConfig(30., 0., (0., 0., 5.), (0., 0., 0.), (0., 1., 0.)), (0., 0., 0.), RenderConfig(8, 8, 1))
    d = sc.desc
    assert d.n_triangles == n
    for i in range(n):
        t = d.triangles[i]
        k = int(round(t.v0[0]))
        last = max(j for j in range(n) if j % 256 == k % 256)
        red = d.textures[d.materials[t.material].albedo_tex].rgb[0]
        assert abs(red - np.float32(last / 1000.0)) < 1e-6, (k, last, red)


def test_atrium_file_at_baseline_size_loads(tmp_path):
    """The 262 267-triangle file through parser, material table and the host's reference-BVH build (the part of --obj that needs no GPU; also
    run under AddressSanitizer + UBSan by tests/tools/sanitize.sh): every triangle arrives, the tree is as deep as a mid-point split makes it."""
    path = export_obj.export_atrium(str(tmp_path), scenes.SPONZA_TRIANGLES, 64)
    t0 = time.time()
    sc = scenes.obj_file_scene(path, RenderConfig(64, 36, 1), camera=ATRIUM_CAMERA, light=ATRIUM_LIGHT)
    print(f"atrium.obj {os.path.getsize(path) / 1e6:.1f} MB: parse + host BVH + flatten {time.time() - t0:.2f} s, reference tree depth {sc.tree_depth}")
    assert sc.desc.n_triangles == scenes.SPONZA_TRIANGLES and sc.desc.n_nodes >= scenes.SPONZA_TRIANGLES - 1 and 18 <= sc.tree_depth < 200


@pytest.mark.gpu
def test_atrium_obj_at_baseline_size_renders_like_the_oracle(tmp_path):
    """configs[2] as a FILE: the heterogeneous atrium, 262 267 triangles, 24 materials, 8 image textures + a bump map, written as OBJ + MTL,
    parsed by the host loader, reference BVH built on the host, GPU tree built on the device, a 128x128 crop of the 1080p frame against the
    oracle (which walks the reference tree of the same loaded scene)."""
    t0 = time.time()
    path = export_obj.export_atrium(str(tmp_path), scenes.SPONZA_TRIANGLES, 256)
    t_export = time.time() - t0
    t0 = time.time()
    sc = scenes.obj_file_scene(path, RenderConfig(1920, 1080, 8), camera=ATRIUM_CAMERA, light=ATRIUM_LIGHT)
    t_load = time.time() - t0
    assert sc.desc.n_triangles == scenes.SPONZA_TRIANGLES
    t0 = time.time()
    with DeviceScene(sc) as ds:
        t_create = time.time() - t0
        ds.render(0, 8, pu.SEED)
        img = ds.read()
    rect = (900, 500, 1028, 628)
    ref, _ = orc.render(sc, 0, 8, pu.SEED, real=orc.ORC_F32, rect=rect)
    res = pu.compare(img, ref, 8, rect)
    print(f"atrium.obj {os.path.getsize(path) / 1e6:.1f} MB: export {t_export:.1f} s, parse + host BVH + flatten {t_load:.2f} s, sol_scene_create {t_create:.2f} s; {res}")
    assert np.isfinite(img).all() and res["bad_pixels"] == 0 and res["max_rel"] <= pu.REL_TOL, res
