"""Host logic above the C ABI (C++ mirror of the reference's host): constructors, transformations, the BVH builder
(src/hittable/bvh.rs:61-162), the flattener, Camera::new, get_lights order, error behaviour."""
import ctypes as C
import math

import numpy as np
import pytest

import orc
import parity_util as pu
from solstrale_amd import (CameraConfig, HostError, RenderConfig, RotationX, RotationY, RotationZ, Scale, SceneBuilder,
                           Translation, _abi, scenes)

CAM = CameraConfig(40., 0., (0, 0, -10), (0, 0, 0), (0, 1, 0))


def _tri_v0(transformation, v=(1., 0., 0.)):
    b = SceneBuilder()
    m = b.Lambertian(b.SolidColor(1, 1, 1))
    t = b.Triangle(v, (9, 9, 9), (7, 8, 9), m, transformation)
    l = b.Sphere((0, 50, 0), 1., b.DiffuseLight(1, 1, 1))
    sc = b.finish(b.Bvh([t, l]), CAM, (0, 0, 0), RenderConfig(8, 8, 1))
    return tuple(sc.desc.triangles[0].v0)


def test_transformation_doc_tests():
    # src/geo/transformation.rs:18-19,34-38,60-64,92-93,126-127,160-161,194-195
    assert _tri_v0(None, (1., 2., 3.)) == (1., 2., 3.)
    assert np.allclose(_tri_v0([RotationY(90.), Translation((1., 0., 0.))], (1., 0., 0.)), (1., 0., -1.), atol=1e-15)
    assert _tri_v0(Translation((4., 5., 6.)), (1., 2., 3.)) == (5., 7., 9.)
    assert np.allclose(_tri_v0(RotationX(90.), (2., 1., 0.)), (2., 0., -1.), atol=1e-8)
    assert np.allclose(_tri_v0(RotationY(90.), (2., 1., 0.)), (0., 1., -2.), atol=1e-8)
    assert np.allclose(_tri_v0(RotationZ(90.), (1., 0., 2.)), (0., -1., 2.), atol=1e-8)
    assert _tri_v0(Scale(3.), (2., 1., 0.)) == (6., 3., 0.)


def test_quad_fields_and_padded_box():
    """Quad::new (src/hittable/quad.rs:34-66): normal, d, w, area; flat axis padded by PAD_DELTA (src/geo/mod.rs:11,137-157)."""
    b = SceneBuilder()
    q = b.Quad((0, 0, 5), (2, 0, 0), (0, 3, 0), b.DiffuseLight(1, 1, 1))
    sc = b.finish(b.Bvh([q]), CAM, (0, 0, 0), RenderConfig(8, 8, 1))
    Q = sc.desc.quads[0]
    assert tuple(Q.normal) == (0., 0., 1.) and Q.d == 5. and Q.area == 6.
    assert np.allclose(tuple(Q.w), (0, 0, 1 / 6.))
    assert np.allclose(tuple(Q.bbox.v), (0, 2, 0, 3, 5 - 5e-5, 5 + 5e-5), atol=1e-12)
    # translation applies to q only, not to the edge vectors (transform(.., skip_translation=true), quad.rs:41-43)
    b = SceneBuilder()
    q = b.Quad((0, 0, 0), (1, 0, 0), (0, 1, 0), b.DiffuseLight(1, 1, 1), Translation((10, 20, 30)))
    sc = b.finish(b.Bvh([q]), CAM, (0, 0, 0), RenderConfig(8, 8, 1))
    assert tuple(sc.desc.quads[0].q) == (10., 20., 30.) and tuple(sc.desc.quads[0].u) == (1., 0., 0.)


def test_new_box_makes_six_quads():
    b = SceneBuilder()
    ids = b.new_box((1, 2, 3), (0, 0, 0), b.DiffuseLight(1, 1, 1))  # min/max are sorted per axis (quad.rs:77-78)
    sc = b.finish(b.Bvh(ids), CAM, (0, 0, 0), RenderConfig(8, 8, 1))
    assert sc.desc.n_quads == 6
    areas = sorted(sc.desc.quads[i].area for i in range(6))
    assert areas == [2., 2., 3., 3., 6., 6.]


def test_triangle_fields():
    b = SceneBuilder()
    t = b.Triangle((0, 0, 0), (2, 0, 0), (0, 2, 0), b.DiffuseLight(1, 1, 1), uv=((0, 0), (1, 0), (0, 1)))
    sc = b.finish(b.Bvh([t]), CAM, (0, 0, 0), RenderConfig(8, 8, 1))
    T = sc.desc.triangles[0]
    assert tuple(T.normal) == (0., 0., 1.) and T.area == 2.
    assert np.allclose(tuple(T.tangent), (1, 0, 0)) and np.allclose(tuple(T.bi_tangent), (0, 1, 0))


def test_diffuse_light_attenuation_factor():
    # attenuation_factor = attenuation_half_length.map(|a| 1. / a) (src/material/mod.rs:335-340); None -> NaN in the ABI
    b = SceneBuilder()
    l1 = b.Sphere((0, 0, 0), 1, b.DiffuseLight(1, 1, 1, 0.25))
    l2 = b.Sphere((3, 0, 0), 1, b.DiffuseLight(1, 1, 1))
    sc = b.finish(b.Bvh([l1, l2]), CAM, (0, 0, 0), RenderConfig(8, 8, 1))
    params = sorted(sc.desc.materials[i].param for i in range(sc.desc.n_materials) if not math.isnan(sc.desc.materials[i].param))
    assert params == [4.0]
    assert sum(math.isnan(sc.desc.materials[i].param) for i in range(sc.desc.n_materials)) == 1


def _walk(desc, ref, boxes_ok, leaves, depth=0):
    k, i = _abi.ref_kind(ref), _abi.ref_index(ref)
    if k == _abi.REF_NONE:
        return None
    if k != _abi.REF_NODE:
        leaves.append(ref)
        arr = {_abi.REF_SPHERE: desc.spheres, _abi.REF_QUAD: desc.quads, _abi.REF_TRIANGLE: desc.triangles,
               _abi.REF_MEDIUM: desc.mediums}[k]
        return np.array(arr[i].bbox.v)
    n = desc.nodes[i]
    bl = _walk(desc, n.left, boxes_ok, leaves, depth + 1)
    br = _walk(desc, n.right, boxes_ok, leaves, depth + 1)
    own = np.array(n.bbox.v)
    for cb in (bl, br):
        if cb is not None:
            boxes_ok.append(bool((own[0::2] <= cb[0::2] + 1e-12).all() and (own[1::2] >= cb[1::2] - 1e-12).all()))
    return own


def test_bvh_builder_invariants():
    """new_bvh (bvh.rs:84-114): every primitive in exactly one leaf, 1 or 2 leaves per leaf node, parent boxes contain the
    children, dfs_index = depth-first leaf order."""
    sc = scenes.sponza_like(RenderConfig(16, 16, 1), n_triangles=3001, texture_size=16)
    d = sc.desc
    ok, leaves = [], []
    _walk(d, d.root, ok, leaves)
    assert all(ok)
    assert len(leaves) == d.n_triangles + d.n_quads and len(set(leaves)) == len(leaves)
    dfs = []
    for r in leaves:
        arr = d.triangles if _abi.ref_kind(r) == _abi.REF_TRIANGLE else d.quads
        dfs.append(arr[_abi.ref_index(r)].dfs_index)
    assert dfs == list(range(len(leaves)))
    for i in range(d.n_nodes):
        lk, rk = _abi.ref_kind(d.nodes[i].left), _abi.ref_kind(d.nodes[i].right)
        assert lk != _abi.REF_NONE
        if rk == _abi.REF_NONE:
            assert lk != _abi.REF_NODE          # (Leaf, None)
        elif lk == _abi.REF_NODE or rk == _abi.REF_NODE:
            pass                                # (Node, Node), or a nested Bvh inlined as a node
    assert 12 <= sc.tree_depth <= 40


def test_bvh_split_rule_small_case():
    """Midpoint split on the axis of largest centroid spread (bvh.rs:116-162): 4 spheres on a line split 2 | 2, and a
    cluster that cannot be split by the midpoint falls back to the median."""
    b = SceneBuilder()
    m = b.DiffuseLight(1, 1, 1)
    ids = [b.Sphere((x, 0, 0), .1, m) for x in (0., 1., 10., 11.)]
    sc = b.finish(b.Bvh(ids), CAM, (0, 0, 0), RenderConfig(8, 8, 1))
    d = sc.desc
    root = d.nodes[_abi.ref_index(d.root)]
    left, right = d.nodes[_abi.ref_index(root.left)], d.nodes[_abi.ref_index(root.right)]
    lx = sorted(d.spheres[_abi.ref_index(r)].center[0] for r in (left.left, left.right))
    rx = sorted(d.spheres[_abi.ref_index(r)].center[0] for r in (right.left, right.right))
    assert lx == [0., 1.] and rx == [10., 11.]
    b = SceneBuilder()
    m = b.DiffuseLight(1, 1, 1)
    ids = [b.Sphere((5., 5., 5.), .1, m) for _ in range(5)]  # identical centres: centre == 0 -> split at len/2
    sc = b.finish(b.Bvh(ids), CAM, (0, 0, 0), RenderConfig(8, 8, 1))
    assert sc.desc.n_spheres == 5 and sc.tree_depth <= 4


def test_get_lights_is_depth_first_and_nested_bvh_is_inlined():
    sc = scenes.create_test_scene(RenderConfig(20, 10, 1))
    d = sc.desc
    assert d.n_lights == 3 and d.n_mediums == 1
    kinds = sorted(_abi.ref_kind(d.lights[i]) for i in range(3))
    assert kinds == [_abi.REF_SPHERE, _abi.REF_QUAD, _abi.REF_TRIANGLE]
    # lights appear in depth-first leaf order
    order = []
    for i in range(3):
        r = d.lights[i]
        arr = {_abi.REF_SPHERE: d.spheres, _abi.REF_QUAD: d.quads, _abi.REF_TRIANGLE: d.triangles}[_abi.ref_kind(r)]
        order.append(arr[_abi.ref_index(r)].dfs_index)
    assert order == sorted(order)
    assert d.n_triangles == 127 and d.n_quads == 20 and d.n_spheres == 2  # 125 + 2 triangles; 1 + 6 + 6 + 6 + 1 quads


def test_camera_new():
    """Camera::new (src/camera.rs:47-74)."""
    sc = scenes.cornell_box(RenderConfig(400, 400, 1))
    c = sc.desc.camera
    assert tuple(c.origin) == (278., 278., -800.)
    h = 2 * math.tan(math.radians(40.) / 2) * 800.
    assert np.allclose(np.abs(tuple(c.horizontal)), (h, 0, 0)) and np.allclose(tuple(c.vertical), (0, h, 0))
    assert np.allclose(np.array(tuple(c.lower_left_corner)) + np.array(tuple(c.horizontal)) / 2 +
                       np.array(tuple(c.vertical)) / 2, (278., 278., 0.))
    assert c.lens_radius == 0.


def test_scene_without_light_is_the_reference_error():
    # tests/integration_tests.rs:174-193: Err("Scene should have at least one light")
    sc = scenes.create_simple_test_scene(RenderConfig(20, 10, 100), add_light=False)
    assert sc.desc.n_lights == 0
    with pytest.raises(HostError, match="Scene should have at least one light"):
        sc.ray_trace()
    with pytest.raises(RuntimeError, match="Scene should have at least one light"):
        orc.render(sc, 0, 1, pu.SEED)


def test_bad_ids_are_errors_not_crashes():
    b = SceneBuilder()
    with pytest.raises(HostError):
        b.Lambertian(99)
    with pytest.raises(HostError):
        b.Sphere((0, 0, 0), 1., 5)
    with pytest.raises(HostError):
        b.Bvh([3])


def test_to_rgb_color_host_matches_reference_kat():
    lib = _abi.load_host()
    out = (C.c_uint8 * 3)()
    lib.solh_to_rgb_color(_abi.d3((0., .3, 1.)), 1, out)
    assert tuple(out) == (0, 140, 255)  # src/util/rgb_color.rs:58
    lib.solh_to_rgb_color(_abi.d3((0., .3, 1.)), 2, out)
    assert tuple(out) == (0, 99, 181)   # src/util/rgb_color.rs:59


def test_load_normal_texture_detects_height_maps():
    # src/material/texture.rs:182-203: wall_n.png is a normal map, sponza-h.jpg a height map (converted by Sobel)
    b = SceneBuilder()
    n = scenes.load_image("textures/wall_n.png")
    hmap = scenes.load_image("textures/sponza-h.jpg")
    t1 = b.load_normal_texture(n)
    t2 = b.load_normal_texture(hmap)
    m1, m2 = b.Lambertian(b.SolidColor(1, 1, 1), t1), b.Lambertian(b.SolidColor(1, 1, 1), t2)
    s = [b.Sphere((0, 0, 0), 1, m1), b.Sphere((3, 0, 0), 1, m2), b.Sphere((0, 9, 0), 1, b.DiffuseLight(1, 1, 1))]
    sc = b.finish(b.Bvh(s), CAM, (0, 0, 0), RenderConfig(8, 8, 1))
    d = sc.desc
    texels = np.ctypeslib.as_array(d.texels, shape=(d.n_texel_bytes,))
    imgs = [d.textures[i] for i in range(d.n_textures) if d.textures[i].kind == _abi.TEX_IMAGE]
    assert len(imgs) == 2
    a = texels[imgs[0].texel_offset: imgs[0].texel_offset + n.size].reshape(n.shape)
    assert (a == n).all()  # normal map kept as is
    bimg = texels[imgs[1].texel_offset: imgs[1].texel_offset + hmap.size].reshape(hmap.shape)
    assert (bimg != hmap).any() and bimg[..., 2].mean() > 128  # converted: blue (z) dominant
