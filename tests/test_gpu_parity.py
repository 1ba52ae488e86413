"""Parity tests proper: the HIP path (through the C ABI) against the fp32 oracle on the same seeded inputs, against the
reference's golden images, and - at BASELINE.json's full sizes - through size-independent properties.

Tolerance (north star): per-channel relative error <= 1e-5 at a fixed seed; the only arithmetic difference between the two
sides is the re-association of the throughput product (DESIGN.md "Flattened recursion"), a few ulp.
"""
import os

import numpy as np
import pytest
from PIL import Image

import image_metric as im
import orc
import parity_util as pu
from ref_cases import CASES
from solstrale_amd import (AlbedoShader, DeviceError, DeviceScene, NormalShader, PathTracingShader, RenderConfig, SceneBuilder,
                           SimpleShader, CameraConfig, _abi, scenes)

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "expected")


def gpu_render(scene, spp, first=0, seed=pu.SEED):
    with DeviceScene(scene) as ds:
        ds.render(first, spp, seed)
        return ds.read()


def assert_parity(scene, spp, rect=None, max_bad=0, first=0):
    img = gpu_render(scene, spp, first)
    ref, _ = orc.render(scene, first, spp, pu.SEED, real=orc.ORC_F32, rect=rect)
    res = pu.compare(img, ref, spp, rect)
    assert np.isfinite(img).all()
    assert res["bad_pixels"] <= max_bad, res
    assert res["max_rel"] <= pu.REL_TOL, res
    assert res["rmse_mean_good"] < 1e-5, res
    return res


# ---- BASELINE configs ------------------------------------------------------------------------------------------
def test_c1_cornell_full():
    """configs[0]: Cornell box 400x400, 50 spp, every pixel."""
    res = assert_parity(scenes.cornell_box(RenderConfig(400, 400, 50)), 50)
    assert res["rmse_mean"] < 1e-5  # north star: per-channel RMSE < 1e-5 vs CPU at fixed seed


def test_c2_cornell_spheres_crop():
    """configs[1] at full resolution and scene size, 128x128 crop x 16 spp against the oracle."""
    assert_parity(scenes.cornell_spheres(RenderConfig(1920, 1080, 16)), 16, rect=(900, 500, 1028, 628))


def test_c3_sponza_class_crop():
    """configs[2]: 262 267 triangles, 1080p, 128x128 crop x 16 spp."""
    sc = scenes.sponza_like(RenderConfig(1920, 1080, 16))
    assert sc.desc.n_triangles == scenes.SPONZA_TRIANGLES
    assert_parity(sc, 16, rect=(900, 500, 1028, 628))
    assert_parity(sc, 4, rect=(0, 952, 128, 1080))  # a corner: floor + walls, other BVH regions


def test_c3_heterogeneous_mesh_crop():
    """The STRESS mesh of configs[2] (bench.py --mesh-preset heterogeneous): the same atrium and triangle count with the statistics of
    a hand-modelled asset - 3 m wall triangles beside leaves 20 000 times smaller, rails and rods of aspect 60:1 to 300:1, rotated
    drapes and arches. The default (GPU-built, pre-split) tree against the oracle, which walks the reference's own tree
    (src/hittable/bvh.rs:84-180) and knows nothing of split references: 128x128 crops x 16 spp at 1080p."""
    sc = scenes.sponza_like(RenderConfig(1920, 1080, 16), mesh="heterogeneous")
    assert sc.desc.n_triangles == scenes.SPONZA_TRIANGLES
    assert_parity(sc, 16, rect=(900, 500, 1028, 628))   # across the hall: rods, ropes, drapes, columns
    assert_parity(sc, 4, rect=(0, 952, 128, 1080))      # a corner: the large floor and wall triangles
    sc = scenes.sponza_like(RenderConfig(1920, 1080, 16), mesh="heterogeneous", camera="interior")
    assert_parity(sc, 8, rect=(1000, 300, 1128, 428))   # under the gallery: capitals, rails, balusters


def test_needle_triangles_do_not_make_the_tree_matter(monkeypatch):
    """fp32 Moller-Trumbore (triangle.rs:119-140 in single precision) on triangles of aspect 300:1 - the tie rods of the heterogeneous
    atrium - accepts rays that miss the triangle by hundreds of round-3 box pads, and whether such a phantom was seen depended on the
    boxes that led to the test: at 1080p, pixel (1350, 137) sample 8 - a "hit" 1e-4 beyond a rod's tip, inside the device's
    quantised leaf box and outside the oracle's exact one - and pixel (1895, 966) sample 26 - a ray skimming a rod's plane, seen by
    the unsplit tree and the oracle, not by the pre-split tree. With the needle rule of the fp32 contract (include/solstrale_hip.h:
    fatter pad + the consistency of the ray's and the triangle's point of a hit, in the oracle's float instantiation and on the
    device) every tree and the oracle agree: those two paths ray by ray, the pre-split and the unsplit frame bit for bit at the size
    where one pixel used to differ, and a full frame against the oracle."""
    sc = scenes.sponza_like(RenderConfig(1920, 1080, 64), mesh="heterogeneous")
    with DeviceScene(sc) as ds:
        assert ds.info()["strict_triangles"] and ds.info()["split_references"] > 10000
        for (x, y, s_) in ((1350, 137, 8), (1895, 966, 26), (578, 56, 468), (1350, 137, 504)):
            rows, colour = ds.debug_path(x, y, s_, pu.SEED)
            orows, ocolour = orc.debug_path(sc, x, y, s_, pu.SEED, real=orc.ORC_F32)
            assert np.abs(colour - ocolour).max() <= 1e-6 * max(1e-3, float(np.abs(ocolour).max())), (x, y, s_, colour, ocolour)
            # (the oracle, like the reference, also follows a scattered ray whose scattering pdf is 0 - weight 0, the device ends the path there)
            assert 0 < len(rows) <= len(orows), (x, y, s_, len(rows), len(orows))
            for r, o in zip(rows, orows):
                assert r[6] == o[6] or abs(r[6] - o[6]) <= 1e-6 * abs(o[6]), (x, y, s_, r[:8], o[:8])  # the same hit parameter, ray by ray
        ds.render(0, 64, pu.SEED)
        split = ds.read()
    monkeypatch.setenv("SOL_SPLIT", "0")
    with DeviceScene(sc) as ds:
        assert ds.info()["strict_triangles"] and ds.info()["split_references"] == 0
        ds.render(0, 64, pu.SEED)
        plain = ds.read()
    monkeypatch.delenv("SOL_SPLIT")
    assert (split == plain).all(), int((split != plain).any(axis=-1).sum())
    small = scenes.sponza_like(RenderConfig(480, 270, 16), mesh="heterogeneous")
    assert_parity(small, 16)  # the whole frame
    # scenes without needles keep the round-3 contract (no rule, the thin pad)
    with DeviceScene(scenes.sponza_like(RenderConfig(64, 64, 1))) as ds:
        assert not ds.info()["strict_triangles"]


@pytest.mark.parametrize("env", [{"SOL_BVH": "ref"}, {"SOL_BVH": "sah"}, {"SOL_SWITCH": "0"}, {"SOL_SWITCH": "40"},
                                 {"SOL_BVH": "sah", "SOL_SLOTS": "octant"}, {"SOL_SPLIT": "0"}, {"SOL_SPLIT": "100", "SOL_SPLIT_SLACK": "0"}],
                         ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_tree_and_schedule_variants_are_bit_identical(env, monkeypatch):
    """The closest hit does not depend on the tree (reference topology, SAH rebuild, slot assignment), and the image is a
    pure function of (scene, seed): every world-tree choice and every search/shade switch threshold must reproduce the default
    build's frame bit for bit ("tree independence"; the kernel variants: test_wavefront_ab_kernels_are_bit_identical)."""
    frames = {}
    for name, make in (("c2", scenes.cornell_spheres), ("c3", scenes.sponza_like), ("c3h", lambda rc: scenes.sponza_like(rc, mesh="heterogeneous"))):
        sc = make(RenderConfig(480, 270, 16))
        frames[name] = (sc, gpu_render(sc, 16))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for name, (sc, want) in frames.items():
        got = gpu_render(sc, 16)
        assert (got == want).all(), (name, env, int((got != want).sum()))


def test_wavefront_ab_kernels_are_bit_identical():
    """The staged wavefront forms of the path (north star: "ballot/prefix compaction of active-ray queues into generate /
    intersect / shade stages") live in the A/B build of the library only (_build_ab/, -DSOL_AB_KERNELS: kernel 2 = wave-private
    pool with an LDS ray queue, 3 = separate shade and trace kernels): measured slower than the in-register search/service switch
    of the product kernel, kept as variants; so does kernel 4, the POOL kernel of round 5 (csrc/sol_pool.hip: a second path
    context per lane in LDS, handed out wave-wide; one-dword stack entries; sample-granular work items - measured slower,
    profiles/r05_pool_kernel_ab.txt). Their frames, and the A/B library's own kernel 1, must equal the product library's frames bit for
    bit - on the sphere scene, the atrium, the heterogeneous atrium (needle triangles: the STRICT variants) and the reference's test scene
    (constant medium), at a ragged frame size; the product library itself refuses 2, 3 and 4."""
    import json
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import frame_crc
    want = frame_crc.crcs([1])
    ab_dir = os.path.join(os.path.dirname(_abi.BUILD_DIR), "_build_ab")
    assert os.path.exists(os.path.join(ab_dir, "libsolstrale_hip.so")), "the A/B library is missing: __graft_entry__.build() makes it"
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "frame_crc.py"), "1", "2", "3", "4"],
                       env=dict(os.environ, SOLSTRALE_BUILD_DIR=ab_dir), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    for key, crc in got.items():
        assert crc == want[key.split("/")[0] + "/1"], (key, got, want)
    assert len(got) == 4 * len(want) - 2  # (kernels 2 and 3 do not implement the needle rule: no frames of the heterogeneous atrium)
    with DeviceScene(scenes.cornell_box(RenderConfig(16, 16, 1))) as ds:
        for k in (2, 3, 4):
            with pytest.raises(DeviceError) as e:
                ds.set_option(_abi.OPT_KERNEL, k)
            assert e.value.code == _abi.SOL_EINVAL and "SOL_AB_KERNELS" in str(e.value)


def test_device_built_tree_renders_the_same_frames():
    """SolCreateOptions.world_tree = SOL_TREE_DEVICE: the world tree built by the GPU kernels of sol_build.hip instead of the
    host builders. Closest hits do not depend on the tree, so the frames must be bit-identical (reference rule the device
    stays results-compatible with: src/hittable/bvh.rs:165-180 + src/util/interval.rs:67-69, the tie rule)."""
    hetero = lambda rc: scenes.sponza_like(rc, mesh="heterogeneous")  # (pre-splitting at work: thousands of extra references)
    hetero.__name__ = "sponza_like_heterogeneous"
    for make, cfg in ((scenes.cornell_box, RenderConfig(200, 200, 16)), (scenes.cornell_spheres, RenderConfig(480, 270, 16)),
                      (scenes.sponza_like, RenderConfig(480, 270, 16)), (scenes.create_test_scene, RenderConfig(200, 100, 16)),
                      (scenes.create_obj_scene, RenderConfig(200, 100, 8)), (hetero, RenderConfig(480, 270, 16))):
        sc = make(cfg)
        with DeviceScene(sc, world_tree=_abi.TREE_HOST_PROBE) as ds:  # the four host candidates + probe
            ds.render(0, cfg.samples_per_pixel, pu.SEED)
            want = ds.read()
            assert ds.build_times()["device_tree"] == 0
        with DeviceScene(sc, world_tree=_abi.TREE_DEVICE) as ds:
            ds.render(0, cfg.samples_per_pixel, pu.SEED)
            got = ds.read()
            bt = ds.build_times()
        assert (got == want).all(), (make.__name__, int((got != want).sum()))
        assert bt["device_tree"] > 0 and bt["probes"] < bt["device_tree"] + 1.0


def test_ties_keep_hit_and_material_together():
    """Hits at EXACTLY equal t: the reference keeps the later primitive in depth-first leaf order (bvh.rs:172-178 with the
    inclusive Interval::contains). Coincident primitives of different colours make every hit a tie: geometry AND material must
    come from the later one, for quads, triangles and spheres, whichever is tested first (both orders are built). (An experiment
    that carried the material index in the hit record lost it exactly here - DESIGN.md 9 - and only a 2-pixel CRC difference
    of a full frame showed it: hence this test.)"""
    for order in (0, 1):
        b = SceneBuilder()
        red, white = b.Lambertian(b.SolidColor(1., 0., 0.)), b.Lambertian(b.SolidColor(.9, .9, .9))
        m = (red, white) if order == 0 else (white, red)
        world = []
        for k in range(2):  # the same quad, triangle and sphere twice: the second (later leaf) must win everywhere
            world.append(b.Quad((-3., 0., -3.), (2., 0., 0.), (0., 2., 0.), m[k]))
            world.append(b.Triangle((0., 0., -3.), (2., 0., -3.), (1., 2., -3.), m[k]))
            world.append(b.Sphere((3.5, 1., -3.), 1., m[k]))
        world.append(b.Sphere((0., 50., 20.), 10., b.DiffuseLight(5, 5, 5)))
        cam = CameraConfig(50., 0., (0.5, 1., 4.), (0.5, 1., -3.), (0, 1, 0))
        sc = b.finish(b.Bvh(world), cam, (.1, .1, .1), RenderConfig(160, 80, 4, AlbedoShader()))
        img = gpu_render(sc, 4)
        ref, _ = orc.render(sc, 0, 4, pu.SEED, real=orc.ORC_F32)
        assert pu.compare(img, ref, 4)["bad_pixels"] == 0
        want = np.array([1., 0., 0.] if order == 1 else [.9, .9, .9], dtype=np.float32) * 4
        inside = np.abs(ref - want.astype(np.float64)).max(axis=-1) < 1e-5  # pixels whose four samples all hit a primitive
        assert inside.sum() > 1000 and np.allclose(img[inside], want, atol=1e-5), order
        loser = np.array([.9, .9, .9] if order == 1 else [1., 0., 0.]) * 4
        assert not (np.abs(img - loser).max(axis=-1) < 1e-5).any()  # the earlier primitive's colour shows nowhere
        # and through the path-tracing shader (materials feed the throughput)
        sc2 = b.finish(b.Bvh(world), cam, (.1, .1, .1), RenderConfig(160, 80, 8))
        assert_parity(sc2, 8)


def test_heavy_first_work_order_changes_nothing(monkeypatch):
    """Scenes with long paths (glass) get their costly pixel blocks scheduled first (cost probe at scene creation,
    sol_path.h decode_item_ordered); the frame must be the one of the plain order, for one rank and for a partition."""
    sc = scenes.create_test_scene(RenderConfig(400, 200, 20))  # glass sphere: heavy blocks exist
    want = None
    for order in ("0", "1"):
        monkeypatch.setenv("SOL_ORDER", order)
        with DeviceScene(sc) as ds:
            ds.render(0, 20, pu.SEED)
            full = ds.read()
            parts = []
            for r in range(3):
                ds.set_partition(r, 3)
                ds.clear()
                ds.render(0, 20, pu.SEED)
                parts.append(ds.read())
            merged = parts[0] + parts[1] + parts[2]  # sol_read writes 0 for the pixels of other ranks
        want = full if want is None else want
        assert (full == want).all() and (merged == want).all(), order


def test_c5_shape_dielectric_metal_crop():
    """configs[4] shape: triangle mesh + dielectric and metal spheres; crop against the oracle."""
    assert_parity(_atrium_with_bsdfs(RenderConfig(640, 360, 16)), 16, rect=(256, 116, 384, 244))


def test_c5_statue_class_crop():
    """configs[4] stand-in at full size: ~1.09 M triangles, Metal(0.1) body, Dielectric(1.5) head and orb; 1080p, 128x128 crops."""
    sc = scenes.statue_like(RenderConfig(1920, 1080, 16))
    assert abs(sc.desc.n_triangles - scenes.STATUE_TRIANGLES) < 2000
    assert_parity(sc, 16, rect=(896, 476, 1024, 604))   # body + drapery
    assert_parity(sc, 8, rect=(900, 60, 1028, 188))     # glass head (x 830-1020, y 55-255) against the background


def test_stress_cameras_and_the_profiling_workload():
    """The stress variants bench.py reports beside the headline (--camera-preset interior / closeup) and the reference's own profiling
    workload (src/bin/profiling.rs:14-37: create_test_scene at 800x400): one 128x128 crop each against the oracle, and the path
    statistics the bench line carries (sol_path_stats) - every camera ray of the interior view hits, its paths are longer."""
    inner = scenes.sponza_like(RenderConfig(1920, 1080, 8), camera="interior")
    assert_parity(inner, 8, rect=(896, 476, 1024, 604))
    assert_parity(scenes.statue_like(RenderConfig(1920, 1080, 8), camera="closeup"), 8, rect=(896, 476, 1024, 604))
    assert_parity(scenes.create_test_scene(RenderConfig(800, 400, 16)), 16, rect=(336, 136, 464, 264))
    stats = {}
    for cam in ("default", "interior"):
        with DeviceScene(scenes.sponza_like(RenderConfig(480, 270, 16), camera=cam)) as ds:
            ds.render(0, 16, pu.SEED, counted=True)
            stats[cam] = (ds.path_stats(), ds.stats())
    for ps, st in stats.values():
        assert ps["samples"] == st["samples"] == 480 * 270 * 16 and abs(sum(ps["rays_per_path_histogram"].values()) - 1.0) < 1e-9
    assert stats["interior"][0]["primary_hit_fraction"] > 0.999 > stats["default"][0]["primary_hit_fraction"] > 0.5
    assert stats["interior"][1]["rays"] > 1.15 * stats["default"][1]["rays"]  # (diffuse paths stay short under the reference's estimator: half the bounces draw a light direction, and one below the horizon ends the path)
    h = stats["default"][0]["rays_per_path_histogram"]
    mean_lo = h["1"] + 2 * h["2"] + 3 * h["3-4"] + 5 * h["5-8"] + 9 * h["9-16"] + 17 * h["17+"]
    mean_hi = h["1"] + 2 * h["2"] + 4 * h["3-4"] + 8 * h["5-8"] + 16 * h["9-16"] + 51 * h["17+"]
    assert mean_lo <= stats["default"][1]["rays"] / stats["default"][1]["samples"] <= mean_hi  # the histogram brackets rays per sample


def test_c5_statue_hdri_crop():
    """configs[4] AS BASELINE.json NAMES IT - "+ HDRI env light": the statue stand-in under the procedural HDR environment map
    (EXTENSION, SolSceneDesc::env_*: the reference's miss branch src/renderer/mod.rs:197-204 returns a constant; the direction ->
    texel mapping is calculate_sphere_uv, sphere.rs:134-140). Glass and metal paths ending on environment texels at 1.09 M
    triangles; the two crops of test_c5_statue_class_crop plus one of the glass orb."""
    sc = scenes.statue_like(RenderConfig(1920, 1080, 16), environment=True)
    assert sc.desc.env_width > 0 and abs(sc.desc.n_triangles - scenes.STATUE_TRIANGLES) < 2000
    assert_parity(sc, 16, rect=(896, 476, 1024, 604))   # body + drapery (metal: reflections of the sky)
    assert_parity(sc, 8, rect=(900, 60, 1028, 188))     # glass head: refracted sky texels
    assert_parity(sc, 8, rect=(1150, 860, 1278, 988))   # the glass orb (centre 1214,990, radius 146 px): sky through glass, floor around it


def _atrium_with_bsdfs(rc):
    b = SceneBuilder()
    mats = [b.Lambertian(b.SolidColor(.7, .6, .5)), b.Lambertian(b.SolidColor(.3, .5, .7))]
    tri, uv = scenes._grid(lambda u, v: (-8 + 16 * u, 0 * u, -8 + 16 * v), 40, 40)
    first, n = b.triangles(tri, np.full(len(tri), mats[0], np.int32), uv)
    tri2, uv2 = scenes._grid(lambda u, v: (-8 + 16 * u, 6 * v, -8 + 0 * u + 0.3 * np.sin(9 * u)), 60, 20)
    f2, n2 = b.triangles(tri2, np.full(len(tri2), mats[1], np.int32), uv2)
    model = b.Bvh_range(first, n + n2)
    glass = b.Sphere((-1.5, 1.2, 0.), 1.2, b.Dielectric(b.SolidColor(1., 1., 1.), None, 1.5))
    metal = b.Sphere((1.8, 1.0, -1.), 1.0, b.Metal(b.SolidColor(.9, .8, .6), None, 0.1))
    light = b.Quad((-3., 9., -3.), (6., 0, 0), (0, 0, 6.), b.DiffuseLight(12., 12., 12.))
    cam = CameraConfig(40., 0., (0., 3., 9.), (0., 1., 0.), (0, 1, 0))
    return b.finish(b.Bvh([model, glass, metal, light]), cam, (.3, .4, .6), rc)


def test_sphere_phantom_hits_are_layout_independent():
    """Regression: camera rays with one direction component near zero grazing a sphere. The fp32 sphere quadratic (origin 800
    units away, |d| = 800) reports points up to 0.02 units off the sphere; before the hit-point-in-own-box rule of the fp32
    contract, whether such a phantom was seen depended on which boxes a traversal tested (reference tree vs extra leaf boxes
    vs 8-wide quantised boxes) - found as 5 pixels of 2 073 600 differing at 64 spp. The paths must match ray for ray."""
    sc = scenes.cornell_spheres(RenderConfig(1920, 1080, 64))
    cases = [(958, 309, 32), (754, 539, 37), (959, 472, 35), (961, 165, 47), (959, 471, 63)]
    with DeviceScene(sc) as ds:
        for (x, y, s) in cases:
            g, gc = ds.debug_path(x, y, s, pu.SEED)
            o, oc = orc.debug_path(sc, x, y, s, pu.SEED)
            n = len(g)  # the device stops a path whose throughput became exactly 0; the reference traces on (result 0 either way)
            assert 1 <= n <= len(o), (x, y, s, len(g), len(o))
            assert (g[:, :8].view(np.uint32) == o[:n, :8].view(np.uint32)).all(), (x, y, s)
            assert np.abs(gc - oc).max() <= 1e-5 * max(1e-3, np.abs(oc).max())


def test_negative_radius_spheres():
    """The hollow-glass idiom (a sphere of negative radius: sphere.rs uses r^2 and a min/max box only) through the device: the fp32 rules
    that use the radius itself use |r| (tests/test_fp32_contract.py::test_a_negative_radius_is_its_positive_twin ties float to f64)."""
    from test_fp32_contract import hollow_glass_scene
    assert_parity(hollow_glass_scene(RenderConfig(240, 160, 16, PathTracingShader(12))), 16)


# ---- the reference's own scenes: every material, primitive and shader -------------------------------------------------
def test_reference_test_scene_all_features():
    """tests/scenes.rs:17-122: image texture, glass, rotated boxes, ConstantMedium, nested BVH, sphere + quad + triangle
    lights, aperture 0.1."""
    assert_parity(scenes.create_test_scene(RenderConfig(200, 100, 25)), 25)


@pytest.mark.parametrize("shader", [AlbedoShader(), NormalShader(), SimpleShader()], ids=["albedo", "normal", "simple"])
def test_single_hit_shaders(shader):
    # shader.rs:129-215
    assert_parity(scenes.create_test_scene(RenderConfig(200, 100, 4, shader)), 4)


@pytest.mark.parametrize("blend", [0., .5, 1.])
def test_blend_material(blend):
    assert_parity(scenes.create_blend_material_scene(RenderConfig(128, 128, 16), blend), 16)


def test_normal_mapping_quad_and_sphere():
    assert_parity(scenes.create_normal_mapping_scene(RenderConfig(128, 128, 16), (30., 30., 30.), True), 16)
    assert_parity(scenes.create_normal_mapping_sphere_scene(RenderConfig(128, 128, 16), (-30., 30., 30.)), 16)


@pytest.mark.parametrize("half", [0.1, 0.8, None])
def test_light_attenuation(half):
    assert_parity(scenes.create_light_attenuation_scene(RenderConfig(128, 128, 16), half), 16)


def test_fp32_records_start_opposite_the_longest_edge():
    """fp32 contract (include/solstrale_hip.h, sol_triangle_rotation): the device's triangle records, like the oracle's float ones,
    start at the vertex opposite the longest edge - all three rotations, texture coordinates rotated along, and a triangle light
    that keeps the reference's order (tests/test_fp32_contract.py has the scene and checks the float oracle against f64)."""
    from test_fp32_contract import rotated_record_scene
    assert_parity(rotated_record_scene(RenderConfig(160, 120, 8, AlbedoShader())), 8)
    assert_parity(rotated_record_scene(RenderConfig(160, 120, 32, PathTracingShader(8))), 32)
    sc = rotated_record_scene(RenderConfig(96, 72, 64, PathTracingShader(8)))
    f64, _ = orc.render(sc, 0, 64, pu.SEED, real=orc.ORC_F64)
    img = gpu_render(sc, 64)
    assert abs(img.mean() - f64.mean()) < 3e-3 * f64.mean(), (img.mean(), f64.mean())


@pytest.mark.parametrize("aspect", [20, 300, 2000])
def test_needle_shaped_lights(aspect):
    """A strip light of two needle triangles (tests/test_fp32_contract.py): the device agrees with the float oracle pixel by pixel and
    with the double one in the mean - its hits go through the rotated records, its samples come from the reference's frame
    (DevScene::light_tri)."""
    from test_fp32_contract import strip_light_scene
    sc = strip_light_scene(aspect, RenderConfig(96, 64, 128, PathTracingShader(8)))
    assert_parity(sc, 128)
    f64, _ = orc.render(sc, 0, 128, pu.SEED, real=orc.ORC_F64)
    img = gpu_render(sc, 128)
    assert abs(img.mean() - f64.mean()) < 1e-4 * f64.mean(), (aspect, img.mean(), f64.mean())


def test_uv_wrapping():
    # triangle UVs outside [0,1] incl. negative (tests/scenes.rs:196-230)
    assert_parity(scenes.create_uv_scene(RenderConfig(128, 128, 8)), 8)


def test_bvh_bench_scene_with_and_without_nested_bvh():
    # tests/scenes.rs:125-167: the same triangles as a nested Bvh or directly in the world give the same image
    a = gpu_render(scenes.new_bvh_test_scene(RenderConfig(120, 60, 8), True, 300), 8)
    b = gpu_render(scenes.new_bvh_test_scene(RenderConfig(120, 60, 8), False, 300), 8)
    assert np.abs(a - b).max() <= 1e-5 * np.abs(a).max()
    assert_parity(scenes.new_bvh_test_scene(RenderConfig(120, 60, 8), True, 300), 8)


# ---- reference golden images through the GPU ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,factory,w,h,ref_spp,_", CASES, ids=[c[0] for c in CASES])
def test_gpu_matches_reference_golden(name, factory, w, h, ref_spp, _):
    scene = factory(ref_spp)
    sums = gpu_render(scene, ref_spp)
    with np.errstate(invalid="ignore"):
        actual = im.sums_to_rgb8(sums, ref_spp)
    expected = np.asarray(Image.open(os.path.join(GOLDEN, f"out_expected_{name}.jpg")).convert("RGB"))
    score = im.compare_output(actual, expected)
    assert score > im.THRESHOLD, f"Comparison score for {name} is: {score}"


# ---- edge cases -------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h", [(2, 2), (9, 5), (203, 97), (64, 8)])
def test_ragged_image_sizes(w, h):
    """Image sizes that are not multiples of the 8x8 work block, and the smallest legal image."""
    assert_parity(scenes.cornell_box(RenderConfig(w, h, 20)), 20)


@pytest.mark.parametrize("spp", [1, 15, 16, 17, 33])
def test_sample_counts_around_the_chunk_size(spp):
    assert_parity(scenes.cornell_box(RenderConfig(96, 96, spp)), spp)


def test_sample_ranges_are_additive():
    """sol_render ADDS samples [first, first+n): rendering [0,32) equals [0,16) then [16,32) bit for bit (chunk sums),
    and a sample range rendered alone equals the oracle's same range."""
    sc = scenes.cornell_box(RenderConfig(96, 96, 32))
    with DeviceScene(sc) as ds:
        ds.render(0, 32, pu.SEED)
        whole = ds.read()
        ds.clear()
        ds.render(0, 16, pu.SEED)
        ds.render(16, 16, pu.SEED)
        parts = ds.read()
        ds.clear()
        ds.render(16, 7, pu.SEED)
        mid = ds.read()
    assert (whole == parts).all()
    ref, _ = orc.render(sc, 16, 7, pu.SEED, real=orc.ORC_F32)
    assert pu.compare(mid, ref, 7)["bad_pixels"] == 0


def test_seed_changes_the_image_and_is_reproducible():
    sc = scenes.cornell_box(RenderConfig(64, 64, 8))
    a = gpu_render(sc, 8, seed=1)
    b = gpu_render(sc, 8, seed=1)
    c = gpu_render(sc, 8, seed=2)
    assert (a == b).all() and (a != c).any()


def test_max_depth_zero_and_one():
    # depth >= max_depth is tested after the hit query (shader.rs:70-72): depth 0 -> black where anything is hit
    for md in (0, 1, 3):
        assert_parity(scenes.cornell_box(RenderConfig(64, 64, 8, PathTracingShader(md))), 8)


def _sphere_chain(b, n=50):
    m = b.Lambertian(b.SolidColor(.8, .8, .8))
    ids = [b.Sphere((float(x), 0.3 * (x % 3), 0.), 0.45, m) for x in range(n)]
    inner = b.Bvh(ids[:2])
    for k in range(2, n):
        inner = b.Bvh([inner, ids[k]]) if k % 2 else b.Bvh([ids[k], inner])
    return inner


@pytest.mark.parametrize("bvh", ["ref", "sah"])
def test_deep_world_tree(bvh, monkeypatch):
    """A BVH nested 48 levels deep (Bvh::new([sphere, Bvh::new([sphere, ...])]), each nested Bvh inlined as a node) as the world.
    SOL_BVH=ref makes the device collapse the reference's own topology (its default is a SAH rebuild, which re-balances the
    chain). The 7-wide search keeps one sibling group per level, so even this chain stays inside the LDS stack."""
    monkeypatch.setenv("SOL_BVH", bvh)
    b = SceneBuilder()
    inner = _sphere_chain(b)
    light = b.Sphere((0., 1e4, 0.), 3e3, b.DiffuseLight(3, 3, 3))
    cam = CameraConfig(12., 0., (-30., 0.4, 0.3), (50., 0.3, 0.), (0, 1, 0))  # looks along the row: every level is entered
    sc = b.finish(b.Bvh([inner, light]), cam, (.1, .1, .1), RenderConfig(64, 64, 4))
    assert sc.tree_depth > 40
    with DeviceScene(sc) as ds:
        ds.render(0, 4, pu.SEED, counted=True)
        img = ds.read()
        st = ds.stats()
    ref, _ = orc.render(sc, 0, 4, pu.SEED, real=orc.ORC_F32)
    assert pu.compare(img, ref, 4)["bad_pixels"] == 0
    assert st["samples"] == 64 * 64 * 4 and st["rays"] >= st["samples"]


def test_deep_medium_boundary_uses_the_spill_stack():
    """The same 48-level chain as the BOUNDARY of a ConstantMedium: boundary searches walk the reference-shaped 2-wide tree with
    one stack entry per level, deeper than the 32-entry LDS stack - the overflow goes to the global spill area."""
    b = SceneBuilder()
    fog = b.ConstantMedium(_sphere_chain(b), 0.4, (.9, .9, .9))
    floor_ = b.Quad((-10., -1., -10.), (80., 0., 0.), (0., 0., 20.), b.Lambertian(b.SolidColor(.5, .6, .5)))
    light = b.Sphere((0., 1e4, 0.), 3e3, b.DiffuseLight(3, 3, 3))
    cam = CameraConfig(12., 0., (-30., 0.4, 0.3), (50., 0.3, 0.), (0, 1, 0))
    sc = b.finish(b.Bvh([fog, floor_, light]), cam, (.1, .1, .1), RenderConfig(64, 64, 4))
    with DeviceScene(sc) as ds:
        ds.render(0, 4, pu.SEED, counted=True)
        img = ds.read()
        st = ds.stats()
        # the uncounted kernel variant (spill stack + medium), without and with the fine tail: the same frame as the counted one
        from solstrale_amd import _abi
        for tail in (0, 4):
            ds.set_option(_abi.OPT_FINE_TAIL, tail)
            ds.clear()
            ds.render(0, 4, pu.SEED)
            assert (ds.read() == img).all(), tail
    ref, _ = orc.render(sc, 0, 4, pu.SEED, real=orc.ORC_F32)
    assert pu.compare(img, ref, 4)["bad_pixels"] == 0
    assert st["max_stack"] > 32, st  # the spill area was really used


@pytest.mark.parametrize("seed", range(48))
def test_random_scenes_parity(seed):
    """Randomised coverage: every primitive, material (nested Blend, normal maps, image textures), transformation chain, nested
    Bvh, constant medium and light shape in seeded random small scenes; every pixel must match the fp32 oracle."""
    import random_scenes
    sc = random_scenes.random_scene(seed)
    res = assert_parity(sc, 4)  # (tests/tools/random_parity_sweep.py ran seeds 100-1599 and 2000-4499 on MI355X: no pixel over 1e-5)
    assert res["pixels"] == 40 * 32


@pytest.mark.parametrize("seed", range(24))
def test_random_needle_scenes_parity(seed):
    """Seeded random scenes of needle triangles (aspect 40:1 .. 2000:1, any orientation and vertex order, textured or not, some of
    them lights; pinhole and lens cameras): the needle rule, the rotated fp32 records and - where the builder keeps it - pre-splitting
    together; every pixel must match the fp32 oracle (tests/tools/needle_parity_sweep.py ran seeds 0-5999 on MI355X: none over 1e-5)."""
    import random_scenes
    sc = random_scenes.needle_scene(seed)
    with DeviceScene(sc) as ds:
        assert ds.info()["strict_triangles"]
    assert_parity(sc, 8)


def test_world_without_a_tree():
    """`Scene.world` may be a single primitive (no Bvh at all) or a Bvh of one or two primitives: nothing to collapse."""
    for n in (0, 1, 2):
        b = SceneBuilder()
        light = b.Sphere((0., 0., 0.), 1., b.DiffuseLight(2., 3., 4.))
        extra = [b.Quad((-2., -1.5, -2.), (4., 0., 0.), (0., 0., 4.), b.Lambertian(b.SolidColor(.6, .6, .6)))] if n == 2 else []
        world = light if n == 0 else b.Bvh([light] + extra)
        sc = b.finish(world, CameraConfig(40., 0., (0., 1., 6.), (0., 0., 0.), (0, 1, 0)), (.1, .2, .3), RenderConfig(48, 40, 8))
        assert_parity(sc, 8)


def test_degenerate_world_parity():
    """Coincident primitives, dust of 1e-4 spheres and a 5e3 sphere around everything (tests/test_world_tree.py checks the tree's
    structure on the CPU): quantisation grids spanning 8 orders of magnitude must still return the oracle's hits."""
    import test_world_tree as twt
    sc = twt._flat_and_huge()
    assert_parity(sc, 8)


def test_errors_are_codes_not_crashes():
    sc = scenes.cornell_box(RenderConfig(16, 16, 1))
    with DeviceScene(sc) as ds:
        with pytest.raises(DeviceError):
            ds.set_partition(2, 2)
        with pytest.raises(DeviceError):
            ds.bind_accum(12345, 1)  # too small
    with pytest.raises(DeviceError) as e:
        DeviceScene(sc, device=99)
    assert e.value.code == _abi.SOL_EDEVICE


def test_counters_match_definitions():
    sc = scenes.cornell_box(RenderConfig(64, 64, 4))
    with DeviceScene(sc) as ds:
        ds.render(0, 4, pu.SEED, counted=True)
        st = ds.stats()
        counted = ds.read()
        ds.clear()
        ds.render(0, 4, pu.SEED)
        plain = ds.read()
    assert (counted == plain).all()  # instrumentation does not change results
    assert st["samples"] == 64 * 64 * 4
    assert st["rays"] >= st["samples"] and st["node_visits"] > st["rays"] and st["quad_tests"] > 0
    # SolStats::rays = the searches the device runs = the oracle's live_rays (it ends a path where a ScatterPdf level multiplies by zero;
    # the reference - OrcStats::rays - traces on): equal to the ray when every path matches
    _, ost = orc.render(sc, 0, 4, pu.SEED, real=orc.ORC_F32)
    assert ost["samples"] == st["samples"] and ost["rays"] >= ost["live_rays"]
    assert abs(st["rays"] - ost["live_rays"]) <= 1e-3 * ost["live_rays"], (st["rays"], ost["live_rays"], ost["rays"])
    # (18 quads: the collapse may hang the wide nodes in a chain - one inner child each - and then no sibling group is ever pushed)
    assert st["sphere_tests"] == 0 and st["triangle_tests"] == 0 and 0 <= st["max_stack"] <= 21  # two dwords per level of the 7-wide tree


# ---- multi-GPU sharding on one GPU: every rank's tiles, gathered, equal the single-GPU image ---------------------------
@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("balanced", [0, 1], ids=["b_mod_n", "balanced_table"])
def test_tile_partition_is_bit_identical(world, balanced):
    """Block b -> rank b mod world (the default), or - SOL_OPT_BALANCED_PARTITION - the blocks dealt out in the order of their cost in
    the creation probe: either way every pixel is rendered by exactly one rank, with the sums it has in the single-rank frame."""
    import torch
    sc = scenes.create_test_scene(RenderConfig(203, 97, 20))
    whole = gpu_render(sc, 20)
    with DeviceScene(sc) as ds:
        ds.set_option(_abi.OPT_BALANCED_PARTITION, balanced)
        ds.set_partition(0, world)
        n = ds.accum_floats()
        gathered = torch.zeros(world * n, dtype=torch.float32, device="cuda")
        for r in range(world):
            ds.set_partition(r, world)
            assert ds.accum_floats() == n
            ds.bind_accum(gathered.data_ptr() + r * n * 4, n)
            ds.render(0, 20, pu.SEED)
            ds.sync()
            part = ds.read()  # other ranks' pixels are zero
            owned = part.any(axis=-1)
            assert (part[owned] == whole[owned]).all()
            seen = owned.astype(np.int32) if r == 0 else (seen + owned)
        assert (seen <= 1).all() and seen.sum() >= (whole.any(axis=-1)).sum()  # no pixel twice
        ds.set_partition(0, world)
        image = torch.empty(sc.height * sc.width * 3, dtype=torch.float32, device="cuda")
        ds.unpermute(gathered.data_ptr(), world, image.data_ptr())
        ds.sync()
        torch.cuda.synchronize()
        out = image.cpu().numpy().reshape(sc.height, sc.width, 3)
    assert (out == whole).all()


def test_background_blocks_change_no_frame():
    """SolSceneInfo::background_blocks: blocks sol_scene_create proved to see only the background are summed, not traced
    (tests/test_background_blocks.py checks the proof against the oracle). Frames with and without (SOL_OPT_BACKGROUND_BLOCKS) must be
    the same bits: whole and partial chunks, a sample range that starts off a chunk edge, a ragged image, a rank's share, single-hit
    shaders; a counted render always traces everything; and the default build agrees with the oracle on a crop of background and
    silhouette."""
    sc = scenes.statue_like(RenderConfig(1283, 717, 40), n_triangles=200000)
    nb = ((sc.width + 7) // 8) * ((sc.height + 7) // 8)
    with DeviceScene(sc) as ds:
        info = ds.info()
        assert 0.3 * nb < info["background_blocks"] < nb and 0 < info["background_pixels"] <= info["background_blocks"] * 64
        host = solstrale_background_blocks(sc)
        assert host.sum() > 0  # (the device tree's proof may find a few blocks more or fewer than a host tree's)
        for first, n in ((0, 40), (7, 16), (3, 5), (0, 1)):
            frames = []
            for on in (1, 0):
                ds.set_option(_abi.OPT_BACKGROUND_BLOCKS, on)
                ds.clear(); ds.render(first, n, pu.SEED); frames.append(ds.read())
            assert (frames[0] == frames[1]).all(), (first, n, int((frames[0] != frames[1]).any(axis=-1).sum()))
        whole = frames[0]  # (0, 1)
        ds.set_option(_abi.OPT_BACKGROUND_BLOCKS, 1)
        ds.clear(); ds.render(0, 8, pu.SEED, counted=True)
        st = ds.stats()
        assert st["samples"] == sc.width * sc.height * 8  # counted: nothing skipped
        counted = ds.read()
        ds.clear(); ds.render(0, 8, pu.SEED)
        assert (ds.read() == counted).all()
        ds.set_partition(3, 8)
        ds.clear(); ds.render(0, 1, pu.SEED)
        part = ds.read()
        owned = part.any(axis=-1)
        assert owned.sum() > sc.width * sc.height // 10 and (part[owned] == whole[owned]).all()
    for shader in (AlbedoShader(), NormalShader()):
        sc2 = scenes.statue_like(RenderConfig(640, 360, 4, shader), n_triangles=50000)
        with DeviceScene(sc2) as ds:
            assert ds.info()["background_blocks"] > 0
            ds.render(0, 4, pu.SEED); a = ds.read()
            ds.set_option(_abi.OPT_BACKGROUND_BLOCKS, 0); ds.clear(); ds.render(0, 4, pu.SEED)
            assert (ds.read() == a).all()
    with DeviceScene(sc, no_background_blocks=True) as ds:
        assert ds.info()["background_blocks"] == 0
    # a camera that looks away from everything: every block is a background block, no render kernel is launched at all
    b = SceneBuilder()
    away = b.finish(b.Bvh([b.Sphere((0., 0., 9.), 1., b.DiffuseLight(5., 5., 5.)), b.Sphere((1., 0., 12.), 1., b.Lambertian(b.SolidColor(.5, .5, .5)))]),
                    CameraConfig(40., 0., (0., 0., 0.), (0., 0., -1.), (0., 1., 0.)), (.2, .3, .5), RenderConfig(99, 61, 21))
    with DeviceScene(away) as ds:
        assert ds.info()["background_blocks"] == 13 * 8 and ds.info()["background_pixels"] == 99 * 61
        ds.render(0, 21, pu.SEED); a = ds.read()
        ds.set_option(_abi.OPT_BACKGROUND_BLOCKS, 0); ds.clear(); ds.render(0, 21, pu.SEED)
        assert (ds.read() == a).all() and np.isfinite(a).all() and (a > 0).all()
    assert_parity(away, 21)
    # cameras far from the origin (the fp32 rounding of the camera ray is part of the proof's margin; tests/test_background_blocks.py)
    from test_background_blocks import _far_scene
    for offset in (1e4, 2e6):
        assert_parity(_far_scene(offset, RenderConfig(256, 256, 8)), 8)
    assert_parity(sc, 24, rect=(0, 0, 160, 96))        # background only
    assert_parity(sc, 24, rect=(560, 40, 720, 200))    # the statue's head against the sky


def solstrale_background_blocks(sc):
    from solstrale_amd import background_blocks
    return background_blocks(sc, 0)


def test_device_tonemap_matches_host_arithmetic():
    import torch
    sc = scenes.create_test_scene(RenderConfig(200, 100, 25))
    with DeviceScene(sc) as ds:
        ds.render(0, 25, pu.SEED)
        sums = ds.read()
        t = torch.from_numpy(sums).cuda()
        rgb = ds.tonemap_rgb8(t.data_ptr(), 25)
    with np.errstate(invalid="ignore"):
        assert (rgb == im.sums_to_rgb8(sums, 25)).all()


# ---- the host mirror: ray_trace(scene, output, abort) ---------------------------------------------------------------------
def test_ray_trace_reports_progress_per_sample_and_final_image():
    sc = scenes.create_test_scene(RenderConfig(200, 100, 25))
    events, image = sc.ray_trace()
    assert len(events) == 25 and abs(events[-1][0] - 1.0) < 1e-12
    assert [e[3] for e in events].count(True) == 1 and events[-1][3]  # OnlyFinal: one image, in the last message
    expected = np.asarray(Image.open(os.path.join(GOLDEN, "out_expected_pathTracing.jpg")).convert("RGB"))
    assert im.compare_output(image, expected) > im.THRESHOLD
    sums = gpu_render(sc, 25)
    assert (image == im.sums_to_rgb8(sums, 25)).all()


def test_ray_trace_every_sample_and_abort():
    sc = scenes.cornell_box(RenderConfig(64, 64, 6))
    events, image = sc.ray_trace(strategy="every_sample")
    assert len(events) == 6 and all(e[3] for e in events)
    n = [0]

    def abort():
        n[0] += 1
        return n[0] > 2

    events, image = sc.ray_trace(strategy="every_sample", abort=abort)
    assert len(events) < 6  # aborted silently, Ok(()) like the reference (src/renderer/mod.rs:237-239)


def test_ray_trace_interval_strategy_and_growing_batches():
    """RenderImageStrategy::Interval (renderer/mod.rs:88-118): images now and then, always one with the last sample; the pass loop
    hands the device batches that grow (multiples of 16: the sums do not depend on the split), so the final picture is OnlyFinal's."""
    sc = scenes.cornell_box(RenderConfig(96, 64, 208))
    events, final_only = sc.ray_trace()
    assert len(events) == 208 and [e[3] for e in events].count(True) == 1
    events, image = sc.ray_trace(strategy="interval", interval_seconds=0.0)
    assert len(events) == 208 and events[-1][3] and [e[3] for e in events].count(True) >= 2
    assert (image == final_only).all()
    events, image = sc.ray_trace(strategy="interval", interval_seconds=3600.0)
    assert len(events) == 208 and [e[3] for e in events].count(True) >= 1 and events[-1][3]
    assert (image == final_only).all()
    sums = gpu_render(sc, 208)
    assert (image == im.sums_to_rgb8(sums, 208)).all()


# ---- full BASELINE sizes: size-independent properties ------------------------------------------------------------------
def test_full_size_c2_properties():
    """configs[1] at its full size (1920x1080, 256 spp, 10 000 spheres): additivity over sample ranges, partition
    independence on a strided subset, agreement of image statistics with the oracle's crop."""
    sc = scenes.cornell_spheres(RenderConfig(1920, 1080, 256))
    with DeviceScene(sc) as ds:
        ds.render(0, 256, pu.SEED)
        whole = ds.read()
        ds.clear()
        for f in range(0, 256, 64):
            ds.render(f, 64, pu.SEED)
        parts = ds.read()
        ds.clear()
        ds.set_partition(3, 8)
        ds.render(0, 256, pu.SEED)
        r3 = ds.read()
    assert np.isfinite(whole).all() and (whole >= 0).all()
    assert (whole == parts).all()
    from solstrale_amd import tiles
    owner, _ = tiles._slots(1920, 1080, 8)
    owned = owner == 3
    assert 0.12 < owned.mean() < 0.13 and (r3[owned] == whole[owned]).all() and not r3[~owned].any()
    ref, _ = orc.render(sc, 0, 256, pu.SEED, real=orc.ORC_F32, rect=(960, 540, 992, 572))
    assert pu.compare(whole, ref, 256, rect=(960, 540, 992, 572))["bad_pixels"] == 0


def test_full_size_c3_properties():
    """configs[2] at its full size (262 267 triangles, 1920x1080, 512 spp - the headline): additivity over sample ranges (row flip
    and sums as src/renderer/mod.rs:261-268,361-365), tile ownership of an 8-way partition, and a 512-spp oracle crop."""
    sc = scenes.sponza_like(RenderConfig(1920, 1080, 512))
    with DeviceScene(sc) as ds:
        ds.render(0, 512, pu.SEED)
        whole = ds.read()
        ds.clear()
        for f in range(0, 512, 128):
            ds.render(f, 128, pu.SEED)
        parts = ds.read()
        ds.clear()
        ds.set_partition(3, 8)
        ds.render(0, 512, pu.SEED)
        r3 = ds.read()
    assert np.isfinite(whole).all() and (whole >= 0).all()
    assert (whole == parts).all()
    from solstrale_amd import tiles
    owner, _ = tiles._slots(1920, 1080, 8)
    owned = owner == 3
    assert 0.12 < owned.mean() < 0.13 and (r3[owned] == whole[owned]).all() and not r3[~owned].any()
    rect = (1000, 600, 1024, 624)
    ref, _ = orc.render(sc, 0, 512, pu.SEED, real=orc.ORC_F32, rect=rect)
    res = pu.compare(whole, ref, 512, rect=rect)
    assert res["bad_pixels"] == 0 and res["max_rel"] <= pu.REL_TOL, res


def test_full_size_c5_properties():
    """configs[4] at its full size (statue stand-in, ~1.09 M triangles, Metal + Dielectric, 1920x1080 x 2048 spp = 128 chunks per
    pixel, the heavy-first work order of the long glass items): additivity over 4 x 512 samples (sums, not means:
    src/renderer/mod.rs:361-365), one rank's share of the 8-way job - rank 3 renders its tiles with all 2048 spp - against the whole
    frame on the pixels it owns, and a 2048-spp oracle crop on the glass head. Mirrors test_full_size_c3_properties."""
    sc = scenes.statue_like(RenderConfig(1920, 1080, 2048))
    assert abs(sc.desc.n_triangles - scenes.STATUE_TRIANGLES) < 2000
    with DeviceScene(sc) as ds:
        ds.render(0, 2048, pu.SEED)
        whole = ds.read()
        ds.clear()
        for f in range(0, 2048, 512):
            ds.render(f, 512, pu.SEED)
        parts = ds.read()
        ds.clear()
        ds.set_partition(3, 8)
        ds.render(0, 2048, pu.SEED)
        r3 = ds.read()
    assert np.isfinite(whole).all() and (whole >= 0).all()
    assert (whole == parts).all()
    from solstrale_amd import tiles
    owner, _ = tiles._slots(1920, 1080, 8)
    owned = owner == 3
    assert 0.12 < owned.mean() < 0.13 and (r3[owned] == whole[owned]).all() and not r3[~owned].any()
    rect = (940, 120, 956, 136)  # 16 x 16 pixels of the glass head (x 830-1020, y 55-255): long refraction paths
    # (the oracle adds the samples of ONE call up in a float, one after the other; over 2048 of them that association alone drifts
    # 1e-5 from the device's sums of 16-sample chunks - the same fp32 sample values either way - so it is asked for 8 x 256)
    ref, rays, samples = None, 0, 0
    for f in range(0, 2048, 256):
        ref, st = orc.render(sc, f, 256, pu.SEED, real=orc.ORC_F32, rect=rect, out=ref)
        rays, samples = rays + st["rays"], samples + st["samples"]
    st = {"rays": rays, "samples": samples}
    assert st["rays"] > 2.5 * st["samples"]  # (glass: enter, leave, go on - the crop lies where the long items are; the frame's mean is 1.77)
    res = pu.compare(whole, ref, 2048, rect=rect)
    assert res["bad_pixels"] == 0 and res["max_rel"] <= pu.REL_TOL, res


def test_c4_4k_crop_and_properties():
    """configs[3]: the Sponza-class scene at 3840x2160 x 1024 spp, tiles over 8 GPUs. On one GPU: two 128x128 oracle crops of the
    4K frame, additivity, and ONE RANK'S SHARE of the real job - rank 3 of 8 renders its tiles with all 1024 spp (64 chunks: the
    6.4 GB-class partial plane at 1/8 scale, blocks_x = 480, item counts near the 32-bit work counter's range when un-split)
    - checked against the sum of four 256-spp ranges and against a 1024-spp oracle crop on the pixels rank 3 owns."""
    sc = scenes.sponza_like(RenderConfig(3840, 2160, 1024))
    assert sc.desc.n_triangles == scenes.SPONZA_TRIANGLES
    from solstrale_amd import tiles
    owner, _ = tiles._slots(3840, 2160, 8)
    owned = owner == 3
    with DeviceScene(sc) as ds:
        assert ds.max_samples_per_call() >= 1024 // 8  # the whole frame needs split calls at 1024 spp; a rank's share does not
        ds.render(0, 16, pu.SEED)
        img16 = ds.read()
        ds.clear()
        ds.render(0, 4, pu.SEED)
        ds.render(4, 12, pu.SEED)  # not chunk aligned: the second call starts inside chunk 0
        img_split = ds.read()
        ds.clear()
        ds.set_partition(3, 8)
        ds.render(0, 1024, pu.SEED)
        r3 = ds.read()
        ds.clear()
        for f in range(0, 1024, 256):
            ds.render(f, 256, pu.SEED)
        r3_parts = ds.read()
    assert np.isfinite(img16).all() and (img16 >= 0).all() and np.isfinite(r3).all()
    for rect, spp in (((1800, 1000, 1928, 1128), 16), ((0, 2032, 128, 2160), 16)):
        ref, _ = orc.render(sc, 0, spp, pu.SEED, real=orc.ORC_F32, rect=rect)
        res = pu.compare(img16, ref, spp, rect=rect)
        assert res["bad_pixels"] == 0 and res["max_rel"] <= pu.REL_TOL, (rect, res)
    # sample ranges split off the chunk grid change the summation order (4 + 12 vs 16 in one chunk): equal within rounding
    assert np.allclose(img_split, img16, rtol=2e-6, atol=1e-6)
    assert (r3 == r3_parts).all() and not r3[~owned].any() and 0.12 < owned.mean() < 0.13
    rect = (2000, 1200, 2016, 1216)
    ref, _ = orc.render(sc, 0, 1024, pu.SEED, real=orc.ORC_F32, rect=rect)
    y, x = np.mgrid[rect[1]:rect[3], rect[0]:rect[2]]
    m = owned[y, x]
    assert m.any()
    g = r3[rect[1]:rect[3], rect[0]:rect[2]][m].astype(np.float64)
    r = ref[rect[1]:rect[3], rect[0]:rect[2]][m].astype(np.float64)
    assert (np.abs(g - r) <= pu.REL_TOL * np.abs(r) + pu.REL_TOL * 1024 * 1e-2).all()


def test_aux_albedo_and_normal_buffers():
    """renderer/mod.rs:175-204: at depth 0 the albedo shader's and the normal shader's colours of the primary hit are accumulated
    beside the pixel colour (background / zero on a miss). They equal the single-hit shaders' renders of the same samples."""
    spp = 6
    sc = scenes.create_test_scene(RenderConfig(120, 60, spp))  # background (.2, .3, .5), sky visible
    with DeviceScene(sc) as ds:
        ds.render(0, spp, pu.SEED)
        colour = ds.read()
        ds.render_aux(0, 2, pu.SEED)
        ds.render_aux(2, spp - 2, pu.SEED)  # additive over sample ranges like the colour accumulator
        albedo, normal = ds.read_aux()
        assert (ds.read() == colour).all()  # the colour accumulator is untouched
        ds.clear_aux()
        ds.render_aux(0, spp, pu.SEED)
        a2, n2 = ds.read_aux()
    want_a, _ = orc.render(scenes.create_test_scene(RenderConfig(120, 60, spp, AlbedoShader())), 0, spp, pu.SEED, real=orc.ORC_F32)
    sc_n = scenes.create_test_scene(RenderConfig(120, 60, spp, NormalShader()))
    for k in range(3):
        sc_n.desc.background[k] = 0.  # a miss adds ZERO_VECTOR to the normal buffer (mod.rs:203), not the background
    want_n, _ = orc.render(sc_n, 0, spp, pu.SEED, real=orc.ORC_F32)
    assert pu.compare(albedo, want_a, spp)["bad_pixels"] == 0
    assert np.abs(normal - want_n).max() <= 1e-5 * spp  # components of a normal are <= 1 in magnitude
    assert np.abs(a2 - albedo).max() <= 1e-5 * spp and np.abs(n2 - normal).max() <= 1e-5 * spp
    # pixels whose every sample missed: albedo = spp * background, normal = 0
    bg = np.array([.2, .3, .5], np.float32)
    sky = np.all(want_n == 0., axis=2) & np.all(np.abs(want_a - spp * bg) < 1e-6, axis=2)
    assert sky.sum() > 500 and (normal[sky] == 0.).all() and np.abs(albedo[sky] - spp * bg).max() < 1e-5
