"""Host-only structure checks of the device's world tree (sol_world_tree_check): the 7-wide quantised tree (64-byte nodes, implicit child addresses, permuted primitive arrays) that
sol_scene_create builds - from the reference's topology and from the binned-SAH rebuild - must hold every primitive reference of
the reference-shaped tree exactly once, and every child's DECODED box (the device's arithmetic) must contain all padded
primitive boxes below it. That is the whole correctness argument of the wide tree ("tree independence"); the GPU
suite then shows the images agree bit for bit."""
import numpy as np
import pytest

from solstrale_amd import CameraConfig, DeviceError, RenderConfig, SceneBuilder, scenes, world_tree_check

RC = RenderConfig(64, 64, 1)


def _chain(depth=48):
    b = SceneBuilder()
    m = b.Lambertian(b.SolidColor(.8, .8, .8))
    ids = [b.Sphere((float(x), 0.3 * (x % 3), 0.), 0.45, m) for x in range(depth + 2)]
    inner = b.Bvh(ids[:2])
    for k in range(2, depth + 2):
        inner = b.Bvh([inner, ids[k]]) if k % 2 else b.Bvh([ids[k], inner])
    light = b.Sphere((0., 1e4, 0.), 3e3, b.DiffuseLight(3, 3, 3))
    return b.finish(b.Bvh([inner, light]), CameraConfig(12., 0., (-30., 0.4, 0.3), (50., 0.3, 0.), (0, 1, 0)), (.1, .1, .1), RC)


def _flat_and_huge():
    """Degenerate inputs: coincident centroids, flat boxes, a huge primitive among tiny ones."""
    b = SceneBuilder()
    m = b.Lambertian(b.SolidColor(.5, .5, .5))
    world = [b.Quad((0., 0., 0.), (1., 0., 0.), (0., 1., 0.), m) for _ in range(9)]          # nine identical quads
    world += [b.Sphere((1e-3 * i, 0., 5.), 1e-4, m) for i in range(40)]                         # dust
    world.append(b.Sphere((0., 0., 0.), 5e3, m))                                                 # a huge sphere around all
    world.append(b.Quad((-1., 9., -1.), (2., 0., 0.), (0., 0., 2.), b.DiffuseLight(5, 5, 5)))
    return b.finish(b.Bvh(world), CameraConfig(40., 0., (0., 1., -8.), (0., 0., 0.), (0, 1, 0)), (0., 0., 0.), RC)


SCENES = {
    "cornell": lambda: scenes.cornell_box(RC),
    "spheres_2000": lambda: scenes.cornell_spheres(RC, n_spheres=2000),
    "atrium_30k": lambda: scenes.sponza_like(RC, n_triangles=30001, texture_size=16),
    "statue_40k": lambda: scenes.statue_like(RC, n_triangles=40000),
    "reference_test_scene": lambda: scenes.create_test_scene(RC),
    "spider_obj": lambda: scenes.create_obj_scene(RC),
    "chain_48": _chain,
    "degenerate": _flat_and_huge,
}


@pytest.mark.parametrize("use_sah", [0, 8, 16, 64], ids=["reference_topology", "sah8", "sah16", "sah64"])
@pytest.mark.parametrize("name", list(SCENES))
def test_wide_tree_is_sound(name, use_sah):
    sc = SCENES[name]()
    r = world_tree_check(sc, use_sah)
    assert r["box_violations"] == 0 and r["leaf_mismatches"] == 0 and r["bad_empty_slots"] == 0, r
    assert r["n_leaf_refs"] == r["n_primitives"] >= 2
    assert 1 <= r["max_children"] <= 7 and r["depth"] >= 1
    # a collapsed tree needs far fewer nodes than the n - 1 of the binary tree (at most ~n/2 even for a chain)
    assert r["n_wide"] <= max(1, (r["n_primitives"] + 1) // 2)
    assert np.isfinite(r["inner_area"]) and np.isfinite(r["leaf_area"])


def test_sah_rebalances_a_chain_and_keeps_good_trees_good():
    ref, sah = world_tree_check(_chain(), False), world_tree_check(_chain(), True)
    assert ref["depth"] >= 7 and sah["depth"] <= 4  # 50 spheres: a chain collapses to >= 49/7 levels, SAH to ~log8
    for name in ("atrium_30k", "statue_40k"):
        sc = SCENES[name]()
        ref, sah = world_tree_check(sc, False), world_tree_check(sc, True)
        cost = lambda r: 2.5 * r["inner_area"] + r["leaf_area"]
        assert cost(sah) < 1.1 * cost(ref), (name, cost(ref), cost(sah))


def test_single_primitive_world_has_no_tree():
    b = SceneBuilder()
    light = b.Sphere((0., 0., 0.), 1., b.DiffuseLight(1, 1, 1))
    sc = b.finish(light, CameraConfig(40., 0., (0., 0., 5.), (0., 0., 0.), (0, 1, 0)), (0., 0., 0.), RC)
    with pytest.raises(DeviceError):
        world_tree_check(sc, True)


@pytest.mark.parametrize("seed", range(40))
def test_random_scenes_build_sound_trees(seed):
    import random_scenes
    sc = random_scenes.random_scene(seed)
    for use_sah in (0, 8, 16, 64):
        r = world_tree_check(sc, use_sah)
        assert r["box_violations"] == 0 and r["leaf_mismatches"] == 0 and r["bad_empty_slots"] == 0, (seed, use_sah, r)
        assert r["n_leaf_refs"] == r["n_primitives"]


_hetero_cache = {}


def _hetero(camera):  # (4 s of numpy per build: shared by the parametrisations)
    if camera not in _hetero_cache:
        _hetero_cache[camera] = scenes.sponza_like(RC, mesh="heterogeneous", texture_size=16, camera=camera)
    return _hetero_cache[camera]


DEVICE_SCENES = dict(SCENES)
DEVICE_SCENES["atrium_heterogeneous"] = lambda: _hetero("default")
DEVICE_SCENES["atrium_heterogeneous_interior"] = lambda: _hetero("interior")


@pytest.mark.gpu
@pytest.mark.parametrize("split", [None, "0", "100"], ids=["default_split", "no_split", "split_100"])
@pytest.mark.parametrize("name", list(DEVICE_SCENES))
def test_device_built_tree_is_sound(name, split, monkeypatch):
    """The tree sol_build.hip builds ON THE GPU (SolCreateOptions.world_tree = SOL_TREE_DEVICE: triangle pre-splitting, Morton sort,
    PLOC clustering, surface-area collapse, per-level emission) under the same structural check: every primitive reference once -
    a pre-split triangle once per part -, every decoded child box containing the padded box of every reference below it, the boxes
    of a split triangle's references covering the triangle between them, valid permutations of the primitive arrays."""
    if split is not None:
        monkeypatch.setenv("SOL_SPLIT", split)
        monkeypatch.setenv("SOL_SPLIT_SLACK", "0" if split == "100" else "3")
    sc = DEVICE_SCENES[name]()
    r = world_tree_check(sc, -1)
    assert r["box_violations"] == 0 and r["leaf_mismatches"] == 0 and r["bad_empty_slots"] == 0 and r["split_uncovered"] == 0, r
    assert r["n_leaf_refs"] == r["n_primitives"] + r["n_extra_references"] and r["n_primitives"] >= 2
    assert r["n_extra_references"] <= (r["n_primitives"] if split == "100" else 0.31 * r["n_primitives"])
    assert (r["n_extra_references"] == 0) == (r["n_split_triangles"] == 0)
    if split == "0":
        assert r["n_extra_references"] == 0
    if name.startswith("atrium_heterogeneous") and split != "0":
        assert r["n_split_triangles"] > 500, r  # the walls, rails and rods are what pre-splitting is for
    assert 1 <= r["max_children"] <= 7 and r["depth"] >= 1
    assert r["n_wide"] <= max(1, (r["n_primitives"] + 1) // 2)
