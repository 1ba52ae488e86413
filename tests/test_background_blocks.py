"""Background blocks (include/solstrale_hip.h, SolSceneInfo::background_blocks): the 8x8 pixel blocks sol_scene_create proves to see
nothing but the constant background, whose samples sol_render sums without tracing them. The proof runs on the host
(sol_background_blocks, no device needed) and is checked here against the oracle, which knows nothing of it: every sample of every
pixel of a flagged block must be the background colour, in the float AND the double instantiation."""
import numpy as np
import pytest

import orc
import parity_util as pu
from solstrale_amd import CameraConfig, PathTracingShader, RenderConfig, SceneBuilder, background_blocks, scenes


def _pixel_mask(flags, sc):
    return np.repeat(np.repeat(flags, 8, axis=0), 8, axis=1)[:sc.height, :sc.width]


def _background_sum(sc, spp):
    bg = np.array(sc.desc.background[:3], dtype=np.float32)
    want = np.zeros(3, dtype=np.float32)
    for _ in range(spp):
        want = want + bg  # the additions sol_fill_background_kernel makes
    return want


@pytest.mark.parametrize("name,make,least", [
    ("c5", lambda: scenes.statue_like(RenderConfig(483, 271, 6), n_triangles=20000), 0.35),    # ragged size: edge blocks
    ("c3", lambda: scenes.sponza_like(RenderConfig(480, 270, 6), n_triangles=20000), 0.01),
    ("c3_heterogeneous", lambda: scenes.sponza_like(RenderConfig(240, 136, 6), mesh="heterogeneous"), 0.0),
    ("bvh_bench", lambda: scenes.new_bvh_test_scene(RenderConfig(200, 100, 6, PathTracingShader(8)), True, 300), 0.0),
    # thin-lens cameras: every pixel's rays leave a disc, not a point
    ("test_scene_aperture_0.1", lambda: scenes.create_test_scene(RenderConfig(400, 200, 6, PathTracingShader(8))), 0.0),  # (its sky lies inside the big light sphere's box)
    ("obj_scene_aperture_20", lambda: scenes.create_obj_scene(RenderConfig(400, 200, 6, PathTracingShader(8))), 0.05),
    ("wide_lens", lambda: _wide_lens_scene(RenderConfig(256, 128, 6, PathTracingShader(4))), 0.02),
])
def test_flagged_blocks_are_background_in_the_oracle(name, make, least):
    sc = make()
    spp = 6
    union = np.zeros(((sc.height + 7) // 8, (sc.width + 7) // 8), dtype=bool)
    for tree in (0, 16):  # the reference's topology and the 16-bin SAH rebuild: any tree may carry the proof
        union |= background_blocks(sc, tree)
    assert union.mean() >= least, (name, union.mean())
    m = _pixel_mask(union, sc)
    want = _background_sum(sc, spp)
    for real in (orc.ORC_F32, orc.ORC_F64):
        img, _ = orc.render(sc, 0, spp, pu.SEED, real=real)
        same = (img[m] == want) if real == orc.ORC_F32 else np.isclose(img[m], want, rtol=1e-6, atol=0)  # (f64 adds in double)
        assert same.all(), (name, real, int((~same).any(axis=-1).sum()))
    # and the proof is not vacuous where it matters: most of the pixels that ARE pure background sit in flagged blocks
    if least > 0.3:
        img, _ = orc.render(sc, 0, spp, pu.SEED, real=orc.ORC_F32)
        pure = (img == want).all(axis=-1)
        assert m.sum() > 0.75 * pure.sum(), (int(m.sum()), int(pure.sum()))


def _wide_lens_scene(rc):
    # a lens as wide as the objects, focused 3 units in front of them: the blur circles are dozens of pixels wide
    b = SceneBuilder()
    cam = CameraConfig(35., 1.2, (0., 1., 8.), (0., 1., 3.), (0., 1., 0.))
    light = b.DiffuseLight(6., 6., 6.)
    grey = b.Lambertian(b.SolidColor(.6, .6, .6))
    world = [b.Sphere((-1.5, 1., 0.), .6, light), b.Sphere((1.2, 1.3, -2.), .5, grey), b.Triangle((-.5, 0., 1.), (.5, 0., 1.), (0., .9, 1.), grey)]
    return b.finish(b.Bvh(world), cam, (.2, .3, .5), rc)


def test_no_blocks_where_the_proof_does_not_apply():
    # an environment map (the background is looked up per ray), a camera inside the geometry
    assert not background_blocks(scenes.create_test_scene_with_environment(RenderConfig(200, 100, 1)), 0).any()
    assert not background_blocks(scenes.cornell_box(RenderConfig(200, 200, 1)), 0).any()                  # closed box


def test_a_silhouette_block_is_traced():
    # one small sphere in front of a pinhole camera: the blocks its silhouette touches (and their neighbours, by the one-pixel
    # margin) are not flagged, the far corners of the image are
    b = SceneBuilder()
    cam = CameraConfig(40., 0., (0., 0., 5.), (0., 0., 0.), (0., 1., 0.))
    light = b.DiffuseLight(5., 5., 5.)
    world = [b.Sphere((0., 0., 0.), .5, light), b.Sphere((0.2, 0.1, -1.), .3, b.Lambertian(b.SolidColor(.5, .5, .5)))]
    sc = b.finish(b.Bvh(world), cam, (.2, .3, .5), RenderConfig(256, 256, 8))
    f = background_blocks(sc, 0)
    assert f[0, 0] and f[-1, -1] and f[0, -1] and f[-1, 0] and not f[16, 16] and not f[15, 15]
    img, _ = orc.render(sc, 0, 8, pu.SEED, real=orc.ORC_F32)
    want = _background_sum(sc, 8)
    touched = ~(img == want).all(axis=-1)
    assert not (touched & _pixel_mask(f, sc)).any()
    # conservative but not useless: every block more than two blocks away from a touched pixel is flagged
    tb = np.zeros_like(f)
    ys, xs = np.nonzero(touched)
    tb[ys // 8, xs // 8] = True
    near = np.zeros_like(f)
    for dy in range(-2, 3):
        for dx in range(-2, 3):
            near |= np.roll(np.roll(tb, dy, axis=0), dx, axis=1)
    assert f[~near].all()


def _far_scene(offset, rc):
    b = SceneBuilder()
    cam = CameraConfig(2., 0., (offset, 0., 500.), (offset, 0., 0.), (0., 1., 0.))
    light = b.DiffuseLight(5., 5., 5.)
    world = [b.Sphere((offset, 0., 0.), 3., light), b.Sphere((offset + 2., 1., -20.), 2.5, b.Lambertian(b.SolidColor(.5, .5, .5)))]
    return b.finish(b.Bvh(world), cam, (.2, .3, .5), rc)


def test_a_camera_far_from_the_origin_widens_the_margin_or_gives_up():
    """generate_path forms the ray in fp32: far from the origin its direction errs by more than the pixel the proof allows for. The margin
    grows with that bound (10 000 units away: a tenth of a pixel) and beyond three pixels no block is flagged (2 000 000 away, 2 degrees
    of view: the fp32 rays are ~30 pixels off - the float oracle, like the device, renders what those rays see)."""
    rc = RenderConfig(256, 256, 6)
    near = _far_scene(1e4, rc)
    f = background_blocks(near, 0)
    assert f.any() and not f[16, 16]
    img, _ = orc.render(near, 0, 6, pu.SEED, real=orc.ORC_F32)
    assert (img[_pixel_mask(f, near)] == _background_sum(near, 6)).all()
    assert not background_blocks(_far_scene(2e6, rc), 0).any()


@pytest.mark.parametrize("first", [0, 40, 80])
def test_random_scenes_flagged_blocks_are_background(first):
    """The proof over seeded random scenes (tests/random_scenes.py: every primitive and transformation, media, 40 % lens cameras; needle
    scenes on the odd seeds): every sample of every pixel of a flagged block is the background in the float oracle. (A sweep of seeds
    0-8999 at this size - 5 690 scenes with flagged blocks, 196 458 of them - found no other pixel.)"""
    import random_scenes
    flagged = 0
    for seed in range(first, first + 40):
        sc = random_scenes.random_scene(seed, width=96, height=72, spp=4) if seed % 2 == 0 else random_scenes.needle_scene(seed, width=96, height=72, spp=4)
        f = background_blocks(sc, 0) | background_blocks(sc, 16)
        flagged += int(f.sum())
        if f.any():
            img, _ = orc.render(sc, 0, 4, pu.SEED, real=orc.ORC_F32)
            assert (img[_pixel_mask(f, sc)] == _background_sum(sc, 4)).all(), seed
    assert flagged > 200
