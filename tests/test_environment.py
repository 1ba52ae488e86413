"""The environment-map EXTENSION (SolSceneDesc::env_*, abi_version 2; not in the reference, which has only a constant
background colour, src/renderer/mod.rs:197-204): the direction -> texel mapping of the oracle against an independent numpy
reading of its definition, version-1 descriptions still accepted, and - on the GPU - parity of the device with the oracle."""
import numpy as np
import pytest

import orc
import parity_util as pu
from solstrale_amd import AlbedoShader, CameraConfig, DeviceScene, PathTracingShader, RenderConfig, SceneBuilder, _abi, scenes


def _sky_scene(look_at, env, w=33, h=33, fov=4.0):
    """Nothing in view: every pixel is a miss, so the AlbedoShader frame is the environment seen through the camera."""
    b = SceneBuilder()
    light = b.Sphere((0., -1e5, 0.), 1., b.DiffuseLight(1, 1, 1))  # far below, out of view (a scene needs a light)
    b.environment(env, 2.0)
    cam = CameraConfig(fov, 0., (0., 0., 0.), look_at, (0., 1., 0.) if abs(look_at[1]) < 0.9 else (1., 0., 0.))
    return b.finish(b.Bvh([light, light]), cam, (9., 9., 9.), RenderConfig(w, h, 1, AlbedoShader()))


def _index_map(w, h):
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([x / w, y / h, np.ones_like(x, float)], -1).astype(np.float32)


@pytest.mark.parametrize("look,u_expect,row_expect", [((1., 0., 0.), 0.5, 0.5), ((0., 0., 1.), 0.25, 0.5), ((0., 0., -1.), 0.75, 0.5),
                                                       ((0., 1., 0.), None, 0.0), ((0., -1., 0.), None, 1.0), ((1., 1., 0.), 0.5, 0.25)])
def test_oracle_direction_to_texel(look, u_expect, row_expect):
    """phi = -atan2(z, x) + pi, u = phi / 2pi; theta = acos(-y), v = theta / pi, row = (1 - v)(H - 1): +x -> u 0.5, +z -> 0.25,
    -z -> 0.75, up -> row 0, down -> last row (calculate_sphere_uv, sphere.rs:134-140, applied to the ray direction)."""
    W, H = 64, 32
    sc = _sky_scene(look, _index_map(W, H))
    img, _ = orc.render(sc, 0, 1, pu.SEED, real=orc.ORC_F64)
    c = img[16, 16] / 2.0  # env_scale 2
    assert abs(c[2] - 1.0) < 1e-6
    if u_expect is not None:
        assert abs(c[0] - u_expect) < 1.5 / W, c
    assert abs(c[1] - row_expect * (H - 1) / H) < 1.5 / H, c
    f32, _ = orc.render(sc, 0, 1, pu.SEED, real=orc.ORC_F32)
    assert np.abs(f32 - img).max() < 2.0 * 1.5 / min(W, H)  # fp32 picks the same or a neighbouring texel


def test_version_1_descriptions_do_not_read_the_environment_fields():
    sc = _sky_scene((1., 0., 0.), _index_map(8, 4))
    with_env, _ = orc.render(sc, 0, 1, pu.SEED, real=orc.ORC_F64)
    sc.desc.abi_version = 1
    without, _ = orc.render(sc, 0, 1, pu.SEED, real=orc.ORC_F64)
    sc.desc.abi_version = _abi.SOL_ABI_VERSION
    assert np.allclose(without, 9.0) and not np.allclose(with_env, 9.0)  # version 1: the constant background colour


@pytest.mark.gpu
def test_environment_parity_with_the_oracle():
    sc = scenes.create_test_scene_with_environment(RenderConfig(200, 100, 16, PathTracingShader(50)))
    with DeviceScene(sc) as ds:
        ds.render(0, 16, pu.SEED)
        img = ds.read()
    ref, _ = orc.render(sc, 0, 16, pu.SEED, real=orc.ORC_F32)
    res = pu.compare(img, ref, 16)
    assert res["bad_pixels"] == 0 and res["max_rel"] <= pu.REL_TOL, res  # (tests/tools/env_probe.py prints the paths of a pixel that differs)
    plain = scenes.create_test_scene(RenderConfig(200, 100, 16, PathTracingShader(50)))
    with DeviceScene(plain) as ds:
        ds.render(0, 16, pu.SEED)
        assert not np.allclose(ds.read(), img)  # the environment is really used


@pytest.mark.gpu
def test_environment_sky_frames_are_exact():
    """Pure-miss frames: the device's lookup against the fp32 oracle's, texel for texel, in six directions."""
    env = _index_map(64, 32)
    for look in ((1., 0., 0.), (0., 0., 1.), (-1., 0., 0.3), (0., 1., 0.), (0., -1., 0.), (1., 1., -1.)):
        sc = _sky_scene(look, env, 64, 48, 60.0)
        with DeviceScene(sc) as ds:
            ds.render(0, 1, pu.SEED)
            img = ds.read()
        ref, _ = orc.render(sc, 0, 1, pu.SEED, real=orc.ORC_F32)
        assert (img == ref.astype(np.float32)).all(), look
