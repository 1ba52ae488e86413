"""Feature isolation on the GPU: small scenes each exercising one feature, compared with the fp32 oracle."""
import _paths  # noqa: F401  (sys.path)
import numpy as np

import parity_util as pu
import orc
from solstrale_amd import CameraConfig, DeviceScene, RenderConfig, SceneBuilder, Translation, scenes


def check(name, sc, spp=16):
    with DeviceScene(sc) as ds:
        ds.render(0, spp, pu.SEED)
        img = ds.read()
    ref, _ = orc.render(sc, 0, spp, pu.SEED, real=orc.ORC_F32)
    res = pu.compare(img, ref, spp)
    print(f"{name:28s} bad {res['bad_pixels']:6d}/{res['pixels']}  mean gpu {res['mean_gpu']:.6f} ref {res['mean_ref']:.6f}", flush=True)


def base(b, light="sphere", aperture=0.0, extra=()):
    cam = CameraConfig(20., aperture, (-5., 3., 6.), (.25, 1., 0.), (0., 1., 0.))
    ground = b.Lambertian(b.SolidColor(.5, .5, .5))
    lm = b.DiffuseLight(10., 10., 10.)
    world = [b.Quad((-5., 0., -15.), (20., 0., 0.), (0., 0., 20.), ground)]
    if light == "sphere":
        world.append(b.Sphere((10., 5., 10.), 10., lm))
    elif light == "quad":
        world.append(b.Quad((0., 0., 0.), (2., 0., 0.), (0., 0., 2.), lm, [scenes.RotationY(45.), Translation((-1., 10., -1.))]))
    elif light == "triangle":
        world.append(b.Triangle((-2., 1., -3.), (0., 1., -3.), (-1., 2., -3.), lm))
    world += list(extra)
    return b.finish(b.Bvh(world), cam, (.2, .3, .5), RenderConfig(200, 100, 16))


if __name__ == "__main__":
    for light in ("sphere", "quad", "triangle"):
        check(f"ground + {light} light", base(SceneBuilder(), light))
    check("aperture 0.1", base(SceneBuilder(), "quad", aperture=0.1))
    b = SceneBuilder()
    check("glass sphere", base(b, "quad", extra=[b.Sphere((-1., 1., 0.), 1., b.Dielectric(b.SolidColor(1., 1., 1.), None, 1.5))]))
    b = SceneBuilder()
    check("metal sphere fuzz .2", base(b, "quad", extra=[b.Sphere((-1., 1., 0.), 1., b.Metal(b.SolidColor(.8, .8, .8), None, .2))]))
    b = SceneBuilder()
    red = b.Lambertian(b.SolidColor(1., 0., 0.))
    med = b.ConstantMedium(b.Bvh(b.new_box((0., 0., -.5), (1., 2., .5), red, Translation((0., 0., 1.)))), 0.1, (1., 1., 1.))
    check("constant medium", base(b, "quad", extra=[med]))
    b = SceneBuilder()
    tex = b.Lambertian(b.ImageMap(scenes.load_image("textures/tex.jpg")))
    check("image texture quad", base(b, "quad", extra=[b.Quad((-2., .01, -2.), (4., 0., 0.), (0., 0., 4.), tex)]))
    check("reference test scene", scenes.create_test_scene(RenderConfig(200, 100, 16)))
