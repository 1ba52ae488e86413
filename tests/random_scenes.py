"""Seeded random small scenes over every primitive, material, texture, transformation, nesting and light kind the path
supports - for randomised parity tests (GPU vs fp32 oracle) and for structure checks of the device's world tree."""
import numpy as np

from solstrale_amd import (CameraConfig, PathTracingShader, RenderConfig, RotationX, RotationY, RotationZ, Scale, SceneBuilder,
                           Translation, scenes)


def random_scene(seed, width=40, height=32, spp=4, max_depth=12, far=1.0):
    """far > 1: the same scene seen through a long lens from `far` times the distance (what BASELINE config 2 does to its Cornell box from
    800 units away: the hit parameters are large, the digits fp32 has left for them few)."""
    rng = np.random.default_rng(seed)
    b = SceneBuilder()
    u = lambda lo, hi: float(rng.uniform(lo, hi))
    v3 = lambda lo, hi: tuple(float(x) for x in rng.uniform(lo, hi, 3))

    def texture():
        if rng.random() < 0.3:
            n = int(rng.integers(2, 9))
            return b.ImageMap(rng.integers(0, 256, (n, n + 1, 3), dtype=np.uint8))
        return b.SolidColor(*v3(0.05, 0.95))

    def normal_map():
        if rng.random() < 0.25:
            n = int(rng.integers(2, 6))
            img = np.clip(np.array([128, 128, 255]) + rng.integers(-40, 40, (n, n, 3)), 0, 255).astype(np.uint8)
            return b.load_normal_texture(img)
        return None

    def material(depth=0):
        k = rng.integers(0, 5 if depth < 2 else 4)
        if k == 0 or k == 3:
            return b.Lambertian(texture(), normal_map())
        if k == 1:
            return b.Metal(texture(), normal_map(), u(0., 0.6))
        if k == 2:
            return b.Dielectric(texture(), None, u(1.1, 2.2))
        return b.Blend(material(depth + 1), material(depth + 1), u(0.1, 0.9))

    def transform():
        ops = []
        for _ in range(int(rng.integers(0, 3))):
            k = rng.integers(0, 5)
            ops.append([Translation(v3(-1., 1.)), RotationX(u(-60., 60.)), RotationY(u(-60., 60.)), RotationZ(u(-60., 60.)),
                        Scale(u(0.6, 1.5))][k])
        return ops or None

    def primitive(mat):
        k = rng.integers(0, 4)
        c = np.array(v3(-3., 3.))
        if k == 0:
            return [b.Sphere(tuple(c), u(0.2, 0.9), mat)]
        if k == 1:
            return [b.Quad(tuple(c), v3(-1.5, 1.5), v3(-1.5, 1.5), mat, transform())]
        if k == 2:
            uv = tuple((u(-1., 2.), u(-1., 2.)) for _ in range(3)) if rng.random() < 0.5 else None
            return [b.Triangle(tuple(c), tuple(c + v3(-1.5, 1.5)), tuple(c + v3(-1.5, 1.5)), mat, transform(), uv=uv)]
        return b.new_box(tuple(c), tuple(c + np.abs(v3(0.3, 1.2))), mat, transform())

    world = []
    for _ in range(int(rng.integers(3, 14))):
        prims = primitive(material())
        if rng.random() < 0.2 and len(prims) > 1:
            world.append(b.Bvh(prims))  # a nested Bvh (inlined as a node, tests/scenes.rs:331-334)
        else:
            world += prims
    if rng.random() < 0.4:  # a constant medium around a box or a sphere
        boundary = b.Bvh(b.new_box(v3(-2., 0.), v3(0.5, 2.), b.Lambertian(b.SolidColor(1., 1., 1.)))) if rng.random() < 0.6 \
            else b.Sphere(v3(-1., 1.), u(0.5, 1.2), b.Lambertian(b.SolidColor(1., 1., 1.)))
        world.append(b.ConstantMedium(boundary, u(0.05, 1.5), v3(0.2, 1.)))
    for _ in range(int(rng.integers(1, 4))):  # lights of every shape, with and without attenuation
        lm = b.DiffuseLight(*v3(2., 12.), None if rng.random() < 0.6 else u(0.2, 3.))
        k = rng.integers(0, 3)
        c = np.array(v3(-4., 4.)) + np.array([0., 5., 0.])
        if k == 0:
            world.append(b.Sphere(tuple(c), u(0.3, 1.5), lm))
        elif k == 1:
            world.append(b.Quad(tuple(c), v3(-2., 2.), v3(-2., 2.), lm))
        else:
            world.append(b.Triangle(tuple(c), tuple(c + v3(-2., 2.)), tuple(c + v3(-2., 2.)), lm))
    order = rng.permutation(len(world))
    world = [world[i] for i in order]
    fov, aperture, look_from, look_at = u(25., 70.), 0. if rng.random() < 0.6 else u(0.02, 0.3), v3(-6., 6.)[:2] + (u(5., 9.),), v3(-1., 1.)
    if far != 1.0:
        look_from = tuple(np.asarray(look_at) + (np.asarray(look_from) - np.asarray(look_at)) * far)
        fov = float(np.degrees(2. * np.arctan(np.tan(np.radians(fov) / 2.) / far)))
    cam = CameraConfig(fov, aperture, look_from, look_at, (0., 1., 0.))
    rc = RenderConfig(width, height, spp, PathTracingShader(max_depth))
    return b.finish(b.Bvh(world), cam, v3(0., 0.6), rc)


def needle_scene(seed, width=64, height=48, spp=8):
    rng = np.random.default_rng(seed)
    b = SceneBuilder()
    tex = b.Lambertian(b.ImageMap(scenes.load_image("textures/checker.jpg")))
    mats = [b.Lambertian(b.SolidColor(*rng.uniform(.2, .9, 3))), tex, b.Metal(b.SolidColor(.8, .8, .8), None, float(rng.uniform(0., .3))), b.Dielectric(b.SolidColor(1., 1., 1.), None, 1.5)]
    light = b.DiffuseLight(8., 8., 8.)
    world = [b.Sphere((0., 6., 2.), 1.5, light), b.Quad((-8., -2., -8.), (16., 0., 0.), (0., 0., 16.), mats[0])]
    for k in range(int(rng.integers(20, 60))):
        c = rng.uniform(-3., 3., 3)
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        length = float(rng.uniform(1., 8.))
        aspect = float(np.exp(rng.uniform(np.log(40.), np.log(2000.))))
        w = np.cross(d, rng.normal(size=3)); w = w / np.linalg.norm(w) * (length / aspect)
        p0, p1 = c - d * length / 2, c + d * length / 2
        order = int(rng.integers(0, 3))  # which vertex is listed first: every rotation of the record occurs
        verts = [tuple(p0), tuple(p1), tuple(p1 + w)]
        verts = verts[order:] + verts[:order]
        m = light if rng.random() < 0.05 else mats[int(rng.integers(0, len(mats)))]
        uv = tuple((float(rng.uniform(-1., 2.)), float(rng.uniform(-1., 2.))) for _ in range(3)) if rng.random() < 0.5 else None
        world.append(b.Triangle(verts[0], verts[1], verts[2], m, None, uv=uv) if uv else b.Triangle(verts[0], verts[1], verts[2], m))
    cam = CameraConfig(float(rng.uniform(30., 60.)), 0. if rng.random() < 0.7 else float(rng.uniform(0.02, 0.2)),
                       tuple(rng.uniform(-5., 5., 2)) + (float(rng.uniform(6., 10.)),), tuple(rng.uniform(-1., 1., 3)), (0., 1., 0.))
    return b.finish(b.Bvh(world), cam, (.2, .3, .5), RenderConfig(width, height, spp, PathTracingShader(12)))
