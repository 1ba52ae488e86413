"""One-off sweep: parity of seeded random scenes made of NEEDLE triangles (aspect 40:1 .. 2000:1, any orientation, textured or not, some of
them lights) against the fp32 oracle: the needle rule, the rotated fp32 records and the pre-splitting of the GPU tree builder together
(not a pytest). Usage: python needle_parity_sweep.py [first_seed] [count]"""
import _paths  # noqa: F401  (sys.path)
import sys

import numpy as np

import orc
import parity_util as pu
from solstrale_amd import CameraConfig, DeviceScene, PathTracingShader, RenderConfig, SceneBuilder, scenes


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


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    bad, strict, split = [], 0, 0
    for seed in range(first, first + count):
        sc = needle_scene(seed)
        with DeviceScene(sc) as ds:
            info = ds.info()
            strict += bool(info["strict_triangles"]); split += info["split_references"] > 0
            ds.render(0, 8, pu.SEED)
            img = ds.read()
        ref, _ = orc.render(sc, 0, 8, pu.SEED, real=orc.ORC_F32)
        res = pu.compare(img, ref, 8)
        if res["bad_pixels"] or not np.isfinite(img).all():
            bad.append((seed, int(res["bad_pixels"]), float(res["max_rel"])))
    print(f"{count} needle scenes from seed {first} ({strict} under the needle rule, {split} with pre-split references): {len(bad)} with pixels over 1e-5: {bad}", flush=True)
