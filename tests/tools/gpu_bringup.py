"""GPU bring-up script (not a pytest): renders a few scenes on cuda:0 through the C ABI, compares with the fp32 oracle,
prints throughput. Usage: python tests/gpu_bringup.py [scene ...]"""
import _paths  # noqa: F401  (sys.path)
import sys
import time

import numpy as np

import parity_util as pu
import orc
from solstrale_amd import DeviceScene, RenderConfig, device_count, scenes


def run(name, sc, spp, rect=None, reps=1):
    t = time.time()
    ds = DeviceScene(sc)
    t_up = time.time() - t
    ds.render(0, spp, pu.SEED)
    ds.sync()
    t = time.time()
    for _ in range(reps):
        ds.clear()
        ds.render(0, spp, pu.SEED)
        ds.sync()
    dt = (time.time() - t) / reps
    img = ds.read()
    ds.clear()
    ds.render(0, spp, pu.SEED, counted=True)
    st = ds.stats()
    t = time.time()
    ref, ost = orc.render(sc, 0, spp, pu.SEED, real=orc.ORC_F32, rect=rect)
    t_or = time.time() - t
    res = pu.compare(img, ref, spp, rect)
    ns = sc.width * sc.height * spp
    print(f"{name}: {sc.width}x{sc.height}x{spp} upload {t_up:.2f}s gpu {dt*1e3:.1f} ms = {ns/dt/1e6:.1f} Msamples/s, "
          f"{st['rays']/dt/1e6:.1f} Mrays/s; oracle f32 {t_or:.1f}s", flush=True)
    print("   stats", st, flush=True)
    print("   parity", res, flush=True)
    ds.close()
    return res


if __name__ == "__main__":
    print("devices", device_count(), flush=True)
    which = sys.argv[1:] or ["cornell", "cornell_big", "spheres", "sponza_small", "test_scene"]
    if "cornell" in which:
        run("cornell", scenes.cornell_box(RenderConfig(400, 400, 50)), 50)
    if "cornell_big" in which:
        run("cornell_1080p", scenes.cornell_box(RenderConfig(1920, 1080, 64)), 64, rect=(900, 500, 1028, 628), reps=3)
    if "spheres" in which:
        run("cornell+10k spheres", scenes.cornell_spheres(RenderConfig(1920, 1080, 16)), 16, rect=(900, 500, 1028, 628), reps=3)
    if "sponza_small" in which:
        run("sponza-like 20k", scenes.sponza_like(RenderConfig(640, 360, 16), n_triangles=20001, texture_size=256), 16,
            rect=(256, 128, 384, 256))
    if "sponza" in which:
        run("sponza-like 262k", scenes.sponza_like(RenderConfig(1920, 1080, 16)), 16, rect=(900, 500, 1028, 628), reps=3)
    if "test_scene" in which:
        run("reference test scene", scenes.create_test_scene(RenderConfig(200, 100, 25)), 25)
