"""Kernel time against samples per pixel (tail / fixed costs of a launch). Usage: python spp_curve.py [c3|c5] (not a pytest)"""
import _paths  # noqa: F401  (sys.path)
import sys

import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, scenes

which = sys.argv[1] if len(sys.argv) > 1 else "c5"
make = {"c3": scenes.sponza_like, "c5": scenes.statue_like, "c2": scenes.cornell_spheres, "test": scenes.create_test_scene}[which]
with DeviceScene(make(RenderConfig(1920, 1080, 16))) as ds:
    ds.kernel_timing(True)
    for spp in (16, 32, 64, 128, 256):
        ds.clear(); ds.render(0, spp, pu.SEED); ds.sync()
        ds.clear(); ds.render(0, spp, pu.SEED); ds.sync()
        ms, grid = ds.last_kernel_ms()
        print(f"{which} spp {spp:4d}: kernel {ms:8.2f} ms  {ms / spp:6.3f} ms/spp", flush=True)
