"""Times ray_trace() of the reference's profiling workload through the host mirror, and the bare sol_render of the same job (not a pytest)."""
import _paths  # noqa: F401
import time

import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, scenes

sc = scenes.create_test_scene(RenderConfig(800, 400, 1000))
for k in range(3):
    t = time.perf_counter()
    ev, img = sc.ray_trace(strategy="only_final")
    print(f"ray_trace #{k}: {time.perf_counter() - t:.4f} s, {len(ev)} events", flush=True)
with DeviceScene(sc) as ds:
    for k in range(3):
        ds.clear()
        t = time.perf_counter()
        ds.render(0, 1000, pu.SEED)
        ds.sync()
        print(f"sol_render 1000 spp #{k}: {time.perf_counter() - t:.4f} s", flush=True)
    for n in (16, 32, 64, 128, 256):
        ds.clear()
        t = time.perf_counter()
        ds.render(0, n, pu.SEED)
        ds.sync()
        print(f"sol_render {n} spp: {time.perf_counter() - t:.4f} s", flush=True)
