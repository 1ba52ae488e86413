"""Times the device bloom post-processor on a 1080p and a 4K frame (not a pytest). Usage: python bloom_bench.py"""
import _paths  # noqa: F401  (sys.path)
import time

import torch

from solstrale_amd import DeviceScene, RenderConfig, scenes

for w, h in ((1920, 1080), (3840, 2160)):
    with DeviceScene(scenes.cornell_box(RenderConfig(w, h, 1))) as ds:
        img = torch.rand(h, w, 3, device="cuda") ** 6 * 40.
        for frac in (0.02, 0.1, 0.2):
            k = int(frac * w) * 2 + 1
            ds.bloom(img.data_ptr(), 1, frac)
            ds.sync()
            t = time.perf_counter()
            for _ in range(3):
                ds.bloom(img.data_ptr(), 1, frac)
            ds.sync()
            dt = (time.perf_counter() - t) / 3
            taps = 2 * k * w * h * 3  # f64 multiply-adds of the two blur passes
            print(f"{w}x{h} kernel_size {k:4d}: {dt * 1e3:8.2f} ms  {taps / dt / 1e12:6.2f} T f64 FMA/s  "
                  f"{2 * k * w * h * 24 / dt / 1e12:6.2f} TB/s of tap reads (LDS-tiled blur: served by LDS)", flush=True)
