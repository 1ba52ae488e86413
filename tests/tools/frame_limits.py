"""The largest frames sol_scene_create accepts (not a pytest; MI355X box): Cornell at 16384 x 16384 and at 32768 x 32767 - one pixel row short of the
2^30 - 1 pixel limit: 16.7 M blocks, 3.2 G accumulator floats (12.9 GB), every index that is a uint32_t within 25 % of its end. One sample per pixel; crops
at the four corners and the centre against the float oracle, the whole frame finite, its mean against a 1/64-size render of the same view.
Usage: python tests/tools/frame_limits.py [WxH ...]"""
import _paths  # noqa: F401
import resource
import sys
import time

import numpy as np

import orc
import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, scenes

if __name__ == "__main__":
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(16384, 16384), (32768, 32767)]
    for w, h in sizes:
        sc = scenes.cornell_box(RenderConfig(w, h, 1))
        t0 = time.perf_counter()
        with DeviceScene(sc) as ds:
            print(f"{w}x{h} = {w * h / 1e6:.0f} M pixels: sol_scene_create {time.perf_counter() - t0:.1f} s {ds.build_times()} background blocks {ds.info()['background_blocks']}", flush=True)
            t0 = time.perf_counter()
            ds.render(0, 1, pu.SEED)
            ds.sync()
            t_render = time.perf_counter() - t0
            t0 = time.perf_counter()
            img = ds.read()
            t_read = time.perf_counter() - t0
        print(f"  render {t_render * 1e3:.0f} ms = {w * h / t_render / 1e6:.0f} Msamples/s, sol_read {t_read:.1f} s ({img.nbytes / 1e9:.1f} GB), host RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024} MB", flush=True)
        finite = bool(np.isfinite(img).all())
        mean = img.reshape(-1, 3)[:: 7].mean(axis=0, dtype=np.float64)
        bad_total = 0
        for rect in ((0, 0, 64, 64), (w - 64, 0, w, 64), (0, h - 64, 64, h), (w - 64, h - 64, w, h), (w // 2 - 32, h // 2 - 32, w // 2 + 32, h // 2 + 32), (w // 3, h - 200, w // 3 + 64, h - 136)):
            ref, _ = orc.render(sc, 0, 1, pu.SEED, real=orc.ORC_F32, rect=rect)
            res = pu.compare(img, ref, 1, rect)
            bad_total += res["bad_pixels"]
            print(f"  crop {rect}: {res['bad_pixels']} of {res['pixels']} outside 1e-5, max rel {res['max_rel']:.2e}, mean {res['mean_gpu']:.4f}", flush=True)
        small = scenes.cornell_box(RenderConfig(w // 8, h // 8, 64))
        with DeviceScene(small) as ds:
            ds.render(0, 64, pu.SEED)
            ref_mean = ds.read().reshape(-1, 3).mean(axis=0, dtype=np.float64) / 64
        print(f"  finite {finite}; frame mean (every 7th pixel) {np.round(mean, 4)} against {np.round(ref_mean, 4)} of the same view at 1/64 the pixels x 64 spp; bad crop pixels {bad_total}", flush=True)
        del img
