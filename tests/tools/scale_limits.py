"""Beyond BASELINE's sizes (not a pytest; MI355X box): the statue stand-in at 4 M triangles (4x config 5: the device tree build, 24-bit base indices of the
wide nodes at a quarter of their range) and at 9 M (more primitive references than the device build takes - 2^23 - so SOL_TREE_AUTO must fall back to the
host-built tree, and say so). For each: host scene build, sol_scene_create (tree name / fallback note, seconds), a 1080p x 8 spp render, a 96x96 crop against
the float oracle (which walks the reference tree of the same scene), device memory. Usage: python tests/tools/scale_limits.py [n_triangles ...]"""
import _paths  # noqa: F401
import resource
import sys
import time

import numpy as np

import orc
import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, scenes

if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [4_000_000, 9_000_000]
    for n in sizes:
        t0 = time.perf_counter()
        sc = scenes.statue_like(RenderConfig(1920, 1080, 8), n_triangles=n)
        t_host = time.perf_counter() - t0
        print(f"{sc.desc.n_triangles} triangles: host scene (mesh, reference Bvh::new, flatten) {t_host:.1f} s, reference tree depth {sc.tree_depth}, "
              f"host RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024} MB", flush=True)
        t0 = time.perf_counter()
        with DeviceScene(sc) as ds:
            t_create = time.perf_counter() - t0
            info = ds.info()
            print(f"  sol_scene_create {t_create:.2f} s {ds.build_times()} tree '{info['tree_name']}' fallback {info['tree_fallback']} note '{info['tree_note']}' "
                  f"stack bound {info['stack_bound']} split references {info['split_references']}", flush=True)
            ds.render(0, 8, pu.SEED)
            ds.sync()
            best = 1e9
            for _ in range(2):
                ds.clear()
                t = time.perf_counter()
                ds.render(0, 8, pu.SEED)
                ds.sync()
                best = min(best, time.perf_counter() - t)
            img = ds.read()
            st = None
            ds.clear()
            ds.render(0, 8, pu.SEED, counted=True)
            st = ds.stats()
            ds.clear()
            ds.render(0, 8, pu.SEED)
            img2 = ds.read()
        print(f"  1080p x 8 spp: {best * 1e3:.1f} ms = {1920 * 1080 * 8 / best / 1e6:.0f} Msamples/s; node visits / ray {st['node_visits'] / st['rays']:.2f}, "
              f"triangle tests / ray {st['triangle_tests'] / st['rays']:.2f}; frames of two renders identical: {bool((img == img2).all())}", flush=True)
        rect = (912, 400, 1008, 496)
        t0 = time.perf_counter()
        ref, _ = orc.render(sc, 0, 8, pu.SEED, real=orc.ORC_F32, rect=rect)
        res = pu.compare(img, ref, 8, rect)
        print(f"  96x96 crop against the float oracle ({time.perf_counter() - t0:.1f} s): {res['bad_pixels']} of {res['pixels']} pixels outside 1e-5, max rel {res['max_rel']:.2e}, "
              f"finite {bool(np.isfinite(img).all())}", flush=True)
        del sc
