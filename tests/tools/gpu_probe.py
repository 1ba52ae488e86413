"""Debug helper: for given pixels of a scene, finds the sample indices whose GPU colour differs from the fp32 oracle.
Usage: python tests/gpu_probe.py <scene> <n_samples> x,y [x,y ...]"""
import _paths  # noqa: F401  (sys.path)
import sys

import numpy as np

import parity_util as pu
import orc
from gpu_full_oracle import scene_of
from solstrale_amd import DeviceScene

if __name__ == "__main__":
    name, n = sys.argv[1], int(sys.argv[2])
    pixels = [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:]]
    sc = scene_of(name, n, 1920, 1080)
    with DeviceScene(sc) as ds:
        for s in range(n):
            ds.clear()
            ds.render(s, 1, pu.SEED)
            img = ds.read()
            for (x, y) in pixels:
                ref, st = orc.render(sc, s, 1, pu.SEED, real=orc.ORC_F32, rect=(x, y, x + 1, y + 1), threads=1)
                g, r = img[y, x].astype(np.float64), ref[y, x]
                if np.abs(g - r).max() > 1e-5 * max(1e-2, np.abs(r).max()):
                    print(f"pixel ({x},{y}) sample {s}: gpu {g} oracle {r} oracle rays {st['rays']}", flush=True)
