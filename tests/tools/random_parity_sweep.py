"""One-off sweep: parity of many seeded random scenes (tests/random_scenes.py) against the fp32 oracle (not a pytest).
Usage: python random_parity_sweep.py [first_seed] [count]"""
import _paths  # noqa: F401  (sys.path)
import sys

import numpy as np

import orc
import parity_util as pu
import random_scenes
from solstrale_amd import DeviceScene

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad_scenes = []
for seed in range(first, first + count):
    sc = random_scenes.random_scene(seed)
    with DeviceScene(sc) as ds:
        ds.render(0, 4, pu.SEED)
        img = ds.read()
    ref, _ = orc.render(sc, 0, 4, pu.SEED, real=orc.ORC_F32)
    res = pu.compare(img, ref, 4)
    if res["bad_pixels"] or not np.isfinite(img).all():
        bad_scenes.append((seed, int(res["bad_pixels"]), float(res["max_rel"])))
print(f"{count} scenes from seed {first}: {len(bad_scenes)} with pixels over 1e-5: {bad_scenes}", flush=True)
