"""One-off sweep: parity of seeded random scenes made of NEEDLE triangles (aspect 40:1 .. 2000:1, any orientation, textured or not, some of
them lights) against the fp32 oracle: the needle rule, the rotated fp32 records and the pre-splitting of the GPU tree builder together
(not a pytest). Usage: python needle_parity_sweep.py [first_seed] [count]"""
import _paths  # noqa: F401  (sys.path)
import sys

import numpy as np

import orc
import parity_util as pu
from solstrale_amd import DeviceScene


from random_scenes import needle_scene  # noqa: E402


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
