"""CPU sweep (no GPU, not a pytest): the float oracle - the arithmetic contract the device is held to, fp32-only rules included - against
the double one - the reference's arithmetic - over seeded random scenes: relative difference of the frame means. Paths that round apart
give differences of either sign around 1e-4 for single scenes and no mean; a RULE that loses or invents energy shows as a signed mean
or as an outlier (round 4 found two that way: needle-shaped lights, and the needle rule inside a light's pdf_value).
Usage: python f32_vs_f64_sweep.py [general|needle|far] [first_seed] [count] [spp]   (far: the general scenes from 100 times the distance)"""
import _paths  # noqa: F401
import sys

import numpy as np

import orc
import parity_util as pu
import random_scenes

if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "general"
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    spp = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    gen = random_scenes.needle_scene if kind == "needle" else random_scenes.random_scene
    if kind == "far":
        gen = lambda seed, spp: random_scenes.random_scene(seed, spp=spp, far=100.0)  # noqa: E731
    rows = []
    for seed in range(first, first + count):
        sc = gen(seed, spp=spp)
        a, _ = orc.render(sc, 0, spp, pu.SEED, real=orc.ORC_F32)
        b, _ = orc.render(sc, 0, spp, pu.SEED, real=orc.ORC_F64)
        if not np.isfinite(b).all() or not np.isfinite(a).all() or b.mean() <= 0:
            print("seed", seed, "skipped (non-finite or black frame)", flush=True)
            continue
        rows.append((seed, (a.mean() - b.mean()) / b.mean()))
    rel = np.array([r for _, r in rows])
    order = np.argsort(-np.abs(rel))
    print(f"{kind}: {len(rows)} scenes from seed {first}, {spp} spp: mean signed {rel.mean():.2e} (+- {rel.std() / np.sqrt(len(rel)):.1e}), median |rel| {np.median(np.abs(rel)):.2e}; "
          f"largest: {[(rows[i][0], float('%.2e' % rel[i])) for i in order[:8]]}", flush=True)
