"""Where does the `pathTracing` golden's residual come from? (not a pytest; CPU only, f64 oracle)
Renders create_test_scene (tests/scenes.rs:17-122) at high spp and variants of it, and compares region-wise with the reference's
out_expected_pathTracing.jpg (25 spp, tests/integration_tests.rs:26-40) under the reference's metric.
Usage: python tests/tools/golden_residual.py [spp]"""
import _paths  # noqa: F401
import os
import sys

import numpy as np
from PIL import Image

import image_metric as im
import orc
import parity_util as pu
from solstrale_amd import PathTracingShader, RenderConfig, scenes

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "expected", "out_expected_pathTracing.jpg")


def small(rgb8):
    return im.resize_gaussian(rgb8, 100, 50).astype(np.float64)


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    gold = np.asarray(Image.open(GOLD).convert("RGB"))
    g = small(gold)
    sc = scenes.create_test_scene(RenderConfig(200, 100, spp, PathTracingShader(50)))
    sums, _ = orc.render(sc, 0, spp, pu.SEED, real=orc.ORC_F64)
    hi = im.sums_to_rgb8(sums, spp)
    print(f"oracle {spp} spp vs golden: score {im.compare_output(hi, gold):.4f}; mean diff (oracle - golden) RGB of 255: {(small(hi) - g).mean(axis=(0, 1)).round(2)}")
    # the golden is a 25-spp image: tone-mapping (sqrt) a noisy pixel before the 100x50 average biases it DOWN (Jensen); emulate
    # it with independent 25-spp renders of the oracle
    tm = []
    for k in range(8):
        s25, _ = orc.render(sc, 1000 + 25 * k, 25, pu.SEED, real=orc.ORC_F64)
        tm.append(im.sums_to_rgb8(s25, 25))
    scores = [im.compare_output(t, gold) for t in tm]
    cross = [im.compare_output(tm[i], tm[j]) for i in range(4) for j in range(4, 8)]
    mean25 = np.mean([small(t) for t in tm], axis=0)
    print(f"oracle 25 spp x8 vs golden: scores {np.round(scores, 4)}; between two independent 25-spp oracle renders: {np.round(cross[:4], 4)} (mean {np.mean(cross):.4f})")
    print(f"mean of the 25-spp tone-mapped oracle images - golden, RGB of 255: {(mean25 - g).mean(axis=(0, 1)).round(2)};  {spp}-spp image - mean 25-spp image: {(small(hi) - mean25).mean(axis=(0, 1)).round(2)}")
    d = mean25 - g
    # regions of the 100x50 thumbnail: rows = top (background/lights) / middle (objects) / bottom (ground); 5 columns
    for name, rows in (("top", slice(0, 15)), ("middle", slice(15, 35)), ("bottom", slice(35, 50))):
        cells = [d[rows, c * 20:(c + 1) * 20].mean().round(1) for c in range(5)]
        print(f"  25-spp oracle - golden, {name:6s} rows, five column bands: {cells}")
    np.save("/tmp/golden_residual.npy", d)


if __name__ == "__main__":
    main()
