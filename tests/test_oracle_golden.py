"""Pins the oracle (f64 instantiation = the reference's arithmetic) against the reference's own golden images under the
reference's own criterion (tests/integration_tests.rs:24,326-349): the fixtures under tests/golden/expected/ are the
reference's tests/output/out_expected_*.jpg, the textures under tests/golden/resources/ its resources/textures/*."""
import os

import numpy as np
import pytest
from PIL import Image

import image_metric as im
import orc
import parity_util as pu
from ref_cases import CASES

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "expected")


@pytest.mark.parametrize("name,factory,w,h,ref_spp,spp", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_reference_golden(name, factory, w, h, ref_spp, spp):
    scene = factory(spp)
    sums, _ = orc.render(scene, 0, spp, pu.SEED, real=orc.ORC_F64)
    with np.errstate(invalid="ignore"):  # SimpleShader colours can be negative: sqrt -> NaN -> `as u8` = 0 like the reference
        actual = im.sums_to_rgb8(sums, spp)
    expected = np.asarray(Image.open(os.path.join(GOLDEN, f"out_expected_{name}.jpg")).convert("RGB"))
    score = im.compare_output(actual, expected)
    assert score > im.THRESHOLD, f"Comparison score for {name} is: {score}"


def test_metric_known_values():
    a = np.zeros((50, 100, 3), np.uint8)
    b = np.full((50, 100, 3), 255, np.uint8)
    assert im.rms_score(a, a) == 1.0
    assert im.rms_score(a, b) == 0.0
    c = a.copy()
    c[..., 1] = 51  # 0.2 in one channel -> min channel score 0.8
    assert abs(im.rms_score(a, c) - 0.8) < 1e-12
    # resizing a constant image keeps it constant; the identity size keeps a smooth image almost unchanged
    g = np.full((200, 300, 3), 77, np.uint8)
    assert (im.resize_gaussian(g, 100, 50) == 77).all()
