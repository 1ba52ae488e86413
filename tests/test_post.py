"""the post-processors that follow the path (to_rgb_color, BloomPostProcessor).

CPU part: pins the numpy restatement (oracle/post.py) with the reference's known-answer test of the blur weights
(src/util/gaussian.rs:33-44) and its bloom golden (tests/integration_tests.rs:239-254: resources/textures/bloom.png through
BloomPostProcessor::new(0.2, None, None), compared with tests/output/out_expected_bloom.jpg under the reference's criterion).
GPU part: the device kernels against that restatement, bit for bit."""
import os
import sys

import numpy as np
import pytest
from PIL import Image

import image_metric as im

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import post as opost  # noqa: E402  (test infrastructure)

KAT_WEIGHTS = [0.05448868454964294, 0.24420134200323332, 0.4026199468942474, 0.24420134200323332, 0.05448868454964294]


def bloom_input():
    """image_to_vec3 (tests/integration_tests.rs:295-303): rgb_to_vec3 of every pixel, `pixel as f64 * (1/255)`."""
    rgb = np.asarray(Image.open(os.path.join(HERE, "golden", "resources", "textures", "bloom.png")).convert("RGB"))
    return rgb.astype(np.float64) * (1.0 / 255.)


def test_gaussian_blur_weights_known_answer():
    w = opost.create_gaussian_blur_weights(5, 1.)
    assert list(w) == KAT_WEIGHTS
    assert abs(1. - sum(w)) < 1e-8


def test_library_gaussian_blur_weights_known_answer():
    import ctypes as C
    from solstrale_amd import _abi
    lib = _abi.load_hip()
    out = (C.c_double * 5)()
    assert lib.sol_gaussian_blur_weights(5, 1., out) == 0
    assert list(out) == KAT_WEIGHTS
    big = (C.c_double * 81)()
    assert lib.sol_gaussian_blur_weights(81, 81 / 5., big) == 0
    assert list(big) == list(opost.create_gaussian_blur_weights(81, 81 / 5.))


def test_bloom_oracle_matches_reference_golden():
    p = bloom_input()
    assert p.shape == (200, 200, 3)
    actual = opost.bloom_post_process(p, 1, 0.2)
    expected = np.asarray(Image.open(os.path.join(HERE, "golden", "expected", "out_expected_bloom.jpg")).convert("RGB"))
    score = im.compare_output(actual, expected)
    assert score > im.THRESHOLD, f"Comparison score for bloom is: {score}"
    # and the effect is really there: bloom only adds light
    assert (actual.astype(int) >= opost.to_rgb8(p, 1).astype(int)).all() and (actual != opost.to_rgb8(p, 1)).any()


def test_bloom_parameter_check_and_limits():
    p = np.ones((4, 6, 3))
    for bad in (-0.1, 0.51):
        with pytest.raises(ValueError, match="kernel_size_fraction must be between 0 and 0.5"):
            opost.bloom_intermediate(p, 1, bad)
    # kernel size 1 (fraction 0): blurred == bright, so a pixel over the threshold doubles; below it nothing changes
    out = opost.bloom_intermediate(p * 2., 1, 0.)
    assert (out == 4.).all()
    assert (opost.bloom_intermediate(p * .5, 1, 0.) == .5).all()
    # max_intensity caps the length of the bright pixel
    out = opost.bloom_intermediate(p * 2., 1, 0., threshold=1., max_intensity=1.)
    assert np.allclose(out, 2. + 1. / np.sqrt(3.))


# ---- device ---------------------------------------------------------------------------------------------------------------
def _device_scene(w, h):
    from solstrale_amd import DeviceScene, RenderConfig, scenes
    return DeviceScene(scenes.cornell_box(RenderConfig(w, h, 1)))


def _hdr(w, h, seed):
    rng = np.random.default_rng(seed)
    img = rng.random((h, w, 3)) ** 6 * 40.  # mostly dark, some pixels far above the threshold
    img[rng.random((h, w)) < 0.01] = 0.
    return img.astype(np.float32)


@pytest.mark.gpu
def test_device_bloom_matches_oracle_on_the_reference_input():
    import torch
    p = bloom_input().astype(np.float32)
    with _device_scene(200, 200) as ds:
        img = torch.from_numpy(p).cuda().contiguous()
        got = ds.bloom_rgb8(img.data_ptr(), 1, 0.2)
    want = opost.bloom_post_process(p.astype(np.float64), 1, 0.2)
    assert (got == want).all(), int((got != want).sum())
    expected = np.asarray(Image.open(os.path.join(HERE, "golden", "expected", "out_expected_bloom.jpg")).convert("RGB"))
    assert im.compare_output(got, expected) > im.THRESHOLD


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,frac,thr,maxi,spp", [(64, 48, 0.1, None, None, 1), (37, 23, 0.5, 2.5, 6., 3), (129, 65, 0., 1., None, 7),
                                                  (320, 180, 0.25, None, 4., 16),
                                                  (1100, 9, 0.3, None, None, 2),   # two row tiles of the LDS-tiled blur, a ragged second one (k = 661)
                                                  (1700, 5, 0.5, 1.5, None, 1)])   # k = 1701 > 1665 taps: the untiled blur
def test_device_bloom_intermediate_and_final_are_exact(w, h, frac, thr, maxi, spp):
    import torch
    p = _hdr(w, h, w * 1000 + h) * spp
    want64 = opost.bloom_intermediate(p.astype(np.float64), spp, frac, thr, maxi)
    with _device_scene(w, h) as ds:
        img = torch.from_numpy(p).cuda().contiguous()
        rgb = ds.bloom_rgb8(img.data_ptr(), spp, frac, thr, maxi)
        assert (img.cpu().numpy() == p).all()  # post_process leaves its input alone
        ds.bloom(img.data_ptr(), spp, frac, thr, maxi)
        ds.sync()
        torch.cuda.synchronize()
        inter = img.cpu().numpy()
    assert (rgb == opost.to_rgb8(want64, spp)).all()
    assert (inter == want64.astype(np.float32)).all()  # the device image keeps fp32 sums: the f64 result rounded once


@pytest.mark.gpu
def test_device_bloom_rejects_bad_fraction_and_chains_with_the_render():
    from solstrale_amd import DeviceError
    import parity_util as pu
    with _device_scene(96, 64) as ds:
        ds.render(0, 8, pu.SEED)
        ptr = ds.resolve_image()
        with pytest.raises(DeviceError, match="kernel_size_fraction must be between 0 and 0.5"):
            ds.bloom(ptr, 8, 0.6)
        sums = ds.read().astype(np.float64)
        ptr = ds.resolve_image()
        got = ds.bloom_rgb8(ptr, 8, 0.1, 1.0)
    assert (got == opost.bloom_post_process(sums, 8, 0.1, 1.0)).all()


@pytest.mark.gpu
def test_ray_trace_applies_the_post_processor_chain():
    """RenderConfig::post_processors (renderer/mod.rs:307-337): all but the last run intermediate_post_process, the last one
    post_process; an empty list yields progress without an image."""
    from solstrale_amd import BloomPostProcessor, DeviceScene, HostError, NopPostProcessor, RenderConfig, scenes
    import parity_util as pu
    spp = 12
    base = scenes.cornell_box(RenderConfig(96, 64, spp))
    with DeviceScene(base) as ds:
        ds.render(0, spp, pu.SEED)
        sums = ds.read().astype(np.float64)

    def run(chain):
        sc = scenes.cornell_box(RenderConfig(96, 64, spp, post_processors=chain))
        return sc.ray_trace()

    _, img = run([NopPostProcessor()])
    assert (img == opost.to_rgb8(sums, spp)).all()
    _, img = run([BloomPostProcessor(0.1, 1.0)])
    assert (img == opost.bloom_post_process(sums, spp, 0.1, 1.0)).all()
    # bloom as an intermediate step keeps fp32 sums on the device, then Nop
    _, img = run([BloomPostProcessor(0.1, 1.0), NopPostProcessor()])
    inter = opost.bloom_intermediate(sums, spp, 0.1, 1.0).astype(np.float32).astype(np.float64)
    assert (img == opost.to_rgb8(inter, spp)).all()
    _, img = run([NopPostProcessor(), BloomPostProcessor(0.05, 1.0, 2.0)])
    assert (img == opost.bloom_post_process(sums, spp, 0.05, 1.0, 2.0)).all()
    events, img = run([])
    assert img is None and len(events) == spp and not any(e[3] for e in events)
    with pytest.raises(HostError, match="kernel_size_fraction must be between 0 and 0.5"):
        run([BloomPostProcessor(0.7)])
