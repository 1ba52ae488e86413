"""The reference's golden-image criterion (tests/integration_tests.rs:24,326-349), re-implemented:

    resize both images to 100x50 with image::imageops::FilterType::Gaussian,
    score = image_compare::rgb_similarity_structure(RootMeanSquared), assert score > 0.95.

Third-party pieces restated from their published algorithms (the crates are not available offline):
  * `image` 0.25 resize: separable sampling, vertical pass into f32 then horizontal pass, kernel gaussian(x, sigma=0.5)
    with support 3.0, window widened by the down-scaling ratio, weights normalised, result rounded to u8;
  * `image-compare` 0.4 RMS: per channel score = 1 - sqrt(mean(((a - b) / 255)^2)); the RGB score is the minimum channel.
"""
import numpy as np

THRESHOLD = 0.95  # IMAGE_COMPARISON_SCORE_THRESHOLD


def _gaussian(x, r=0.5):
    return np.exp(-(x * x) / (2.0 * r * r)) / (np.sqrt(2.0 * np.pi) * r)


def _sample_axis(img, new_len, axis):
    """img: float array; resamples `axis` to new_len (image::imageops::sample::{vertical,horizontal}_sample)."""
    img = np.moveaxis(img, axis, 0)
    n = img.shape[0]
    ratio = n / new_len
    sratio = max(ratio, 1.0)
    src_support = 3.0 * sratio
    out = np.zeros((new_len,) + img.shape[1:], dtype=np.float32)
    for o in range(new_len):
        inputx = (o + 0.5) * ratio
        left = int(np.clip(np.floor(inputx - src_support), 0, n - 1))
        right = int(np.clip(np.ceil(inputx + src_support), left + 1, n))
        c = inputx - 0.5
        idx = np.arange(left, right)
        w = _gaussian((idx - c) / sratio).astype(np.float32)
        w /= w.sum()
        out[o] = np.tensordot(w, img[left:right].astype(np.float32), axes=(0, 0))
    return np.moveaxis(out, 0, axis)


def resize_gaussian(rgb8, width, height):
    a = np.asarray(rgb8, dtype=np.float32)
    v = _sample_axis(a, height, 0)
    h = _sample_axis(v, width, 1)
    return np.clip(np.rint(h), 0, 255).astype(np.uint8)


def rms_score(a_rgb8, b_rgb8):
    a = np.asarray(a_rgb8, dtype=np.float64) / 255.0
    b = np.asarray(b_rgb8, dtype=np.float64) / 255.0
    per_channel = 1.0 - np.sqrt(((a - b) ** 2).reshape(-1, 3).mean(axis=0))
    return float(per_channel.min())


def compare_output(actual_rgb8, expected_rgb8):
    """compare_output (tests/integration_tests.rs:326-349) without the file writes."""
    return rms_score(resize_gaussian(expected_rgb8, 100, 50), resize_gaussian(actual_rgb8, 100, 50))


def sums_to_rgb8(sums, spp):
    """NopPostProcessor: pixel_colors_to_rgb_image + to_rgb_color (src/post/mod.rs:57-77, src/util/rgb_color.rs:14-35)."""
    c = np.sqrt(np.asarray(sums, dtype=np.float64) * (1.0 / spp))
    c = np.clip(c, -0.999, 0.999)
    v = 256.0 * c
    v = np.where(np.isnan(v), 0.0, v)
    return np.clip(np.floor(v), 0, 255).astype(np.uint8)
