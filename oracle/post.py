"""oracle/post.py -- CPU restatement (numpy, f64) of the reference's post-processors on the path's output.

TEST INFRASTRUCTURE: imported only by tests/ (and never by the product). Each function cites the reference lines it follows.
Pinned by the reference's own known-answer test of the blur weights (src/util/gaussian.rs:33-44), the to_rgb_color cases
(src/util/rgb_color.rs:49-61) and the golden image tests/output/out_expected_bloom.jpg with its input
resources/textures/bloom.png (tests/integration_tests.rs:239-254), both committed under tests/golden/."""
import math

import numpy as np

F64_MAX = 1.7976931348623157e308


def create_gaussian_blur_weights(kernel_size, std_dev):
    """src/util/gaussian.rs:3-25 (math.exp = the C library's exp, as Rust's f64::exp; sum left to right)."""
    mean = (kernel_size - 1) / 2.0
    w = [math.exp(-0.5 * ((i - mean) / std_dev) * ((i - mean) / std_dev)) for i in range(kernel_size)]
    total = 0.0
    for x in w:
        total += x
    return np.array([x / total for x in w], dtype=np.float64)


def bloom_intermediate(pixel_colors, num_samples, kernel_size_fraction, threshold=None, max_intensity=None):
    """BloomPostProcessor::intermediate_post_process (src/post/bloom.rs:76-150). pixel_colors: (H, W, 3) f64 sums."""
    if not (0.0 <= kernel_size_fraction <= 0.5):
        raise ValueError("kernel_size_fraction must be between 0 and 0.5")  # bloom.rs:33-37
    p = np.asarray(pixel_colors, dtype=np.float64)
    h, w, _ = p.shape
    threshold = (math.sqrt(3.0) if threshold is None else threshold) * float(num_samples)        # bloom.rs:39,86
    max_intensity = (F64_MAX if max_intensity is None else max_intensity) * float(num_samples)   # bloom.rs:40,87
    kernel_size = int(kernel_size_fraction * float(w)) * 2 + 1                                    # bloom.rs:88
    half = kernel_size // 2
    weights = create_gaussian_blur_weights(kernel_size, kernel_size / 5.0)
    with np.errstate(all="ignore"):
        length = np.sqrt(p[..., 0] * p[..., 0] + p[..., 1] * p[..., 1] + p[..., 2] * p[..., 2])  # Vec3::length
        unit_scaled = (p / length[..., None]) * max_intensity                                     # p.unit() * max_intensity
        bright = np.where((length >= threshold)[..., None], np.where((length > max_intensity)[..., None], unit_scaled, p), 0.0)
        xs = np.arange(w)
        ys = np.arange(h)
        col = np.zeros_like(p)
        for i in range(kernel_size):  # bloom.rs:108-124: col += get_pixel_safe(x + i - half, y) * weights[i]
            col = col + bright[:, np.clip(xs + i - half, 0, w - 1), :] * weights[i]
        col2 = np.zeros_like(p)
        for i in range(kernel_size):  # bloom.rs:126-142
            col2 = col2 + col[np.clip(ys + i - half, 0, h - 1), :, :] * weights[i]
        return p + col2  # bloom.rs:144-148


def to_rgb8(pixel_colors, num_samples):
    """pixel_colors_to_rgb_image / to_rgb_color (src/post/mod.rs:57-77, src/util/rgb_color.rs:14-35):
    sqrt(col / spp) clamped to [-0.999, 0.999], times 256, `as u8` (saturating, NaN -> 0)."""
    p = np.asarray(pixel_colors, dtype=np.float64)
    with np.errstate(all="ignore"):
        v = np.sqrt((1.0 / float(num_samples)) * p)
        v = np.where(v < -0.999, -0.999, v)
        v = np.where(v > 0.999, 0.999, v)
        sc = 256.0 * v
        sc = np.where(np.isnan(sc), 0.0, np.clip(sc, 0.0, 255.0))
    return sc.astype(np.uint8)


def bloom_post_process(pixel_colors, num_samples, kernel_size_fraction, threshold=None, max_intensity=None):
    """BloomPostProcessor::post_process (bloom.rs:51-73)."""
    return to_rgb8(bloom_intermediate(pixel_colors, num_samples, kernel_size_fraction, threshold, max_intensity), num_samples)
