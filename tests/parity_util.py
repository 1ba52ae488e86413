"""Shared helpers for the parity tests: image comparison at the north star's tolerances."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "solstrale-rust_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

SEED = 0x5017A1E
REL_TOL = 1e-5  # north star: "matches the CPU reference at a fixed RNG seed within 1e-5 per-channel relative error"


def compare(gpu_sum, ref_sum, spp, rect=None):
    """Compares per-pixel SUMS over `spp` samples. Returns a dict:
    bad_pixels  : pixels where some channel differs by more than REL_TOL relative (floor: REL_TOL absolute on the mean)
    max_rel     : largest relative error among the pixels within tolerance
    rmse_mean   : per-channel RMSE of the per-sample means over the compared region
    """
    g = np.asarray(gpu_sum, dtype=np.float64)
    r = np.asarray(ref_sum, dtype=np.float64)
    if rect:
        x0, y0, x1, y1 = rect
        g, r = g[y0:y1, x0:x1], r[y0:y1, x0:x1]
    diff = np.abs(g - r)
    tol = REL_TOL * np.abs(r) + REL_TOL * spp * 1e-2
    bad = (diff > tol).any(axis=-1)
    rel = diff / np.maximum(np.abs(r), 1e-2 * spp)
    ok_rel = rel[~bad] if (~bad).any() else np.zeros(1)
    return {
        "pixels": int(bad.size),
        "bad_pixels": int(bad.sum()),
        "max_rel": float(ok_rel.max()),
        "rmse_mean": float(np.sqrt(((g - r) ** 2).mean()) / spp),
        "rmse_mean_good": float(np.sqrt((((g - r) ** 2)[~bad]).mean()) / spp) if (~bad).any() else 0.0,
        "mean_gpu": float(g.mean() / spp),
        "mean_ref": float(r.mean() / spp),
    }
