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


class WindowScene:
    """The pixels [x0, x1) x [y0, y1) (row 0 = top) of `scene` as a frame of their own: the same world, the same pixel footprints - the
    camera's lower-left corner moved to the window's, its horizontal / vertical spans scaled by (w' - 1) / (W - 1) and (h' - 1) / (H - 1), so
    that the window's (x' + xi) / (w' - 1) addresses the point the frame's (x0 + x' + xi) / (W - 1) does (src/renderer/mod.rs:261-264,
    src/camera.rs:77-89). Both the oracle and the device take it like any scene (desc_ptr / width / height): a crop of a BASELINE frame
    whose RAY COUNTS can be compared, which a rect of the full frame does not give on the device side. (The random streams are keyed by
    the window's own pixel indices: it is another sample set of the same pixels, not a bit-copy of the frame's crop.)"""

    def __init__(self, scene, rect):
        import ctypes as C
        from solstrale_amd import _abi
        x0, y0, x1, y1 = rect
        W, H = scene.width, scene.height
        w, h = x1 - x0, y1 - y0
        assert 0 <= x0 < x1 <= W and 0 <= y0 < y1 <= H and w >= 2 and h >= 2 and W >= 2 and H >= 2
        self._parent = scene  # (owns the arrays the copied descriptor points into)
        self.desc = _abi.SolSceneDesc()
        C.memmove(C.byref(self.desc), C.byref(scene.desc), C.sizeof(_abi.SolSceneDesc))
        cam = self.desc.camera
        fu, fv = x0 / (W - 1.0), (H - y1) / (H - 1.0)
        su, sv = (w - 1.0) / (W - 1.0), (h - 1.0) / (H - 1.0)
        for k in range(3):
            cam.lower_left_corner[k] = cam.lower_left_corner[k] + cam.horizontal[k] * fu + cam.vertical[k] * fv
            cam.horizontal[k] = cam.horizontal[k] * su
            cam.vertical[k] = cam.vertical[k] * sv
        self.desc.width, self.desc.height = w, h
        self.desc_ptr = C.pointer(self.desc)
        self.width, self.height = w, h
