"""Debug helper (not a pytest): for one pixel of the heterogeneous atrium, the samples whose colour differs between the oracle, the
default (pre-split) device tree and the unsplit one, with every ray of the three paths.
Usage: python tests/tools/pixel_probe.py x y n_samples"""
import _paths  # noqa: F401
import os
import sys

import numpy as np

import orc
import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, _abi, scenes

if __name__ == "__main__":
    x, y, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    sc = scenes.sponza_like(RenderConfig(1920, 1080, n), mesh="heterogeneous")
    d_split = DeviceScene(sc)
    os.environ["SOL_SPLIT"] = "0"
    d_plain = DeviceScene(sc)
    fmt = lambda r: f"o {r[0]:.6f},{r[1]:.6f},{r[2]:.6f} d {r[3]:.6f},{r[4]:.6f},{r[5]:.6f} t {r[6]:.7g} ref {r[7:8].view(np.uint32)[0]:08x}"
    for s in range(n):
        ra, ca = d_split.debug_path(x, y, s, pu.SEED)
        rb, cb = d_plain.debug_path(x, y, s, pu.SEED)
        ro, co = orc.debug_path(sc, x, y, s, pu.SEED, real=orc.ORC_F32)
        if (ca != cb).any() or np.abs(ca - co).max() > 1e-6 * max(1e-3, np.abs(co).max()) or not (len(ra) == len(rb) == len(ro)):
            print(f"sample {s}: split {ca} plain {cb} oracle {co}; rays {len(ra)} / {len(rb)} / {len(ro)}")
            for k in range(max(len(ra), len(rb), len(ro))):
                print(f"  ray {k}:\n    split  {fmt(ra[k]) if k < len(ra) else '-'}  raw {ra[k][6:] if k < len(ra) else ''}\n    plain  {fmt(rb[k]) if k < len(rb) else '-'}\n    oracle {fmt(ro[k]) if k < len(ro) else '-'}  raw {ro[k][6:] if k < len(ro) else ''}")
