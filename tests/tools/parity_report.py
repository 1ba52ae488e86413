"""Prints the parity figures of the BASELINE configs (GPU fp32 vs oracle fp32, fixed seed) as JSON lines (not a pytest)."""
import _paths  # noqa: F401  (sys.path)
import json

import orc
import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, scenes

CASES = [("C1 full 400x400x50", lambda: scenes.cornell_box(RenderConfig(400, 400, 50)), 50, None),
         ("C2 crop 128x128x16 @1080p", lambda: scenes.cornell_spheres(RenderConfig(1920, 1080, 16)), 16, (900, 500, 1028, 628)),
         ("C3 crop 128x128x16 @1080p", lambda: scenes.sponza_like(RenderConfig(1920, 1080, 16)), 16, (900, 500, 1028, 628)),
         ("C5 crop 128x128x16 @1080p", lambda: scenes.statue_like(RenderConfig(1920, 1080, 16)), 16, (896, 476, 1024, 604))]
for name, make, spp, rect in CASES:
    sc = make()
    with DeviceScene(sc) as ds:
        ds.render(0, spp, pu.SEED)
        img = ds.read()
    ref, _ = orc.render(sc, 0, spp, pu.SEED, real=orc.ORC_F32, rect=rect)
    res = pu.compare(img, ref, spp, rect)
    print(json.dumps({"case": name, **{k: (float(v) if hasattr(v, "__float__") else v) for k, v in res.items()}}), flush=True)
