"""Lane occupancy of the two parts of a search step (not a pytest). Needs the probe build of the device library:
  python tests/tools/variants.py build probe="-DSOL_PROBE_STEP"                                   (CPU container)
  SOLSTRALE_BUILD_DIR=_var/probe python tests/tools/step_probe.py [c1 c2 c3]                      (GPU box)
Prints, per scene: lanes active per executed node part and primitive part, the share of a node part's lanes that hold
primitives instead (and how many of those were then postponed), node / primitive part executions per node visit."""
import _paths  # noqa: F401
import sys

import parity_util as pu
from solstrale_amd import DeviceScene, PathTracingShader, RenderConfig, scenes
from solstrale_amd import _abi
import ctypes as C

if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c3"]
    cfg = RenderConfig(1920, 1080, 8, PathTracingShader(50))
    make = {"c1": scenes.cornell_box, "c2": scenes.cornell_spheres, "c3": scenes.sponza_like}
    for w in which:
        with DeviceScene(make[w](cfg)) as ds:
            ds.render(0, 8, pu.SEED, counted=True)
            st = _abi.SolStats()
            ds._chk(ds.lib.sol_stats(ds.h, C.byref(st)))
            p = [int(x) for x in st.phase]
            d = st.as_dict()
            prims = d["triangle_tests"] + d["sphere_tests"] + d["quad_tests"]
            print(f"{w}: node part {p[0] / p[1] * 64:5.1f} of 64 lanes ({p[1] // 64} executions, {d['node_visits']} visits), "
                  f"primitive part {p[2] / p[3] * 64:5.1f} of 64 ({p[3] // 64} executions, {prims} tests); during node parts "
                  f"{p[4] / p[1] * 64:4.1f} lanes hold primitives, finished or idle {64 - (p[0] + p[4]) / p[1] * 64:4.1f}; "
                  f"postponed lane-steps {p[5]} ({p[5] / max(p[4], 1):.2f} of the holding ones); rays {d['rays']}", flush=True)
