"""Search-loop statistics of a counted render: lane-steps and wave-level step executions per ray (not a pytest).
Usage: [SOL_COLLAPSE=greedy] python step_stats.py [c2|c3|c5]"""
import _paths  # noqa: F401  (sys.path)
import sys

import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, _abi, scenes
import ctypes as C

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
make = {"c3": scenes.sponza_like, "c5": scenes.statue_like, "c2": scenes.cornell_spheres}[which]
with DeviceScene(make(RenderConfig(1920, 1080, 16))) as ds:
    ds.render(0, 16, pu.SEED, counted=True)
    st = _abi.SolStats()
    ds.lib.sol_stats(ds.h, C.byref(st))
    rays = st.rays
    lane_steps, slots = st.phase[0], st.phase[1]
    print(f"{which}: rays {rays}  nodes/ray {st.node_visits / rays:.2f}  prims/ray {(st.triangle_tests + st.quad_tests + st.sphere_tests) / rays:.2f}  "
          f"lane-steps/ray {lane_steps / rays:.2f}  wave step executions per ray {slots / 64 / rays:.4f}  occupancy {lane_steps / slots:.3f}", flush=True)
