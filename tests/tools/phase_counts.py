"""Executions of the kernel's phases per sample (counted build; not a pytest): search steps, shading passes, generate passes, and
the lanes enabled in each. Usage: python tests/tools/phase_counts.py [c1 c2 c3 c5]"""
import _paths  # noqa: F401
import ctypes as C
import sys

import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, _abi, scenes

if __name__ == "__main__":
    make = {"c1": scenes.cornell_box, "c2": scenes.cornell_spheres, "c3": scenes.sponza_like, "c5": scenes.statue_like,
            "c3h": lambda rc: scenes.sponza_like(rc, mesh="heterogeneous"), "test": lambda rc: scenes.create_test_scene(RenderConfig(800, 400, 16))}
    for w in (sys.argv[1:] or ["c2", "c3", "c5"]):
        with DeviceScene(make[w](RenderConfig(1920, 1080, 16))) as ds:
            ds.render(0, 16, pu.SEED, counted=True)
            st = _abi.SolStats()
            ds._chk(ds.lib.sol_stats(ds.h, C.byref(st)))
            p = [int(x) for x in st.phase]
            n = int(st.samples)
            print(f"{w}: per sample: search steps {p[1] / 64 / n:.4f} (lanes {p[0] / p[1] * 64:.1f}), shade passes {p[3] / 64 / n:.4f} (lanes {p[2] / p[3] * 64:.1f}), "
                  f"generate passes {p[5] / 64 / n:.4f} (lanes {p[4] / p[5] * 64:.1f}); rays {st.rays / n:.3f}, node visits {st.node_visits / n:.2f}", flush=True)
