"""Search / service switch threshold (SOL_OPT_SWITCH_BELOW) against the workloads (not a pytest): ms per frame at each threshold.
Usage: python tests/tools/switch_sweep.py [spp] [thresholds..]"""
import _paths  # noqa: F401
import sys
import time

import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, _abi, scenes

if __name__ == "__main__":
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    ths = [int(a) for a in sys.argv[2:]] or [8, 12, 16, 20, 24, 28, 32, 40]
    for name, make in (("c1", scenes.cornell_box), ("c2", scenes.cornell_spheres), ("c3", scenes.sponza_like),
                       ("c3i", lambda rc: scenes.sponza_like(rc, camera="interior")), ("c5", scenes.statue_like),
                       ("c5c", lambda rc: scenes.statue_like(rc, camera="closeup")), ("test", scenes.create_test_scene),
                       ("c3h", lambda rc: scenes.sponza_like(rc, mesh="heterogeneous"))):
        with DeviceScene(make(RenderConfig(1920, 1080, spp))) as ds:
            row = []
            for th in ths:
                ds.set_option(_abi.OPT_SWITCH_BELOW, th)
                best = 1e9
                for _ in range(3):
                    ds.clear()
                    ds.sync()
                    t0 = time.perf_counter()
                    ds.render(0, spp, pu.SEED)
                    ds.sync()
                    best = min(best, time.perf_counter() - t0)
                row.append(best * 1e3)
            print(f"{name:5s} " + " ".join(f"{th}:{ms:7.2f}" for th, ms in zip(ths, row)), flush=True)
