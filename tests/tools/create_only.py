"""Scene creation alone (not a pytest), for a kernel trace of the tree builder: python tests/tools/create_only.py [c3|c5|c2|c3h|test|c1]
(rocprofv3 --kernel-trace --stats -- python3 tests/tools/create_only.py c5)"""
import _paths  # noqa: F401
import sys
import time

from solstrale_amd import DeviceScene, RenderConfig, scenes

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "c5"
    make = {"c2": scenes.cornell_spheres, "c3": scenes.sponza_like, "c5": scenes.statue_like, "c3h": lambda rc: scenes.sponza_like(rc, mesh="heterogeneous"),
            "test": lambda rc: scenes.create_test_scene(rc), "c1": scenes.cornell_box}[which]
    sc = make(RenderConfig(1920, 1080, 16))
    for k in range(2):
        t0 = time.perf_counter()
        with DeviceScene(sc) as ds:
            print(f"{which}: sol_scene_create {1e3 * (time.perf_counter() - t0):.1f} ms {ds.build_times()}", flush=True)
