import sys, time
sys.path.insert(0, "tests/tools"); import _paths
from solstrale_amd import DeviceScene, RenderConfig, scenes
for name, mk in (("c3", scenes.sponza_like), ("c5", scenes.statue_like), ("c2", scenes.cornell_spheres)):
    sc = mk(RenderConfig(1920, 1080, 16))
    for k in range(2):
        t = time.time()
        with DeviceScene(sc) as ds:
            dt = time.time() - t
            print(name, "create %.3f s" % dt, {k: round(v, 3) for k, v in ds.build_times().items()}, ds.info()["tree_name"], flush=True)
