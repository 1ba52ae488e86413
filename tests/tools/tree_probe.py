"""Prints the world-tree probe of the BASELINE scenes for a candidate list (not a pytest).
Usage: SOL_SAH_LIST=4,8,16 python tree_probe.py"""
import _paths  # noqa: F401  (sys.path)
import os

from solstrale_amd import DeviceScene, RenderConfig, scenes

os.environ["SOL_VERBOSE"] = "1"
for name, make in (("c2", scenes.cornell_spheres), ("c3", scenes.sponza_like), ("c5", scenes.statue_like), ("test", scenes.create_test_scene)):
    print(name, flush=True)
    DeviceScene(make(RenderConfig(1920, 1080, 16))).close()
