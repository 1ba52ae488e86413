"""World-tree comparison (not a pytest): for the BASELINE-shaped scenes, the host path (four candidates + probe) against the
GPU builder (SolCreateOptions.world_tree = SOL_TREE_DEVICE): sol_scene_create time split, node visits and primitive tests per
ray (counted 16-spp render), render time, frame CRC (must be identical). SOL_VERBOSE=1 also prints the host probe.
Usage: python tests/tools/tree_probe.py [c2 c3 c5]"""
import _paths  # noqa: F401
import sys
import time
import zlib

import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, _abi, scenes


def one(name, sc, spp=64):
    for label, tree in (("host", _abi.TREE_HOST_PROBE), ("device", _abi.TREE_DEVICE)):
        t0 = time.perf_counter()
        with DeviceScene(sc, world_tree=tree) as ds:
            t_create = time.perf_counter() - t0
            bt = ds.build_times()
            ds.render(0, spp, pu.SEED)
            ds.sync()
            ds.clear()
            t0 = time.perf_counter()
            ds.render(0, spp, pu.SEED)
            ds.sync()
            dt = time.perf_counter() - t0
            crc = zlib.crc32(ds.read().tobytes())
            ds.clear()
            ds.render(0, 16, pu.SEED, counted=True)
            st = ds.stats()
        prims = st["triangle_tests"] + st["sphere_tests"] + st["quad_tests"]
        print(f"{name:8s} {label:6s} create {t_create * 1e3:7.1f} ms (host trees {bt['host_trees'] * 1e3:6.1f}, device tree {bt['device_tree'] * 1e3:6.1f}, "
              f"upload {bt['upload'] * 1e3:6.1f}, probes {bt['probes'] * 1e3:6.1f})  nodes/ray {st['node_visits'] / st['rays']:5.2f}  "
              f"prims/ray {prims / st['rays']:5.2f}  render {dt * 1e3:7.2f} ms  crc {crc:08x}", flush=True)


if __name__ == "__main__":
    which = [a for a in sys.argv[1:]] or ["c2", "c3"]
    if "c1" in which:
        one("cornell", scenes.cornell_box(RenderConfig(1920, 1080, 64)))
    if "c2" in which:
        one("spheres", scenes.cornell_spheres(RenderConfig(1920, 1080, 64)))
    if "c3" in which:
        one("sponza", scenes.sponza_like(RenderConfig(1920, 1080, 64)))
    if "c5" in which:
        one("statue", scenes.statue_like(RenderConfig(1920, 1080, 64)))
    if "test" in which:
        one("test", scenes.create_test_scene(RenderConfig(800, 400, 64)))
