"""Pre-splitting sweep of the GPU tree builder (not a pytest): for each scene, the device build with several split budgets
(SOL_SPLIT, percent of the primitive count; 0 = off) and level slacks (SOL_SPLIT_SLACK): tree check (references, coverage), node
visits and primitive tests per ray (counted 16-spp render), render time at --spp, frame CRC (must not change).
Also the reinsertion rounds of the same build (SOL_REINSERT, SOL_REINSERT_STRIDE). A budget of -1 is the default (automatic) rule.
Usage: python tests/tools/split_sweep.py [c3 c3h c3hi c5 c2 test] [--spp N] [--budgets 0,10,30,60] [--slacks 3] [--reinsert 0,4] [--strides 1] [--check]"""
import _paths  # noqa: F401
import ctypes as C
import os
import sys
import time
import zlib

import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, _abi, scenes


def one(name, sc, spp, budgets, slacks, check, reinserts=(0,), strides=(1,)):
    base_crc = None
    for slack, b, rr, stride in [(sl, b, rr, st) for sl in slacks for b in budgets for rr in reinserts for st in (strides if rr else strides[:1])]:
        if True:
            if b >= 0:
                os.environ["SOL_SPLIT"] = str(b)
            else:
                os.environ.pop("SOL_SPLIT", None)  # the default: automatic
            os.environ["SOL_SPLIT_SLACK"] = str(slack)
            os.environ["SOL_REINSERT"] = str(rr)
            os.environ["SOL_REINSERT_STRIDE"] = str(stride)
            extra = ""
            if check:
                out = _abi.SolTreeCheck()
                rc = _abi.load_hip().sol_world_tree_check(sc.desc_ptr, -1, C.byref(out))
                d = out.as_dict()
                extra = (f"  check rc {rc} refs {d['n_leaf_refs']} (+{d['n_extra_references']}, {d['n_split_triangles']} split) wide {d['n_wide']} depth {d['depth']} "
                         f"violations {d['box_violations']}/{d['leaf_mismatches']}/{d['bad_empty_slots']}/{d['split_uncovered']}")
            t0 = time.perf_counter()
            with DeviceScene(sc, world_tree=_abi.TREE_DEVICE) as ds:
                t_create = time.perf_counter() - t0
                bt = ds.build_times()
                inf = ds.info()
                ds.render(0, spp, pu.SEED)
                ds.sync()
                best = 1e9
                for _ in range(3):
                    ds.clear()
                    t0 = time.perf_counter()
                    ds.render(0, spp, pu.SEED)
                    ds.sync()
                    best = min(best, time.perf_counter() - t0)
                crc = zlib.crc32(ds.read().tobytes())
                ds.clear()
                ds.render(0, 16, pu.SEED, counted=True)
                st = ds.stats()
            base_crc = base_crc if base_crc is not None else crc
            prims = st["triangle_tests"] + st["sphere_tests"] + st["quad_tests"]
            print(f"{name:6s} split {b:3d}% slack {slack} reinsert {rr}/{stride} ({inf['reinsertion_moves']} moves, area {inf['reinsertion_area_ratio']:.3f})  create {t_create * 1e3:7.1f} ms (device tree {bt['device_tree'] * 1e3:6.1f})  nodes/ray {st['node_visits'] / st['rays']:6.2f}  "
                  f"prims/ray {prims / st['rays']:5.2f}  area ratio {inf['split_area_ratio']:.3f} (+{inf['split_references']})  render {best * 1e3:8.2f} ms  crc {crc:08x}{'' if crc == base_crc else '  CRC CHANGED'}{extra}", flush=True)
    for k in ("SOL_SPLIT", "SOL_SPLIT_SLACK", "SOL_REINSERT", "SOL_REINSERT_STRIDE"):
        os.environ.pop(k, None)


if __name__ == "__main__":
    a = sys.argv[1:]

    def opt(flag, default):
        if flag in a:
            i = a.index(flag)
            v = a[i + 1]
            del a[i:i + 2]
            return v
        return default
    spp = int(opt("--spp", "64"))
    budgets = [int(x) for x in opt("--budgets", "0,10,30,60").split(",")]
    slacks = [int(x) for x in opt("--slacks", "3").split(",")]
    reinserts = [int(x) for x in opt("--reinsert", "0").split(",")]
    strides = [int(x) for x in opt("--strides", "1").split(",")]
    check = "--check" in a
    which = [x for x in a if not x.startswith("--")] or ["c3", "c3h"]
    rc = RenderConfig(1920, 1080, spp)
    make = {"c2": lambda: scenes.cornell_spheres(rc), "c3": lambda: scenes.sponza_like(rc), "c3h": lambda: scenes.sponza_like(rc, mesh="heterogeneous"),
            "c3hi": lambda: scenes.sponza_like(rc, mesh="heterogeneous", camera="interior"), "c3i": lambda: scenes.sponza_like(rc, camera="interior"),
            "c5": lambda: scenes.statue_like(rc), "c1x": lambda: scenes.cornell_box(rc), "test": lambda: scenes.create_test_scene(RenderConfig(800, 400, spp))}
    for w in which:
        one(w, make[w](), spp, budgets, slacks, check, reinserts, strides)
