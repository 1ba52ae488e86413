"""Debug helper (not a pytest): renders a scene with two settings of the pre-split budget (SOL_SPLIT a / b) and, where the frames
differ, finds the sample and prints the path under both trees (sol_debug_path) next to the oracle's colour.
Usage: python tests/tools/split_diff.py <c3h|c3|c5> <spp> <split a> <split b> [slack]"""
import _paths  # noqa: F401
import os
import sys

import numpy as np

import orc
import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, _abi, scenes

if __name__ == "__main__":
    name, spp, sa, sb = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    slack = sys.argv[5] if len(sys.argv) > 5 else "3"
    rc = RenderConfig(1920, 1080, spp)
    sc = {"c3h": lambda: scenes.sponza_like(rc, mesh="heterogeneous"), "c3": lambda: scenes.sponza_like(rc), "c5": lambda: scenes.statue_like(rc)}[name]()
    os.environ["SOL_SPLIT_SLACK"] = slack
    os.environ["SOL_SPLIT"] = sa
    da = DeviceScene(sc, world_tree=_abi.TREE_DEVICE)
    os.environ["SOL_SPLIT"] = sb
    db = DeviceScene(sc, world_tree=_abi.TREE_DEVICE)
    da.render(0, spp, pu.SEED)
    db.render(0, spp, pu.SEED)
    ia, ib = da.read(), db.read()
    bad = np.argwhere((ia != ib).any(axis=-1))
    print(f"{len(bad)} pixels differ", flush=True)
    for (y, x) in bad[:4]:
        print(f"pixel ({x},{y}): a {ia[y, x]} b {ib[y, x]}")
        for s in range(spp):
            ra, ca = da.debug_path(int(x), int(y), s, pu.SEED)
            rb, cb = db.debug_path(int(x), int(y), s, pu.SEED)
            if (ca != cb).any() or len(ra) != len(rb):
                ref, st = orc.render(sc, s, 1, pu.SEED, real=orc.ORC_F32, rect=(int(x), int(y), int(x) + 1, int(y) + 1), threads=1)
                print(f"  sample {s}: colour a {ca} b {cb} oracle {ref[y, x]}  rays a {len(ra)} b {len(rb)}")
                for k in range(max(len(ra), len(rb))):
                    fa = ra[k] if k < len(ra) else None
                    fb = rb[k] if k < len(rb) else None
                    def fmt(r):
                        if r is None:
                            return "-"
                        return f"o {r[0]:.6f},{r[1]:.6f},{r[2]:.6f} d {r[3]:.6f},{r[4]:.6f},{r[5]:.6f} t {r[6]:.7g} ref {r[7:8].view(np.uint32)[0]:08x} dfs {r[8:9].view(np.uint32)[0]}"
                    print(f"    ray {k}: a {fmt(fa)}\n           b {fmt(fb)}")
                break
