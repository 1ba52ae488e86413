"""Debug helper: prints the rays of one path on the GPU and in the fp32 oracle side by side.
Usage: python tests/gpu_trace_case.py <scene> x,y,sample [x,y,sample ...]"""
import _paths  # noqa: F401  (sys.path)
import sys

import numpy as np

import parity_util as pu
import orc
from gpu_full_oracle import scene_of
from solstrale_amd import DeviceScene


def fmt(r):
    ref = int(np.float32(r[7]).view(np.uint32))
    return (f"d{int(r[9]):2d} o=({r[0]:.5f},{r[1]:.5f},{r[2]:.5f}) dir=({r[3]:.6g},{r[4]:.6g},{r[5]:.6g}) t={r[6]:.9g} "
            f"ref={ref >> 28}:{ref & 0xFFFFFFF}")


if __name__ == "__main__":
    name = sys.argv[1]
    sc = scene_of(name, 64, 1920, 1080)
    with DeviceScene(sc) as ds:
        for a in sys.argv[2:]:
            x, y, s = (int(v) for v in a.split(","))
            g, gc = ds.debug_path(x, y, s, pu.SEED)
            o, oc = orc.debug_path(sc, x, y, s, pu.SEED)
            print(f"=== pixel ({x},{y}) sample {s}: gpu colour {gc} ({len(g)} rays), oracle colour {oc} ({len(o)} rays)")
            for i in range(max(len(g), len(o))):
                same = i < len(g) and i < len(o) and (g[i, :8].view(np.uint32) == o[i, :8].view(np.uint32)).all()
                if same:
                    continue
                print("  first difference at ray", i)
                for j in range(max(0, i - 1), min(i + 2, max(len(g), len(o)))):
                    if j < len(g):
                        print("   gpu   ", fmt(g[j]))
                    if j < len(o):
                        print("   oracle", fmt(o[j]))
                break
