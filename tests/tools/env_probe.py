"""Debug helper: the pixels of the environment test scene (tests/test_environment.py) whose GPU sum differs from the fp32
oracle, the samples that differ, and those paths on GPU and oracle side by side.
Usage: python tests/tools/env_probe.py [width height spp]"""
import _paths  # noqa: F401  (sys.path)
import sys

import numpy as np

import parity_util as pu
import orc
from gpu_trace_case import fmt
from solstrale_amd import DeviceScene, PathTracingShader, RenderConfig, scenes

if __name__ == "__main__":
    w, h, spp = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (200, 100, 16)
    sc = scenes.create_test_scene_with_environment(RenderConfig(w, h, spp, PathTracingShader(50)))
    ref, _ = orc.render(sc, 0, spp, pu.SEED, real=orc.ORC_F32)
    with DeviceScene(sc) as ds:
        ds.render(0, spp, pu.SEED)
        img = ds.read().astype(np.float64)
        tol = pu.REL_TOL * np.abs(ref) + pu.REL_TOL * spp * 1e-2
        bad = np.argwhere((np.abs(img - ref) > tol).any(axis=-1))
        print(f"{len(bad)} pixels outside 1e-5", flush=True)
        for (y, x) in bad[:8]:
            print(f"pixel ({x},{y}): gpu {img[y, x]} oracle {ref[y, x]}")
            for s in range(spp):
                g, gc = ds.debug_path(int(x), int(y), s, pu.SEED)
                o, oc = orc.debug_path(sc, int(x), int(y), s, pu.SEED)
                if np.abs(gc - oc).max() <= 1e-6 * max(1e-2, np.abs(oc).max()):
                    continue
                print(f"  sample {s}: gpu colour {gc} ({len(g)} rays), oracle colour {oc} ({len(o)} rays)")
                for i in range(max(len(g), len(o))):
                    same = i < len(g) and i < len(o) and (g[i, :8].view(np.uint32) == o[i, :8].view(np.uint32)).all()
                    if same:
                        continue
                    print("   first difference at ray", i)
                    for j in range(max(0, i - 1), min(i + 2, max(len(g), len(o)))):
                        if j < len(g):
                            print("    gpu   ", fmt(g[j]))
                        if j < len(o):
                            print("    oracle", fmt(o[j]))
                    break
