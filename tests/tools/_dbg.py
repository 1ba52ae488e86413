import sys, os
sys.path.insert(0,"solstrale-rust_amd"); sys.path.insert(0,"tests")
import numpy as np
from solstrale_amd import DeviceScene, RenderConfig, scenes
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30001
sc = scenes.sponza_like(RenderConfig(192,108,4), n_triangles=n, texture_size=64) if which == "c3" else scenes.cornell_spheres(RenderConfig(192,108,4))
print("scene built", flush=True)
ds = DeviceScene(sc)
print("created", ds.build_times(), flush=True)
ds.render(0, 4, 1)
img = ds.read()
print("rendered", float(img.mean()), flush=True)
