"""Debug helper: renders one scene with two library builds (SOLSTRALE_BUILD_DIR A/B, separate processes), finds the pixels
that differ and asks the fp32 oracle which build is right. Usage: python tests/gpu_diff.py <scene> <spp> <dirA> <dirB>"""
import _paths  # noqa: F401  (sys.path)
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def scene_of(name, spp):
    import parity_util  # noqa: F401
    from solstrale_amd import RenderConfig, scenes
    if name == "c2":
        return scenes.cornell_spheres(RenderConfig(1920, 1080, spp))
    if name == "c3":
        return scenes.sponza_like(RenderConfig(1920, 1080, spp))
    return scenes.cornell_box(RenderConfig(1920, 1080, spp))


def render_to(path, name, spp):
    import parity_util as pu
    from solstrale_amd import DeviceScene
    sc = scene_of(name, spp)
    with DeviceScene(sc) as ds:
        ds.render(0, spp, pu.SEED)
        np.save(path, ds.read())


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        render_to(sys.argv[2], sys.argv[3], int(sys.argv[4]))
        sys.exit(0)
    name, spp, da, db = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    imgs = []
    for i, d in enumerate((da, db)):
        env = dict(os.environ)
        if d != "default":
            env["SOLSTRALE_BUILD_DIR"] = os.path.abspath(d)
        out = f"/tmp/diff_{i}.npy"
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", out, name, str(spp)], env=env)
        imgs.append(np.load(out))
    a, b = imgs
    diff = (a != b).any(axis=-1)
    ys, xs = np.nonzero(diff)
    print(f"{diff.sum()} pixels differ between A={da} and B={db}")
    if len(xs) == 0:
        sys.exit(0)
    import parity_util as pu
    import orc
    sc = scene_of(name, spp)
    for y, x in list(zip(ys, xs))[:12]:
        ref, _ = orc.render(sc, 0, spp, pu.SEED, real=orc.ORC_F32, rect=(int(x), int(y), int(x) + 1, int(y) + 1), threads=1)
        r = ref[y, x]
        ea, eb = np.abs(a[y, x] - r).max(), np.abs(b[y, x] - r).max()
        print(f"pixel ({x},{y}): A {a[y, x]} B {b[y, x]} oracle {r}  |A-o| {ea:.3e} |B-o| {eb:.3e}")
