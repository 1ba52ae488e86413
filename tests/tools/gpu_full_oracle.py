"""Debug helper: full-frame comparison of one or more library builds against the fp32 oracle (cached).
Usage: python tests/gpu_full_oracle.py <scene> <spp> <width> <height> <dir> [<dir> ...]   ('default' = _build)"""
import _paths  # noqa: F401  (sys.path)
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def scene_of(name, spp, w, h):
    import parity_util  # noqa: F401
    from solstrale_amd import RenderConfig, scenes
    rc = RenderConfig(w, h, spp)
    return {"c1": scenes.cornell_box, "c2": scenes.cornell_spheres, "c3": scenes.sponza_like,
            "c3h": lambda rc: scenes.sponza_like(rc, mesh="heterogeneous"), "c5": scenes.statue_like,
            "test": scenes.create_test_scene}[name](rc)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        import parity_util as pu
        from solstrale_amd import DeviceScene
        out, name, spp, w, h = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
        sc = scene_of(name, spp, w, h)
        with DeviceScene(sc) as ds:
            ds.render(0, spp, pu.SEED)
            np.save(out, ds.read())
        sys.exit(0)
    name, spp, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dirs = sys.argv[5:]
    import parity_util as pu
    import orc
    sc = scene_of(name, spp, w, h)
    ref, _ = orc.render(sc, 0, spp, pu.SEED, real=orc.ORC_F32)
    print("oracle done", flush=True)
    for d in dirs:
        env = dict(os.environ)
        if d != "default":
            env["SOLSTRALE_BUILD_DIR"] = os.path.abspath(d)
        out = "/tmp/full_img.npy"
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", out, name, str(spp), str(w), str(h)], env=env)
        img = np.load(out)
        res = pu.compare(img, ref, spp)
        d_abs = np.abs(img.astype(np.float64) - ref).max(axis=-1)
        ys, xs = np.nonzero(d_abs > 1e-4 * np.maximum(np.abs(ref).max(axis=-1), 1e-2 * spp))
        print(f"{d}: bad pixels {res['bad_pixels']} of {res['pixels']}, rmse(good) {res['rmse_mean_good']:.2e}; "
              f"first {[(int(x), int(y)) for x, y in list(zip(xs, ys))[:6]]}", flush=True)
