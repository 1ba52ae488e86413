"""CPU experiment (no GPU, not a pytest): what the needle rule of the fp32 contract does to the image. Renders the heterogeneous atrium with
the oracle's three arithmetics - f64 (the reference's), f32 under the contract, and "plain f32" (the rule switched off in the checker by
ORC_NO_NEEDLE_RULE) - and prints their differences. The record of round 4 is profiles/r04_needle_rule_bias_and_vertex_order.txt.
Usage: python tests/tools/needle_bias.py [width height spp]"""
import _paths  # noqa: F401
import os
import subprocess
import sys

import numpy as np

SEED = 0x5017A1E


def render(mode, w, h, spp):
    import orc
    from solstrale_amd import RenderConfig, scenes
    sc = scenes.sponza_like(RenderConfig(w, h, spp), mesh="heterogeneous")
    img, st = orc.render(sc, 0, spp, SEED, real=orc.ORC_F64 if mode == "f64" else orc.ORC_F32)
    np.save(f"/tmp/needle_{mode}.npy", img / spp)
    print(mode, "mean", img.mean() / spp, "rays/sample", st["rays"] / st["samples"], flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        render(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
        sys.exit(0)
    w, h, spp = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (320, 180, 48)
    for mode in ("f64", "f32", "f32plain"):  # (the switch is read when the oracle instantiates the scene: one process per mode)
        env = dict(os.environ)
        if mode == "f32plain":
            env["ORC_NO_NEEDLE_RULE"] = "1"
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", mode, str(w), str(h), str(spp)], env=env)
    a, b, c = (np.load(f"/tmp/needle_{m}.npy") for m in ("f64", "f32", "f32plain"))

    def stats(x, y, name):
        d = x - y
        print(f"{name}: mean diff {d.mean():.3e}  rms {np.sqrt((d ** 2).mean()):.3e}  pixels differing > 1e-3: {(np.abs(d).max(axis=-1) > 1e-3).sum()}  max {np.abs(d).max():.3e}")

    stats(b, a, "f32 under the contract - f64")
    stats(c, a, "f32 without the needle rule - f64")
    stats(b, c, "with the rule - without")
