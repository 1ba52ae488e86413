"""CRC of small frames of the BASELINE-shaped scenes for each render-kernel variant of the loaded library (SOLSTRALE_BUILD_DIR
selects it), one JSON line: {"<scene>/<kernel>": crc, ..}. The product library has kernel 1 only; the A/B build (_build_ab/,
-DSOL_AB_KERNELS) also 2, 3 (wavefront variants) and 4 (the pool kernel). Images are a pure function of (scene, seed): every CRC of a scene must be the same.
Usage: python tests/tools/frame_crc.py 1 2 3"""
import _paths  # noqa: F401  (sys.path)
import json
import sys
import zlib

import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, _abi, scenes

SCENES = {"c2": scenes.cornell_spheres, "c3": scenes.sponza_like, "c3h": lambda rc: scenes.sponza_like(rc, mesh="heterogeneous"),
          "test": scenes.create_test_scene}


def crcs(kernels, size=(250, 131), spp=37):
    """(a ragged frame - edge blocks with padding pixels - and a ragged last chunk: 37 = 2 x 16 + 5 samples)"""
    out = {}
    for name, make in SCENES.items():
        with DeviceScene(make(RenderConfig(size[0], size[1], spp))) as ds:
            for k in kernels:
                if name == "c3h" and k in (2, 3):
                    continue  # (the wavefront variants do not implement the consistency rule of scenes with needle triangles)
                ds.set_option(_abi.OPT_KERNEL, k)
                ds.clear()
                ds.render(0, spp, pu.SEED)
                out[f"{name}/{k}"] = zlib.crc32(ds.read().tobytes())
    return out


if __name__ == "__main__":
    print(json.dumps(crcs([int(a) for a in sys.argv[1:]] or [1])))
