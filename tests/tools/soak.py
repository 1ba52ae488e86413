"""Determinism soak (not a pytest): the same frame over and over - C3, C5, the test scene (medium), the heterogeneous atrium - at a few sample counts, with the
scheduler options drawn at random per run (switch threshold, resident workgroups, fine tail, work order, background blocks): every CRC must equal the first
of its (scene, spp). A race in the reservoir, the fine tail's staging or the resolve order would show as ONE differing frame in thousands.
Usage: python tests/tools/soak.py [seconds, default 150]"""
import _paths  # noqa: F401
import sys
import time
import zlib

import numpy as np

import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, _abi, scenes

if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
    rng = np.random.default_rng(7)
    made = {"c3": lambda: scenes.sponza_like(RenderConfig(960, 540, 16)), "c5": lambda: scenes.statue_like(RenderConfig(960, 540, 16)),
            "test": lambda: scenes.create_test_scene(RenderConfig(800, 400, 16)), "c3h": lambda: scenes.sponza_like(RenderConfig(960, 540, 16), mesh="heterogeneous")}
    t_end = time.time() + budget
    total = 0
    for name, make in made.items():
        sc = make()
        with DeviceScene(sc) as ds:
            first = {}
            t_scene = time.time() + budget / len(made)
            runs = 0
            while time.time() < t_scene:
                spp = int(rng.choice([5, 16, 37, 64]))
                ds.set_option(_abi.OPT_SWITCH_BELOW, int(rng.choice([0, 8, 16, 16, 16, 32, 64])))
                ds.set_option(_abi.OPT_MAX_BLOCKS_PER_CU, int(rng.choice([0, 0, 0, 1, 2, 3])))
                ds.set_option(_abi.OPT_FINE_TAIL, int(rng.choice([-1, -1, 0, 4, 32, 64])))
                ds.set_option(_abi.OPT_WORK_ORDER, int(rng.integers(2)))
                ds.set_option(_abi.OPT_BACKGROUND_BLOCKS, int(rng.integers(2)))
                ds.clear()
                if rng.integers(3) == 0 and spp > 16:  # (a split of the sample range at a multiple of 16)
                    ds.render(0, 16, pu.SEED)
                    ds.render(16, spp - 16, pu.SEED)
                else:
                    ds.render(0, spp, pu.SEED)
                crc = zlib.crc32(ds.read().tobytes())
                if spp not in first:
                    first[spp] = crc
                assert crc == first[spp], f"{name} spp {spp}: run {runs} gave {crc:08x}, the first gave {first[spp]:08x}"
                runs += 1
            total += runs
            print(f"{name}: {runs} renders, CRCs by spp {{{', '.join(f'{k}: {v:08x}' for k, v in sorted(first.items()))}}} - all equal", flush=True)
    print(f"{total} renders, no frame differed", flush=True)
