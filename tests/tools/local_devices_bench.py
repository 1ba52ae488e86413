"""One process, n handles (not a pytest): times the C3 job (1080p, --spp, default 512) as ray_trace() on several devices does it - every handle renders its
blocks of the frame, sol_gather_local brings them to the first device - and prints Msamples/s per device list. On a node: --devices 0 0,1 0,1,2,3
0,1,2,3,4,5,6,7 is a strong-scaling curve from ONE process (no launcher, no RCCL). On a one-GPU box the lists can only repeat device 0: what that shows is
the cost of the mechanism (n persistent kernels sharing one GPU, n creations, the gather), not scaling.
Usage: python tests/tools/local_devices_bench.py [--spp N] [--workload c3|c5|c2] --devices 0 0,0 0,0,0,0"""
import _paths  # noqa: F401
import ctypes as C
import sys
import threading
import time
import zlib

import numpy as np

import parity_util as pu
from solstrale_amd import RenderConfig, _abi, scenes

if __name__ == "__main__":
    spp = int(sys.argv[sys.argv.index("--spp") + 1]) if "--spp" in sys.argv else 512
    which = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "c3"
    lists = [[int(x) for x in a.split(",")] for a in sys.argv[sys.argv.index("--devices") + 1:]] if "--devices" in sys.argv else [[0], [0, 0]]
    sc = {"c3": scenes.sponza_like, "c5": scenes.statue_like, "c2": scenes.cornell_spheres}[which](RenderConfig(1920, 1080, spp))
    lib = _abi.load_hip()
    for devices in lists:
        n = len(devices)
        hs = [C.c_void_p() for _ in devices]
        t0 = time.perf_counter()

        def create(i):
            assert lib.sol_scene_create(sc.desc_ptr, devices[i], C.byref(hs[i])) == _abi.SOL_OK, lib.sol_last_error()
            assert lib.sol_scene_set_partition(hs[i], i, n) == _abi.SOL_OK

        threads = [threading.Thread(target=create, args=(i,)) for i in range(n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        t_create = time.perf_counter() - t0
        arr = (C.c_void_p * n)(*[h.value for h in hs])
        img = C.c_void_p()
        best = 1e9
        for rep in range(4):
            for h in hs:
                lib.sol_clear(h)
            for h in hs:
                lib.sol_sync(h)
            t0 = time.perf_counter()
            for h in hs:
                assert lib.sol_render(h, 0, spp, pu.SEED) == _abi.SOL_OK
            assert lib.sol_gather_local(arr, n, C.byref(img)) == _abi.SOL_OK
            assert lib.sol_sync(hs[0]) == _abi.SOL_OK
            if rep:
                best = min(best, time.perf_counter() - t0)
        out = np.zeros((1080, 1920, 3), np.float32)
        assert lib.sol_read_image(hs[0], out.ctypes.data_as(C.POINTER(C.c_float))) == _abi.SOL_OK
        print(f"{which} 1920x1080x{spp} on devices {devices}: create (parallel) {t_create:.2f} s, step {best * 1e3:.1f} ms = {1920 * 1080 * spp / best / 1e6:.1f} Msamples/s, "
              f"frame crc {zlib.crc32(out.tobytes()):08x}", flush=True)
        for h in hs:
            lib.sol_scene_destroy(h)
