"""One rank of an N-rank gather THROUGH THE C ABI on a box with one GPU (tests/test_gpu_gather.py starts N of these):
sol_comm_unique_id (rank 0; shipped through a file) -> sol_comm_init -> sol_render -> sol_gather -> rank 0: sol_read_image.
Every rank uses device 0; the process must find tests/stub_rccl/_build/librccl.so.1 first on LD_LIBRARY_PATH (RCCL itself refuses
two ranks on one device) and must not load torch (its RCCL would be the one the product's dlopen hands back).
Usage: python gather_rank.py <rank> <world> <rendezvous dir> <scene> <width> <height> <spp> [balanced]"""
import _paths  # noqa: F401  (sys.path)
import ctypes
import os
import sys
import time

import numpy as np

import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, comm_unique_id, scenes


def wait_for(path, timeout=120.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise SystemExit(f"rank timed out waiting for {path}")
        time.sleep(0.02)


if __name__ == "__main__":
    rank, world, rdv, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    w, h, spp = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    assert "torch" not in sys.modules
    make = {"c1": scenes.cornell_box, "c2": scenes.cornell_spheres, "c3": scenes.sponza_like, "test": scenes.create_test_scene}[name]
    sc = make(RenderConfig(w, h, spp))
    id_file = os.path.join(rdv, "unique_id")
    if rank == 0:
        uid = comm_unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(bytes(uid))
        os.rename(id_file + ".tmp", id_file)
    else:
        wait_for(id_file)
        uid = open(id_file, "rb").read()
    rccl = ctypes.CDLL("librccl.so.1")  # already loaded by sol_comm_unique_id on rank 0; the same object either way
    if not hasattr(rccl, "sol_stub_rccl_marker"):
        raise SystemExit("the RCCL in this process is not the test stub: LD_LIBRARY_PATH?")
    with DeviceScene(sc, 0) as ds:
        if len(sys.argv) > 8 and sys.argv[8] == "balanced":
            from solstrale_amd import _abi
            ds.set_option(_abi.OPT_BALANCED_PARTITION, 1)  # (every rank, before sol_comm_init)
        ds.comm_init(rank, world, uid)
        for rep in range(2):  # twice: the second gather reuses rank 0's receive buffer
            ds.clear()
            ds.render(0, spp, pu.SEED)
            ds.gather(0)  # into the scene's own image buffer
            if rank == 0:
                img = ds.read_image()
                np.save(os.path.join(rdv, f"frame{rep}.npy"), img)
        ds.sync()
        open(os.path.join(rdv, f"done{rank}"), "w").close()
        for r in range(world):  # nobody tears its sockets down before everyone is through
            wait_for(os.path.join(rdv, f"done{r}"))
