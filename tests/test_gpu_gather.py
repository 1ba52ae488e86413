"""sol_gather with world > 1, executed: N rank processes on the one GPU of the box go through the C ABI's collective sequence
(sol_comm_unique_id -> sol_comm_init -> sol_render -> sol_gather -> sol_read_image, include/solstrale_hip.h) and rank 0's frame must
equal the single-rank frame bit for bit. RCCL refuses two ranks on one device, so the rank processes find a TEST-ONLY transport
stub (tests/stub_rccl: the eight ncclXxx entry points the product binds, over unix sockets) in front of the real librccl.so.1 on
their LD_LIBRARY_PATH; the product library is the unmodified one. No reference analogue: the reference's only parallelism is
Rayon rows inside one process (src/renderer/mod.rs:232-291)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
STUB_DIR = os.path.join(HERE, "stub_rccl")
RCCL_SYMBOLS = ["ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv",
                "ncclGetErrorString"]


def build_stub():
    subprocess.check_call(["make", "-s", "-C", STUB_DIR])
    return os.path.join(STUB_DIR, "_build")


def test_stub_exports_what_the_product_binds():
    """(CPU) The stub must offer every entry point sol_comm.cpp looks up - and carry the marker the rank processes check."""
    lib = os.path.join(build_stub(), "librccl.so.1")
    names = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    for sym in RCCL_SYMBOLS + ["sol_stub_rccl_marker"]:
        assert f" {sym}\n" in names, sym
    src = open(os.path.join(HERE, "..", "solstrale-rust_amd", "csrc", "sol_comm.cpp")).read()
    for sym in RCCL_SYMBOLS:
        assert f'sym("{sym}")' in src, sym  # the list above IS what the product binds


def run_ranks(world, scene, w, h, spp, balanced=False):
    stub = build_stub()
    with tempfile.TemporaryDirectory(prefix="solgather_") as rdv:
        env = dict(os.environ, LD_LIBRARY_PATH=stub + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), TMPDIR=rdv)
        cmd = [sys.executable, os.path.join(HERE, "tools", "gather_rank.py")]
        procs = [subprocess.Popen(cmd + [str(r), str(world), rdv, scene, str(w), str(h), str(spp)] + (["balanced"] if balanced else []), env=env, stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(world)]
        outs = []
        try:
            for p in procs:
                outs.append(p.communicate(timeout=300)[0])
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        for r, p in enumerate(procs):
            assert p.returncode == 0, (r, outs[r][-2000:])
        return [np.load(os.path.join(rdv, f"frame{k}.npy")) for k in range(2)]


@pytest.mark.gpu
@pytest.mark.parametrize("balanced", [False, True], ids=["b_mod_n", "balanced_table"])
@pytest.mark.parametrize("world,scene,w,h", [(2, "c3", 256, 128), (3, "c3", 256, 128), (2, "test", 260, 131), (3, "c2", 250, 131), (5, "c1", 97, 61)],
                         ids=["2_ranks", "3_ranks", "2_ranks_odd_block_count_medium", "3_ranks_edge_blocks", "5_ranks_ragged"])
def test_gather_through_the_abi_equals_the_single_rank_frame(world, scene, w, h, balanced):
    """Block counts: 256x128 = 512 blocks (2 | 512, 3 does not divide it); 260x131 -> 33 x 17 = 561 (odd); 250x131 -> 32 x 17 =
    544 = 3 * 181 + 1 with padding pixels on two edges; 97x61 -> 13 x 8 = 104 = 5 * 20 + 4. Ranks whose compact buffer has fewer
    blocks than rank 0's still send rank 0's size (equal counts, sol_scene_set_partition): the receive offsets must match. `balanced`:
    the same with SOL_OPT_BALANCED_PARTITION (blocks dealt out by their cost in the creation probe; the table behind the partition)."""
    spp = 16
    make = {"c1": scenes.cornell_box, "c2": scenes.cornell_spheres, "c3": scenes.sponza_like, "test": scenes.create_test_scene}[scene]
    with DeviceScene(make(RenderConfig(w, h, spp))) as ds:
        ds.render(0, spp, pu.SEED)
        want = ds.read()
    frames = run_ranks(world, scene, w, h, spp, balanced)
    for k, got in enumerate(frames):
        assert got.shape == want.shape and (got == want).all(), (k, int((got != want).sum()))
