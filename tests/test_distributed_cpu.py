"""The N>1 path on CPU: tile ownership, the compact per-rank layout, the gather protocol (torch.distributed, gloo,
world_size 2 and 3) and the inverse permutation. The pixel values come from the oracle, so the test also shows that the
image is a pure function of (scene, seed, pixel, sample): any partition reassembles to the single-rank image."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import orc
import parity_util as pu
from solstrale_amd import RenderConfig, scenes, tiles

W, H, SPP = 45, 27, 3  # not multiples of the 8x8 block


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = scenes.create_test_scene(RenderConfig(W, H, SPP))
        # each rank renders only the pixels it owns (oracle crop per owned block row segment is wasteful: render the rows
        # that contain owned blocks, then keep the owned pixels)
        full, _ = orc.render(sc, 0, SPP, pu.SEED, real=orc.ORC_F32, threads=1)
        local = torch.from_numpy(tiles.compact_from_image(full.astype(np.float32), rank, world))
        assert local.numel() == tiles.accum_floats(W, H, world)
        gathered = tiles.gather_to_rank0(local, world, rank)
        if rank == 0:
            img = tiles.image_from_gathered(gathered.numpy(), W, H, world)
            ret["ok"] = bool((img == full.astype(np.float32)).all())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_reassembles_the_image(world):
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
        assert all(p.exitcode == 0 for p in procs)
        assert ret.get("ok") is True


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("w,h", [(2, 2), (45, 27), (64, 64), (1920, 1080)])
def test_partition_covers_every_pixel_once(world, w, h):
    owner, slot = tiles._slots(w, h, world)
    n = tiles.accum_floats(w, h, world) // 3
    assert slot.max() < n
    for r in range(world):
        s = slot[owner == r]
        assert len(np.unique(s)) == len(s)  # no two pixels of a rank share a slot
    counts = np.bincount(owner.ravel(), minlength=world)
    assert counts.sum() == w * h
    if w * h >= 64 * 64:
        assert counts.max() - counts.min() <= 64 * max(1, (w // 8 + 1))  # round-robin over blocks is balanced


def test_compact_roundtrip():
    rng = np.random.default_rng(0)
    img = rng.random((27, 45, 3)).astype(np.float32)
    for world in (1, 2, 5):
        g = np.concatenate([tiles.compact_from_image(img, r, world) for r in range(world)])
        assert (tiles.image_from_gathered(g, 45, 27, world) == img).all()


# ---- bench.py's rendezvous of the ranks (outside the data path): no process group, no second communicator -------------------
def _run_coord_ranks(world, env_of):
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "coord_check.py")
    procs = [subprocess.Popen([sys.executable, script], env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), **env_of(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0 and f"rank {r}/{world} ok" in outs[r], outs[r][-1500:]


@pytest.mark.parametrize("world", [2, 3])
def test_bench_rendezvous_through_the_parents_socket(world):
    """`python bench.py --gpus N`: the parent binds the store's socket before it starts the ranks and serves set / get / add."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    server = bench._StoreServer()
    server.start()
    _run_coord_ranks(world, lambda r: {"SOLBENCH_STORE": f"127.0.0.1:{server.port}"})
    server.sock.close()


def test_bench_rendezvous_through_a_tcpstore():
    """Under torch.distributed.run the ranks find MASTER_ADDR / MASTER_PORT: a c10d TCPStore (rank 0 serves it unless the launcher's
    agent already does), still no process group."""
    port = _free_port()
    env = {k: v for k, v in os.environ.items() if k not in ("SOLBENCH_STORE", "TORCHELASTIC_USE_AGENT_STORE")}
    os_env = dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    saved = dict(os.environ)
    try:
        os.environ.clear()
        os.environ.update(env)
        _run_coord_ranks(2, lambda r: os_env)
    finally:
        os.environ.clear()
        os.environ.update(saved)


def test_bench_rendezvous_under_torch_distributed_run():
    """The driver's launch line (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P):
    the agent serves the store on MASTER_PORT, the ranks join it as clients."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "coord_check.py")
    env = {k: v for k, v in os.environ.items() if k not in ("SOLBENCH_STORE", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), script], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rank 0/2 ok" in r.stdout and "rank 1/2 ok" in r.stdout, (r.stdout + r.stderr)[-2000:]
