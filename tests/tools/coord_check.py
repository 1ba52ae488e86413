"""One rank of a rendezvous check (tests/test_distributed_cpu.py): bench.py's Coord - the key-value store the ranks use OUTSIDE the
data path (the 128-byte communicator id, the barriers around the timed region, the max-over-ranks time) - over whichever backend the
environment offers: the parent's socket (SOLBENCH_STORE) or the launcher's TCPStore (MASTER_ADDR / MASTER_PORT). No GPU."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

if __name__ == "__main__":
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    c = bench.Coord(rank, world)
    c.barrier()
    uid = c.bcast(bytes(range(128)) if rank == 0 else None)
    assert uid == bytes(range(128)), uid
    time.sleep(0.05 * rank)  # ranks arrive at different times
    t0 = time.perf_counter()
    c.barrier()
    waited = time.perf_counter() - t0
    m = c.max(10.0 + rank)
    assert m == 10.0 + world - 1, m
    for k in range(50):  # many rounds: keys never collide, nobody runs ahead
        assert c.max(float((rank + k) % world)) == float(world - 1)
        c.barrier()
    print(f"rank {rank}/{world} ok waited {waited:.3f}", flush=True)
