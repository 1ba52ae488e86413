"""Strong-scaling estimate on ONE GPU (not a pytest): renders every rank's tile share of the C3 job (1080p x 512 spp) for world =
1, 2, 4, 8 one after the other and reports max-over-ranks time - what an N-GPU run would take if the gather were free (it moves
3-12 MB per rank once per frame). The real curve comes from the driver's multi-GPU run; this shows the tile balance and the tail.
Both partitions: block b -> rank b mod N, and the balanced table (SOL_OPT_BALANCED_PARTITION: blocks dealt out by their cost in the
creation probe). Usage: python tests/tools/strong_scaling_estimate.py [c3|c4] [spp]"""
import _paths  # noqa: F401
import sys
import time

import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, _abi, scenes

if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    w, h, spp = (1920, 1080, 512) if wl == "c3" else (3840, 2160, 1024)
    if len(sys.argv) > 2:
        spp = int(sys.argv[2])
    sc = scenes.sponza_like(RenderConfig(w, h, spp))
    with DeviceScene(sc) as ds:
      for balanced in (0, 1):
        ds.set_partition(0, 1)
        ds.set_option(_abi.OPT_BALANCED_PARTITION, balanced)
        print("partition:", "balanced table" if balanced else "b mod N", flush=True)
        base = None
        for world in (1, 2, 4, 8):
            times = []
            for rank in range(world):
                ds.set_partition(rank, world)
                mx = ds.max_samples_per_call() // 16 * 16
                best = 1e9
                for _ in range(2):
                    ds.clear()
                    ds.sync()
                    t0 = time.perf_counter()
                    f = 0
                    while f < spp:
                        n = min(spp - f, mx)
                        ds.render(f, n, pu.SEED)
                        f += n
                    ds.sync()
                    best = min(best, time.perf_counter() - t0)
                times.append(best)
            worst = max(times)
            base = base or worst
            print(f"{wl} world {world}: per-rank ms min {min(times) * 1e3:8.2f} max {worst * 1e3:8.2f} (spread {(worst / min(times) - 1) * 100:4.2f} %) -> speed-up {base / worst:5.2f}x "
                  f"(efficiency {base / worst / world:5.3f}), {w * h * spp / worst / 1e6:8.1f} Msamples/s aggregate", flush=True)
