"""Prints the table of the fp32-vs-f64 gate (tests/f64_gate.py): every BASELINE workload at its full scene and resolution, two 128x128 crops
each, the fp32 side next to the oracle's double instantiation (= the reference's arithmetic). Not a pytest.
Usage: python gpu_vs_f64.py [--cpu] [--spp N] [case name ...]
  default: the fp32 side is the DEVICE (sol_render through the C ABI; needs the GPU) - the record kept as profiles/rNN_gpu_vs_f64.txt
  --cpu:   the fp32 side is the oracle's float instantiation (the contract the device is held to bit for bit; runs anywhere)"""
import _paths  # noqa: F401
import sys
import time

import f64_gate as fg
import parity_util as pu


def device_frame(scene, spp, rect):
    from solstrale_amd import DeviceScene
    with DeviceScene(scene) as ds:
        ds.render(0, spp, pu.SEED)
        return ds.read()


def device_window(win, spp):
    from solstrale_amd import DeviceScene
    with DeviceScene(win) as ds:
        ds.render(0, spp, pu.SEED, counted=True)
        st = ds.stats()
    return st["rays"], st["samples"]


if __name__ == "__main__":
    args = sys.argv[1:]
    cpu = "--cpu" in args
    spp = int(args[args.index("--spp") + 1]) if "--spp" in args else fg.SPP
    names = [a for i, a in enumerate(args) if not a.startswith("--") and (i == 0 or args[i - 1] != "--spp")]
    frame, window = (fg.float_oracle_frame, fg.float_oracle_window) if cpu else (device_frame, device_window)
    print(f"fp32 side: {'oracle, float instantiation (CPU)' if cpu else 'DEVICE (HIP path through the C ABI)'}; f64 side: oracle, double instantiation; "
          f"{spp} spp, seed {pu.SEED:#x}, 128x128 crops of the full frame", flush=True)
    print(fg.header(), flush=True)
    for case in fg.CASES:
        if names and case[0] not in names:
            continue
        t0 = time.time()
        sc = fg.make_scene(case, spp)
        for crop, rect in case[4]:
            m = fg.measure(sc, rect, spp, frame, window)
            print(fg.row(case[0], crop, m), flush=True)
        print(f"  ({case[0]}: {time.time() - t0:.1f} s)", file=sys.stderr, flush=True)
