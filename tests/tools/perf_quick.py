"""Quick performance loop for kernel work (not a pytest): times the BASELINE-shaped workloads at reduced spp and prints a
CRC of every image. The image is a pure function of (scene, seed), so a pure-performance change must keep every CRC.
Usage: python tests/tools/perf_quick.py [c1 c2 c3 test] [--spp N] [--phases] [--with-read: also the rate with sol_read (device -> host image) inside the timed region]"""
import _paths  # noqa: F401  (sys.path)
import sys
import time
import zlib

import parity_util as pu
from solstrale_amd import DeviceScene, RenderConfig, scenes


WITH_READ = "--with-read" in sys.argv


def run(name, sc, spp, reps=3, phases=False):
    with DeviceScene(sc) as ds:
        ds.render(0, spp, pu.SEED)
        ds.sync()
        best = 1e9
        for _ in range(reps):
            ds.clear()
            t = time.perf_counter()
            ds.render(0, spp, pu.SEED)
            ds.sync()
            best = min(best, time.perf_counter() - t)
        img = ds.read()
        ns = sc.width * sc.height * spp
        with_read = 1e9
        for _ in range(reps if WITH_READ else 0):
            ds.clear()
            t = time.perf_counter()
            ds.render(0, spp, pu.SEED)
            ds.read()
            with_read = min(with_read, time.perf_counter() - t)
        line = f"{name:10s} {sc.width}x{sc.height}x{spp:<4d} {best * 1e3:9.2f} ms  {ns / best / 1e6:9.1f} Msamples/s  crc {zlib.crc32(img.tobytes()):08x}"
        if WITH_READ:
            line += f"  with sol_read to the host: {with_read * 1e3:9.2f} ms  {ns / with_read / 1e6:9.1f} Msamples/s"
        if phases:
            ds.clear()
            ds.render(0, min(spp, 16), pu.SEED, counted=True)
            st = ds.stats()
            line += f"  rays/s {st['rays'] / min(spp, 16) * spp / best / 1e6:8.1f}M  nodes/ray {st['node_visits'] / st['rays']:.1f}"
            ph = ds.phase_stats() if hasattr(ds, "phase_stats") else None
            if ph:
                line += "  util " + " ".join(f"{k}={v:.2f}" for k, v in ph.items())
        print(line, flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    which = args or ["c1", "c2", "c3", "test"]
    spp = int(sys.argv[sys.argv.index("--spp") + 1]) if "--spp" in sys.argv else 64
    phases = "--phases" in sys.argv
    from solstrale_amd import AlbedoShader, NormalShader, PathTracingShader
    shader = PathTracingShader(50)
    if "--shader" in sys.argv:
        shader = {"normal": NormalShader(), "albedo": AlbedoShader(), "path": PathTracingShader(50),
                  "path1": PathTracingShader(1), "path2": PathTracingShader(2)}[sys.argv[sys.argv.index("--shader") + 1]]
    if "c1" in which:
        run("cornell", scenes.cornell_box(RenderConfig(1920, 1080, spp, shader)), spp, phases=phases)
    if "c2" in which:
        run("spheres", scenes.cornell_spheres(RenderConfig(1920, 1080, spp, shader)), spp, phases=phases)
    if "c3" in which:
        run("sponza", scenes.sponza_like(RenderConfig(1920, 1080, spp, shader)), spp, phases=phases)
    if "c3h" in which:
        run("sponza_het", scenes.sponza_like(RenderConfig(1920, 1080, spp, shader), mesh="heterogeneous"), spp, phases=phases)
    if "c5" in which:
        run("statue", scenes.statue_like(RenderConfig(1920, 1080, spp, shader)), spp, phases=phases)
    if "test" in which:
        run("testscene", scenes.create_test_scene(RenderConfig(800, 400, spp, shader)), spp, phases=phases)
