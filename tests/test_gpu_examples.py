"""The native example program (solstrale-rust_amd/examples/profiling.cpp = the reference's src/bin/profiling.rs on the device path):
built by build(), no Python or torch inside it - the C++ mirror of Scene / ray_trace over the C ABI. Its picture must be, byte for
byte, the one the Python harness gets for the same scene through the same ABI; without a GPU it must refuse to render."""
import json
import os
import subprocess

import numpy as np
import pytest

import parity_util as pu  # noqa: F401  (sys.path, build)
from solstrale_amd import RenderConfig, _abi, device_count, scenes

EXE = os.path.join(os.path.dirname(_abi.load_hip()._name), "profiling")


def _write_ppm(path, rgb):
    with open(path, "wb") as f:
        f.write(b"P6\n# the reference's resources/textures/tex.jpg, decoded by the harness\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]))
        f.write(np.ascontiguousarray(rgb, dtype=np.uint8).tobytes())


def _read_ppm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P6"
        w, h = (int(x) for x in f.readline().split())
        assert f.readline().strip() == b"255"
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(h, w, 3)


def test_example_is_built_and_has_no_cpu_fallback(tmp_path):
    assert os.access(EXE, os.X_OK), f"{EXE} is missing: build() makes it beside the libraries"
    if device_count() > 0:
        pytest.skip("a GPU is present: the refusal cannot be shown here")
    r = subprocess.run([EXE, "--spp", "16", "--out", str(tmp_path / "o.ppm")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no HIP device" in r.stderr and not (tmp_path / "o.ppm").exists()


def test_example_rejects_bad_input(tmp_path):
    bad = tmp_path / "bad.ppm"
    bad.write_bytes(b"P6\n4 4\n255\nxx")  # truncated
    r = subprocess.run([EXE, "--texture", str(bad)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "truncated" in r.stderr
    r = subprocess.run([EXE, "--no-such-option"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2


@pytest.mark.gpu
def test_native_profiling_program_renders_the_harness_image(tmp_path):
    spp = 80  # (64 + 16: the adaptive batches of ray_trace's OnlyFinal path split the range at multiples of 16)
    tex, out = tmp_path / "tex.ppm", tmp_path / "out.ppm"
    _write_ppm(tex, scenes.load_image("textures/tex.jpg"))
    r = subprocess.run([EXE, "--spp", str(spp), "--texture", str(tex), "--out", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["width"] == 800 and line["height"] == 400 and line["spp"] == spp and line["progress_events"] == spp and line["msamples_per_s"] > 0
    native = _read_ppm(out)
    _, image = scenes.create_test_scene(RenderConfig(800, 400, spp)).ray_trace()
    assert native.shape == image.shape == (400, 800, 3)
    assert (native == image).all(), f"{int((native != image).any(axis=-1).sum())} pixels differ between the native program and the harness"
