"""The standing gate that is NOT common-mode: the HIP fp32 frame next to the oracle's DOUBLE instantiation - the reference's own arithmetic
(src/hittable/sphere.rs:64-108, triangle.rs:119-173, quad.rs:150-194, geo/mod.rs:159-188; pinned by the reference's 22 golden images) -
on two 128x128 crops of every BASELINE workload at its full scene and resolution: C1, C2, C3, the heterogeneous stress mesh (both cameras),
C4 at 4K, C5, C5 + HDRI, the reference's profiling workload, and far-camera variants of C3 and C5 (the regime that hid the fp32 sphere
defect of rounds 1-3: BASELINE config 2 rendered 8.3 % darker than f64 while every fp32-vs-fp32 parity test was green).

Statistics and cases: tests/f64_gate.py. Per crop, at 64 spp and the seed of every parity test:
  (i)   rays per sample of the device within 0.3 % of f64's (crop rendered as a window frame by both sides);
  (ii)  |relative difference of the crop means| below the case's `rel_of_noise` x the measured noise of the crop mean (two independent f64
        sample sets of the same size), and the pixel differences' own t statistic |z| < 4.5 - a rule that loses or invents energy gives
        the differences one sign (the sphere defect: z ~ -37 on C2), paths that merely round apart give |z| ~ 1;
  (iii) the fraction of pixels whose paths rounded apart below the case's ceiling.
Measured figures (MI355X): profiles/r05_gpu_vs_f64.txt (tests/tools/gpu_vs_f64.py prints the table).
"""
import pytest

import f64_gate as fg
import parity_util as pu
from solstrale_amd import DeviceScene

pytestmark = pytest.mark.gpu

RAYS_REL = 3e-3
Z_MAX = 4.5
# case -> (rel_of_noise, ceiling of `apart`). Measured |rel| / noise is at most 0.33 (c5_statue_far, glass rim) and 0.14 elsewhere;
# `apart` at 64 spp: C2 0.27 (12 rays per sample through 10 000 spheres seen from 800 units), atrium 0.02 - 0.07, statue 0.002 - 0.044.
BOUNDS = {
    "c1_cornell": (0.6, 0.02),
    "c2_cornell_spheres": (0.6, 0.40),
    "c3_atrium": (0.6, 0.08),
    "c3_heterogeneous": (0.6, 0.08),
    "c3_heterogeneous_interior": (0.6, 0.12),
    "c4_atrium_4k": (0.6, 0.08),
    "c5_statue": (0.6, 0.02),
    "c5_statue_hdri": (0.6, 0.02),
    "profiling_workload": (0.6, 0.01),
    "c3_atrium_far": (0.6, 0.10),
    "c5_statue_far": (0.6, 0.08),
}


def device_frame(scene, spp, rect):
    with DeviceScene(scene) as ds:
        ds.render(0, spp, pu.SEED)
        return ds.read()


def device_window(win, spp):
    with DeviceScene(win) as ds:
        ds.render(0, spp, pu.SEED, counted=True)
        st = ds.stats()
    assert st["samples"] == win.width * win.height * spp
    return st["rays"], st["samples"]


@pytest.mark.parametrize("case", fg.CASES, ids=[c[0] for c in fg.CASES])
def test_device_follows_the_reference_arithmetic(case):
    name = case[0]
    rel_of_noise, apart_max = BOUNDS[name]
    sc = fg.make_scene(case)
    frames = {}

    def frame(scene, spp, rect):  # one device render of the full frame serves both crops
        if "f" not in frames:
            frames["f"] = device_frame(scene, spp, rect)
        return frames["f"]

    for crop, rect in case[4]:
        m = fg.measure(sc, rect, fg.SPP, frame, device_window)
        print(fg.row(name, crop, m))
        assert m["mean_f64"] > 0
        assert abs(m["rays_rel"]) <= RAYS_REL, (name, crop, m)
        assert abs(m["rel"]) <= rel_of_noise * m["noise"], (name, crop, m)
        assert abs(m["z"]) < Z_MAX, (name, crop, m)
        assert m["apart"] <= apart_max, (name, crop, m)


def test_the_gate_sees_a_shared_defect():
    """The gate's own sensitivity: a fp32 side that loses 0.5 % of its energy - a sixteenth of what the sphere defect lost - fails (ii),
    although it would pass any fp32-vs-fp32 comparison with an oracle that shares the loss."""
    import numpy as np
    case = [c for c in fg.CASES if c[0] == "profiling_workload"][0]
    sc = fg.make_scene(case)
    rect = case[4][0][1]

    def dimmed(scene, spp, r):
        img = device_frame(scene, spp, r).astype(np.float64)
        return img * 0.995

    m = fg.measure(sc, rect, fg.SPP, dimmed, device_window)
    assert abs(m["z"]) > Z_MAX or abs(m["rel"]) > 0.6 * m["noise"], m
