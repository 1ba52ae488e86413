"""The standing fp32-vs-f64 gate: the cases and the statistics (tests/test_gpu_vs_f64.py asserts them, tests/tools/gpu_vs_f64.py prints the
table kept under profiles/). TEST INFRASTRUCTURE.

Every other parity test holds the device to the oracle's FLOAT instantiation, which this repository designs together with the device
(the fp32-only rules of include/solstrale_hip.h and DESIGN.md 4): a defect the two share is invisible there - round 4 found BASELINE
config 2 rendering 8.3 % darker than the reference's arithmetic for three rounds that way. Here the fp32 side is put next to the
oracle's DOUBLE instantiation - the reference's own arithmetic (src/hittable/sphere.rs:64-108, triangle.rs:119-173, quad.rs:150-194,
geo/mod.rs:159-188), pinned by the reference's 22 golden images - on crops of every BASELINE workload at its full scene and resolution.

At one seed the two renders follow the same paths except where a rounding decides a branch; a path that rounds apart contributes a
difference of either sign, a RULE that loses or invents energy (or rays) a signed one. Per crop:
  rel        (mean fp32 - mean f64) / mean f64 of the crop
  noise      relative standard error of the crop mean at this sample count, from the pixelwise differences of two independent f64 sample
             sets of the same size (samples [0, spp) and [spp, 2 spp))
  two_sets   the relative difference of those two sets' means (one draw of that noise; the figure profiles/r04_float_vs_double.txt quotes)
  apart      fraction of pixels whose fp32 and f64 values differ by more than 1e-4 + 1e-3 |f64| in some channel (paths that rounded apart)
  z          sum of the pixel differences / sqrt(sum of their squares): the differences' own t statistic; independent zero-mean
             differences give |z| ~ 1, a one-signed offset over n pixels sqrt(n)
  rays       rays per sample of the fp32 side and of f64 on the crop as a WINDOW frame (parity_util.WindowScene), and their ratio - 1. Rays = the
             searches the DEVICE runs (SolStats::rays): it ends a path at a ScatterPdf level whose factor is zero, where the reference traces on
             and multiplies by zero; the oracle counts those apart (OrcStats::live_rays = the device's definition)
"""
import numpy as np

import orc
import parity_util as pu
from solstrale_amd import RenderConfig, scenes

SPP = 64
# (name, factory(render_config), width, height, [(crop name, rect), ...])
CASES = [
    ("c1_cornell", lambda rc: scenes.cornell_box(rc), 400, 400,
     [("tall_box_and_wall", (60, 120, 188, 248)), ("light_and_ceiling", (136, 0, 264, 128))]),
    ("c2_cornell_spheres", lambda rc: scenes.cornell_spheres(rc), 1920, 1080,
     [("dense", (900, 500, 1028, 628)), ("box_edges", (1180, 560, 1308, 688))]),
    ("c3_atrium", lambda rc: scenes.sponza_like(rc), 1920, 1080,
     [("across_hall", (900, 500, 1028, 628)), ("corner", (0, 952, 128, 1080))]),
    ("c3_heterogeneous", lambda rc: scenes.sponza_like(rc, mesh="heterogeneous"), 1920, 1080,
     [("across_hall", (900, 500, 1028, 628)), ("rods_and_rails", (600, 380, 728, 508))]),
    ("c3_heterogeneous_interior", lambda rc: scenes.sponza_like(rc, mesh="heterogeneous", camera="interior"), 1920, 1080,
     [("under_gallery", (1000, 300, 1128, 428)), ("colonnade", (896, 476, 1024, 604))]),
    ("c4_atrium_4k", lambda rc: scenes.sponza_like(rc), 3840, 2160,
     [("across_hall", (1800, 1000, 1928, 1128)), ("corner", (0, 2032, 128, 2160))]),
    ("c5_statue", lambda rc: scenes.statue_like(rc), 1920, 1080,
     [("body_drapery", (896, 476, 1024, 604)), ("glass_head_rim", (900, 60, 1028, 188))]),
    ("c5_statue_hdri", lambda rc: scenes.statue_like(rc, environment=True), 1920, 1080,
     [("body_drapery", (896, 476, 1024, 604)), ("glass_orb", (1150, 860, 1278, 988))]),
    ("profiling_workload", lambda rc: scenes.create_test_scene(rc), 800, 400,
     [("centre", (336, 136, 464, 264)), ("left_objects", (120, 150, 248, 278))]),
    # the regime that hid the sphere defect: a long lens far from the objects (an origin's digits lost against the objects' size)
    ("c3_atrium_far", lambda rc: scenes.sponza_like(rc, camera="far"), 1920, 1080,
     [("roof_opening", (900, 500, 1028, 628)), ("gallery_edge", (900, 250, 1028, 378))]),
    ("c5_statue_far", lambda rc: scenes.statue_like(rc, camera="far"), 1920, 1080,
     [("body_drapery", (896, 476, 1024, 604)), ("glass_head_rim", (900, 60, 1028, 188))]),
]


def float_oracle_frame(scene, spp, rect):
    """The fp32 side on the CPU (tools only): the oracle's float instantiation on the crop."""
    img, _ = orc.render(scene, 0, spp, pu.SEED, real=orc.ORC_F32, rect=rect)
    return img


def float_oracle_window(win, spp):
    _, st = orc.render(win, 0, spp, pu.SEED, real=orc.ORC_F32)
    return st["live_rays"], st["samples"]


def measure(scene, rect, spp, fp32_frame, fp32_window_rays):
    """fp32_frame(scene, spp, rect) -> (H, W, 3) sums of the whole frame (only the crop is read);
    fp32_window_rays(window_scene, spp) -> (rays, samples) of a counted render of the window frame."""
    x0, y0, x1, y1 = rect
    crop = (slice(y0, y1), slice(x0, x1))
    g = np.asarray(fp32_frame(scene, spp, rect), dtype=np.float64)[crop] / spp
    a, _ = orc.render(scene, 0, spp, pu.SEED, real=orc.ORC_F64, rect=rect)
    b, _ = orc.render(scene, spp, spp, pu.SEED, real=orc.ORC_F64, rect=rect)
    a, b = a[crop] / spp, b[crop] / spp
    assert np.isfinite(g).all() and np.isfinite(a).all() and np.isfinite(b).all()
    mean = a.mean()
    n = a.size
    noise = np.sqrt(((a - b) ** 2).sum() / 2.0) / n / mean
    d = (g - a).sum(axis=-1)
    apart = (np.abs(g - a) > 1e-4 + 1e-3 * np.abs(a)).any(axis=-1)
    ss = np.sqrt((d ** 2).sum())
    win = pu.WindowScene(scene, rect)
    rays32, samples32 = fp32_window_rays(win, spp)
    _, st = orc.render(win, 0, spp, pu.SEED, real=orc.ORC_F64)
    r32, r64 = rays32 / samples32, st["live_rays"] / st["samples"]
    return {"mean_f64": float(mean), "rel": float((g.mean() - mean) / mean), "noise": float(noise), "two_sets": float((b.mean() - mean) / mean),
            "apart": float(apart.mean()), "z": float(d.sum() / ss) if ss > 0 else 0.0,
            "rays_fp32": float(r32), "rays_f64": float(r64), "rays_rel": float(r32 / r64 - 1.0)}


def header():
    return (f"{'workload':28s} {'crop':16s} {'mean f64':>9s} {'rel':>10s} {'noise':>9s} {'two f64 sets':>12s} {'apart':>7s} {'z':>6s} "
            f"{'rays fp32':>9s} {'rays f64':>9s} {'rays rel':>9s}")


def row(name, crop, m):
    return (f"{name:28s} {crop:16s} {m['mean_f64']:9.5f} {m['rel']:+10.2e} {m['noise']:9.2e} {m['two_sets']:+12.2e} {m['apart']:7.4f} {m['z']:+6.2f} "
            f"{m['rays_fp32']:9.4f} {m['rays_f64']:9.4f} {m['rays_rel']:+9.2e}")


def make_scene(case, spp=SPP):
    name, factory, w, h, crops = case
    return factory(RenderConfig(w, h, spp))
