"""Python loader of the parity oracle (oracle/_build/liboracle.so). TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product package
(solstrale-rust_amd/) never does. It needs the product's ctypes struct mirror only to pass the flattened scene.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.environ.get("SOLSTRALE_ORACLE_LIB") or os.path.join(HERE, "_build", "liboracle.so")  # env: the sanitizer build (tests/tools/sanitize.sh)
sys.path.insert(0, os.path.join(ROOT, "solstrale-rust_amd"))
from solstrale_amd import _abi  # noqa: E402

ORC_F64, ORC_F32 = 0, 1


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "rays", "node_visits", "sphere_tests", "quad_tests",
                                          "triangle_tests", "shades", "texel_fetches")] + \
               [("threads", C.c_uint32), ("_pad", C.c_uint32), ("live_rays", C.c_uint64)]  # live_rays: the rays the device traces too (oracle.h)

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "_pad"}


_lib = None
_D = C.POINTER(C.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        build()
    lib = C.CDLL(LIB)
    U8 = C.POINTER(C.c_uint8)

    def sig(name, res, args):
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = args

    sig("orc_render", C.c_int, [C.POINTER(_abi.SolSceneDesc), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, _D, C.POINTER(OrcStats)])
    sig("orc_vec3_ops", None, [_D, _D, _D])
    sig("orc_vec3_reflect", None, [_D, _D, _D])
    sig("orc_vec3_refract", None, [_D, _D, C.c_double, _D])
    sig("orc_vec3_unit", None, [_D, _D])
    sig("orc_ray_at", None, [_D, _D, C.c_double, _D])
    sig("orc_aabb_hit", C.c_int, [_D, _D, _D])
    sig("orc_onb_local", None, [_D, _D, _D, _D, _D])
    sig("orc_transform_normal_by_map", None, [_D, _D, _D, _D, _D])
    sig("orc_rgb_to_vec3", None, [U8, _D])
    sig("orc_to_rgb_color", None, [_D, C.c_uint32, U8])
    sig("orc_rng_bits", C.c_uint32, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32])
    sig("orc_f32_funcs", None, [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)])
    sig("orc_eval_f32", C.c_int, [C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32])
    sig("orc_closest_hit", C.c_int, [C.POINTER(_abi.SolSceneDesc), C.c_int, _D, _D, _D, C.POINTER(C.c_uint32)])
    _lib = lib
    return lib


def render(scene, first_sample, n_samples, seed, real=ORC_F32, rect=None, threads=0, out=None):
    """Sums of samples [first, first+n) per pixel; returns (H, W, 3) float64 (row 0 = top) and the counters."""
    lib = load()
    h, w = scene.height, scene.width
    if out is None:
        out = np.zeros((h, w, 3), dtype=np.float64)
    x0, y0, x1, y1 = rect if rect else (0, 0, w, h)
    st = OrcStats()
    rc = lib.orc_render(scene.desc_ptr, real, x0, y0, x1, y1, first_sample, n_samples, seed, threads,
                        out.ctypes.data_as(_D), C.byref(st))
    if rc != 0:
        raise RuntimeError({-1: "orc_render: bad input", -2: "Scene should have at least one light"}.get(rc, str(rc)))
    return out, st.as_dict()


def debug_path(scene, x, y, sample, seed, real=ORC_F32, max_rows=80):
    """Rays of one path: rows of (o xyz, d xyz, t, ref bits, 0, depth, 0, 0); returns (rows, colour)."""
    lib = load()
    lib.orc_debug_path.restype = C.c_int
    lib.orc_debug_path.argtypes = [C.POINTER(_abi.SolSceneDesc), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                   C.c_void_p, C.c_uint32]
    buf = np.zeros((max_rows, 12), dtype=np.float32)
    n = lib.orc_debug_path(scene.desc_ptr, real, x, y, sample, seed, buf.ctypes.data, max_rows)
    return buf[:n], buf[n, :3].copy()


def eval_f32(fn, rows, out_cols):
    """fp32 function table (same row layouts as the device's sol_eval)."""
    lib = load()
    a = np.ascontiguousarray(rows, dtype=np.float32)
    out = np.zeros((a.shape[0], out_cols), dtype=np.float32)
    rc = lib.orc_eval_f32(fn, a.ctypes.data, a.shape[0], a.shape[1], out.ctypes.data, out_cols)
    if rc != 0:
        raise RuntimeError("orc_eval_f32 failed")
    return out


def v3(a):
    return (C.c_double * 3)(*[float(x) for x in a])
