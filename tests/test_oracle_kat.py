"""Pins the oracle's building blocks against the reference's exact known-answer tests (doc-tests and #[test]s that
assert exact values). Each case cites the reference assertion it restates."""
import ctypes as C

import numpy as np

import orc

LIB = orc.load()
D3 = C.c_double * 3


def _v(*a):
    return D3(*[float(x) for x in a])


def _call3(fn, *args):
    out = D3()
    fn(*args, out)
    return tuple(out)


def test_vec3_operators():
    out = (C.c_double * 16)()
    LIB.orc_vec3_ops(_v(1, 2, 3), _v(4, 5, 6), out)
    assert tuple(out[0:3]) == (5., 7., 9.)      # src/geo/vec3.rs:51-52   Add
    assert tuple(out[3:6]) == (-3., -3., -3.)   # Sub (vec3.rs:107-108 uses (1,2,3)-(6,5,4); same operator)
    assert tuple(out[6:9]) == (4., 10., 18.)    # vec3.rs:145-146  Mul<Vec3>
    assert out[9] == 32.                        # vec3.rs:253-254  dot
    LIB.orc_vec3_ops(_v(2, 3, 4), _v(5, 6, 7), out)
    assert tuple(out[10:13]) == (-3., 6., -3.)  # vec3.rs:264-265  cross
    LIB.orc_vec3_ops(_v(1, 2, 3), _v(0, 0, 0), out)
    assert out[13] == 14.                       # vec3.rs:280      length_squared
    LIB.orc_vec3_ops(_v(0, 3, 4), _v(0, 0, 0), out)
    assert out[14] == 5.                        # vec3.rs:291      length
    LIB.orc_vec3_ops(_v(1, 2, 3), _v(6, 5, 4), out)
    assert tuple(out[3:6]) == (-5., -3., -1.)   # vec3.rs:107-108


def test_vec3_unit():
    u = _call3(LIB.orc_vec3_unit, _v(1, 2, 3))
    assert abs(np.linalg.norm(u) - 1.) < 1e-8 and np.dot(u, (1, 2, 3)) > 0  # vec3.rs:303-304


def test_vec3_reflect():
    assert _call3(LIB.orc_vec3_reflect, _v(0, 3, 4), _v(0, 1, 0)) == (0., -3., 4.)  # vec3.rs:327
    assert _call3(LIB.orc_vec3_reflect, _v(0, 3, 4), _v(0, 0, 1)) == (0., 3., -4.)  # vec3.rs:328


def test_vec3_refract_ior_one_is_identity():
    v = _call3(LIB.orc_vec3_unit, _v(-3, -3, 0))
    out = D3()
    LIB.orc_vec3_refract(_v(*v), _v(0, 1, 0), 1.0, out)
    assert np.abs(np.array(tuple(out)) - np.array(v)).max() < 1e-8  # vec3.rs:338-340 near_zero


def test_ray_at():
    d = _call3(LIB.orc_vec3_unit, _v(4, 5, 6))
    o = (1., 2., 3.)
    out = D3()
    LIB.orc_ray_at(_v(*o), _v(*d), 0.0, out)
    assert tuple(out) == o  # src/geo/mod.rs:319
    l = float(np.linalg.norm(d))
    LIB.orc_ray_at(_v(*o), _v(*d), l, out)
    assert np.abs(np.array(tuple(out)) - np.array(o) - np.array(d)).max() < 1e-8  # mod.rs:320
    LIB.orc_ray_at(_v(*o), _v(*d), -l, out)
    assert np.abs(np.array(tuple(out)) - np.array(o) + np.array(d)).max() < 1e-8  # mod.rs:321


def test_aabb_hit_semantics():
    """Aabb::hit (src/geo/mod.rs:159-188): forward slab over [0, inf), strict t_min < t_max, NaN ignored by max/min."""
    box = (C.c_double * 6)(-1, 1, -2, 2, -3, 3)
    assert LIB.orc_aabb_hit(box, _v(0, 0, -10), _v(0, 0, 1)) == 1
    assert LIB.orc_aabb_hit(box, _v(0, 0, -10), _v(0, 0, -1)) == 0   # behind the origin
    assert LIB.orc_aabb_hit(box, _v(5, 0, -10), _v(0, 0, 1)) == 0    # misses in x (1/0 = inf slabs)
    assert LIB.orc_aabb_hit(box, _v(0, 0, 0), _v(1, 1, 1)) == 1      # origin inside
    assert LIB.orc_aabb_hit(box, _v(1, 0, -10), _v(0, 0, 1)) == 1    # on a face plane: (1-1)*inf = NaN is ignored
    flat = (C.c_double * 6)(-1, 1, 0, 0, -1, 1)                      # zero thickness: t_min == t_max -> no hit
    assert LIB.orc_aabb_hit(flat, _v(0, 5, 0), _v(0, -1, 0)) == 0


def test_transform_normal_by_map():
    # src/material/mod.rs:456-469
    n = _call3(LIB.orc_transform_normal_by_map, _v(1., .5, .5), _v(0, 1, 0), _v(0, 0, 1), _v(1, 0, 0))
    assert np.abs(np.array(n) - np.array((0., 1., 0.))).max() < 1e-8


def test_rgb_to_vec3():
    out = D3()
    LIB.orc_rgb_to_vec3((C.c_uint8 * 3)(0, 100, 255), out)
    assert tuple(out) == (0., 0.39215686274509803, 1.)  # src/util/rgb_color.rs:49-54


def test_to_rgb_color():
    out = (C.c_uint8 * 3)()
    LIB.orc_to_rgb_color(_v(0., .3, 1.), 1, out)
    assert tuple(out) == (0, 140, 255)  # src/util/rgb_color.rs:58
    LIB.orc_to_rgb_color(_v(0., .3, 1.), 2, out)
    assert tuple(out) == (0, 99, 181)   # src/util/rgb_color.rs:59


def test_rng_is_a_pure_function_with_unit_range():
    # range tests of src/random.rs:27-59 carried over to the counter generator; plus determinism and key sensitivity
    a = [LIB.orc_rng_bits(0x5017A1E, 17, 3, c) for c in range(64)]
    b = [LIB.orc_rng_bits(0x5017A1E, 17, 3, c) for c in range(64)]
    assert a == b and len(set(a)) == 64
    assert [LIB.orc_rng_bits(0x5017A1E, 18, 3, c) for c in range(64)] != a
    assert [LIB.orc_rng_bits(0x5017A1F, 17, 3, c) for c in range(64)] != a
    assert [LIB.orc_rng_bits(0x5017A1E, 17, 4, c) for c in range(64)] != a
    u = np.array([LIB.orc_rng_bits(1, p, s, c) >> 8 for p in range(20) for s in range(20) for c in range(20)]) / 2 ** 24
    assert (u >= 0).all() and (u < 1).all()
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005


def test_fp32_elementary_functions_are_accurate():
    out = (C.c_float * 5)()
    rs = np.linspace(0, 1, 4001, endpoint=False)
    worst = 0.0
    for r in rs:
        x = 2 * r - 1
        LIB.orc_f32_funcs(r, x, 0.37, out)
        r32 = np.float32(r).astype(np.float64)
        x32 = np.float32(x).astype(np.float64)
        worst = max(worst, abs(out[0] - np.cos(2 * np.pi * r32)), abs(out[1] - np.sin(2 * np.pi * r32)),
                    abs(out[2] - np.arccos(x32)), abs(out[3] - np.arctan2(np.float32(0.37), x32)))
        if r32 > 0:
            worst = max(worst, abs(out[4] - np.log(r32)) / max(1.0, abs(np.log(r32))))
    assert worst < 2e-6
