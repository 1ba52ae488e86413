"""Function-level parity: device functions (through the C ABI's sol_eval) vs the fp32 CPU restatement, BIT FOR BIT.

This pins the "fp32 arithmetic contract" of DESIGN.md: the same IEEE operation sequence on both sides, for the arithmetic
primitives (correctly rounded / and sqrt, unfused a*b+c), the spec'd elementary functions, the counter RNG, the vector
helpers of src/geo/vec3.rs, the primitive hit tests of src/hittable/{sphere,quad,triangle}.rs and Aabb::hit.
"""
import numpy as np
import pytest

import orc
from solstrale_amd.device import eval_functions

pytestmark = pytest.mark.gpu
N = 200_000


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _same(gpu, cpu, what):
    g, c = _bits(gpu), _bits(cpu)
    both_nan = np.isnan(gpu) & np.isnan(cpu)
    both_zero = (gpu == 0) & (cpu == 0)  # the sign of an exact zero is not part of the contract (it decides nothing)
    diff = (g != c) & ~both_nan & ~both_zero
    rows = np.nonzero(diff.any(axis=1))[0][:3]
    assert not diff.any(), (f"{what}: {int(diff.sum())} of {diff.size} values differ; per column {diff.sum(axis=0).tolist()}; "
                            f"first rows {rows.tolist()} gpu {gpu[rows].tolist()} cpu {cpu[rows].tolist()}")


def _rng(seed):
    return np.random.default_rng(seed)


def test_arithmetic_is_ieee():
    r = _rng(1)
    x = np.concatenate([r.standard_normal((N, 3)) * 10.0 ** r.integers(-6, 6, (N, 1)),
                        np.array([[0., 1., 2.], [1., 0., 3.], [-0., 5., 1.], [3., 3., -9.], [1e-30, 1e20, 0.], [1e30, 1e-20, 1.]])])
    x = x.astype(np.float32)
    _same(eval_functions(0, x, 7), orc.eval_f32(0, x, 7), "arithmetic")


def test_elementary_functions():
    r = _rng(2)
    u = (r.integers(0, 1 << 24, N) / 16777216.0)
    x = np.stack([u, r.uniform(-1, 1, N), r.uniform(-1, 1, N)], 1).astype(np.float32)
    x[:4, 0] = [0.0, 0.25, 0.5, 0.999999940395]
    x[:4, 1] = [1.0, -1.0, 0.0, 0.5]
    g, c = eval_functions(1, x, 5), orc.eval_f32(1, x, 5)
    _same(g, c, "elementary")
    # and they are accurate: within 2e-7 absolute of libm in double
    phi = 2 * np.pi * x[:, 0].astype(np.float64)
    assert np.abs(g[:, 0] - np.cos(phi)).max() < 3e-7 and np.abs(g[:, 1] - np.sin(phi)).max() < 3e-7
    assert np.abs(g[:, 2] - np.arccos(x[:, 1].astype(np.float64))).max() < 2e-6
    assert np.abs(g[:, 3] - np.arctan2(x[:, 2].astype(np.float64), x[:, 1].astype(np.float64))).max() < 2e-6
    pos = x[:, 0] > 0
    assert np.abs(g[pos, 4] - np.log(x[pos, 0].astype(np.float64))).max() < 2e-6


def test_rng_bits():
    r = _rng(3)
    x = r.integers(0, 1 << 32, (N, 5), dtype=np.uint64).astype(np.uint32).view(np.float32)
    g, c = eval_functions(2, x, 2), orc.eval_f32(2, x, 2)
    assert (_bits(g) == _bits(c)).all()
    assert (g[:, 1] >= 0).all() and (g[:, 1] < 1).all()


def test_vector_helpers():
    r = _rng(4)
    v = r.standard_normal((N, 3))
    n = r.standard_normal((N, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    x = np.concatenate([v, n, r.uniform(0.5, 2.0, (N, 1))], 1).astype(np.float32)
    _same(eval_functions(3, x, 18), orc.eval_f32(3, x, 18), "vector helpers")


def test_sphere_hit():
    r = _rng(5)
    c = r.uniform(-50, 50, (N, 3))
    rad = r.uniform(0.5, 10, (N, 1))
    o = r.uniform(-60, 60, (N, 3))
    d = (c + r.standard_normal((N, 3)) * rad * 0.8) - o  # mostly hitting
    d *= r.uniform(0.1, 30, (N, 1))
    # column 12: slack of the hit-point-in-own-box rule (fp32 contract)
    x = np.concatenate([c, rad, o, d, np.full((N, 1), 0.001), np.full((N, 1), np.inf), np.full((N, 1), 4e-4)], 1).astype(np.float32)
    o2 = x.copy()
    o2[: N // 4, 4:7] = (c + rad * (d / np.linalg.norm(d, axis=1, keepdims=True)))[: N // 4]  # origins on the surface
    far = x.copy()  # distant origins with long directions: the regime where the fp32 quadratic loses its digits
    far[:, 4:7] = (c - d / np.linalg.norm(d, axis=1, keepdims=True) * 900.0).astype(np.float32)
    far[:, 7:10] = (d / np.linalg.norm(d, axis=1, keepdims=True) * 800.0 + r.standard_normal((N, 3)) * rad * 0.9).astype(np.float32)
    for rows in (x, o2, far):
        g, cc = eval_functions(4, rows, 2), orc.eval_f32(4, rows, 2)
        _same(g, cc, "sphere hit")
        assert g[:, 0].mean() > 0.2


def test_quad_hit():
    r = _rng(6)
    q = r.uniform(-500, 500, (N, 3))
    u = r.standard_normal((N, 3)) * 100
    v = np.cross(u, r.standard_normal((N, 3)))
    nvec = np.cross(u, v)
    nn = nvec / np.linalg.norm(nvec, axis=1, keepdims=True)
    dd = (nn * q).sum(1, keepdims=True)
    w = nvec / (nvec * nvec).sum(1, keepdims=True)
    o = r.uniform(-800, 800, (N, 3))
    target = q + u * r.uniform(-0.2, 1.2, (N, 1)) + v * r.uniform(-0.2, 1.2, (N, 1))
    d = (target - o) * r.uniform(0.2, 3, (N, 1))
    x = np.concatenate([nn, dd, q, w, u, v, o, d, np.full((N, 1), 0.001), np.full((N, 1), np.inf)], 1).astype(np.float32)
    g, c = eval_functions(5, x, 4), orc.eval_f32(5, x, 4)
    _same(g, c, "quad hit")
    assert 0.3 < g[:, 0].mean() < 0.9


def test_triangle_hit():
    r = _rng(7)
    v0 = r.uniform(-20, 20, (N, 3))
    e1 = r.standard_normal((N, 3)) * r.uniform(0.01, 3, (N, 1))
    e2 = r.standard_normal((N, 3)) * r.uniform(0.01, 3, (N, 1))
    o = r.uniform(-30, 30, (N, 3))
    a, b = r.uniform(-0.2, 1.0, (N, 1)), r.uniform(-0.2, 1.0, (N, 1))
    d = (v0 + e1 * a + e2 * b - o) * r.uniform(0.2, 3, (N, 1))
    x = np.concatenate([v0, e1, e2, o, d, np.full((N, 1), 0.001), np.full((N, 1), np.inf)], 1).astype(np.float32)
    # exact edge / vertex hits: target points on the edges
    x[: N // 10, 12:15] = ((v0 + e1 * a) - o)[: N // 10].astype(np.float32)
    g, c = eval_functions(6, x, 4), orc.eval_f32(6, x, 4)
    _same(g, c, "triangle hit")
    assert 0.2 < g[:, 0].mean() < 0.8


def test_aabb_hit():
    r = _rng(8)
    lo = r.uniform(-100, 100, (N, 3))
    hi = lo + r.uniform(0, 30, (N, 3)) * (r.uniform(0, 1, (N, 3)) > 0.2)  # some flat boxes
    box = np.stack([lo[:, 0], hi[:, 0], lo[:, 1], hi[:, 1], lo[:, 2], hi[:, 2]], 1)
    o = r.uniform(-150, 150, (N, 3))
    d = (lo + (hi - lo) * r.uniform(-0.3, 1.3, (N, 3))) - o
    d[: N // 10, 0] = 0.0  # axis-parallel rays: 1/0 = inf, (b - o) * inf, NaN handling of fmax/fmin
    d[N // 10: N // 5, 1] = -0.0
    x = np.concatenate([box, o, d], 1).astype(np.float32)
    g, c = eval_functions(7, x, 2), orc.eval_f32(7, x, 2)
    assert (g[:, 0] == c[:, 0]).all(), f"{int((g[:, 0] != c[:, 0]).sum())} slab decisions differ"
    assert 0.2 < g[:, 0].mean() < 0.95


def test_sampling_streams():
    r = _rng(9)
    x = r.integers(0, 1 << 32, (N, 4), dtype=np.uint64).astype(np.uint32).view(np.float32)
    g, c = eval_functions(8, x, 7), orc.eval_f32(8, x, 7)
    _same(g, c, "cosine / unit-sphere sampling")
