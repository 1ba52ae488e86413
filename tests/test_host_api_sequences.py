"""The C wrappers of the host mirror (include/solstrale_host.h) take plain ints where the reference's Rust API takes owned values: a texture, material,
transformation or hittable id can be anything a foreign caller passes. Seeded random BUILDER sequences - ids from the valid range, one past it, negative,
huge; NaN / infinite / zero geometry; empty and self-containing BVHs; a constant medium around a medium; images of zero size - must end in an error
string or a scene, never in a crash; whatever solh_finish hands out then goes through sol_scene_create's validation (SOL_EDEVICE without a GPU; with one
the scene is created, rendered and read back) and the host-side tree diagnostics. Also run under AddressSanitizer + UBSan (tests/tools/sanitize.sh)."""
import ctypes as C
import os

import numpy as np
import pytest

from solstrale_amd import _abi

N_SEQUENCES = int(os.environ.get("SOL_TEST_HOST_SEQUENCES", "150"))  # (a longer campaign: SOL_TEST_HOST_SEQUENCES=20000 SOL_TEST_HOST_SEED=k)
SEED_SHIFT = int(os.environ.get("SOL_TEST_HOST_SEED", "0"))
NUMBERS = [0.0, -0.0, 1.0, -1.0, 0.5, 2.0, 10.0, 555.0, 1e-300, 1e300, -1e300, 1e15, float("nan"), float("inf"), float("-inf")]


def d3(v):
    return (C.c_double * 3)(*v)


class Seq:
    def __init__(self, rng):
        self.L = _abi.load_host()
        self.rng = rng
        self.b = self.L.solh_builder_new()
        self.n = {"tex": 0, "mat": 0, "tf": 0, "hit": 0}
        self.errors = 0

    def close(self):
        self.L.solh_builder_free(self.b)

    def num(self, tame=0.7):
        r = self.rng
        return float(r.uniform(-5, 5)) if r.random() < tame else NUMBERS[int(r.integers(len(NUMBERS)))]

    def vec(self):
        return d3([self.num(), self.num(), self.num()])

    def ident(self, kind, allow_none=False):
        """Mostly a valid id of `kind`; sometimes one past the end, -1, -2, or huge."""
        r, n = self.rng, self.n[kind]
        k = int(r.integers(12))
        if k == 0:
            return n
        if k == 1:
            return [-1, -2, -0x80000000, 0x7FFFFFFF, 1 << 20][int(r.integers(5))]
        if allow_none and k == 2:
            return -1
        return int(r.integers(n)) if n else 0

    def note(self, kind, rc, count=1):
        if rc >= 0:
            assert rc >= self.n[kind], (kind, rc, self.n)  # (ids grow: an earlier id is never handed out twice)
            self.n[kind] = rc + count
        else:
            assert self.L.solh_last_error(), kind
            self.errors += 1

    def step(self):
        L, b, r = self.L, self.b, self.rng
        k = int(r.integers(17))
        if k == 0:
            n_ops = int(r.integers(0, 4))
            kinds = (C.c_int * max(1, n_ops))(*[int(r.integers(-1, 6)) for _ in range(max(1, n_ops))])
            params = (C.c_double * (3 * max(1, n_ops)))(*[self.num() for _ in range(3 * max(1, n_ops))])
            self.note("tf", L.solh_transform(b, n_ops, kinds, params))
        elif k == 1:
            self.note("tex", L.solh_solid_color(b, self.num(), self.num(), self.num()))
        elif k == 2:
            w, h = [(4, 4), (1, 1), (0, 4), (4, 0), (0, 0), (7, 3)][int(r.integers(6))]
            a = np.ascontiguousarray(r.integers(0, 256, (max(1, h), max(1, w), 3)), dtype=np.uint8)
            fn = L.solh_image_map if r.integers(2) else L.solh_normal_texture
            self.note("tex", fn(b, w, h, None if r.integers(10) == 0 else a.ctypes.data))
        elif k == 3:
            which = int(r.integers(3))
            if which == 0:
                rc = L.solh_lambertian(b, self.ident("tex"), self.ident("tex", True))
            elif which == 1:
                rc = L.solh_metal(b, self.ident("tex"), self.ident("tex", True), self.num())
            else:
                rc = L.solh_dielectric(b, self.ident("tex"), self.ident("tex", True), self.num())
            self.note("mat", rc)
        elif k == 4:
            self.note("mat", L.solh_diffuse_light(b, self.num(), self.num(), self.num(), self.num(0.3)))
        elif k == 5:
            self.note("mat", L.solh_blend(b, self.ident("mat"), self.ident("mat"), self.num()))
        elif k == 6:
            self.note("hit", L.solh_sphere(b, self.vec(), self.num(), self.ident("mat")))
        elif k == 7:
            self.note("hit", L.solh_quad(b, self.vec(), self.vec(), self.vec(), self.ident("mat"), self.ident("tf", True)))
        elif k == 8:
            self.note("hit", L.solh_box(b, self.vec(), self.vec(), self.ident("mat"), self.ident("tf", True)), 6)
        elif k == 9:
            uv = None if r.integers(2) else (C.c_float * 6)(*[self.num() for _ in range(6)])
            self.note("hit", L.solh_triangle(b, self.vec(), self.vec(), self.vec(), uv, self.ident("mat"), self.ident("tf", True)))
        elif k == 10:
            n = int(r.integers(0, 5))
            v = np.array([self.num() for _ in range(9 * max(1, n))], np.float64)
            m = np.array([self.ident("mat") for _ in range(max(1, n))], np.int32)
            self.note("hit", L.solh_triangles(b, n, v.ctypes.data, None, m.ctypes.data, self.ident("tf", True)), n)
        elif k == 11:
            n = int(r.integers(0, 5))
            c = np.array([self.num() for _ in range(3 * max(1, n))], np.float64)
            rad = np.array([self.num() for _ in range(max(1, n))], np.float64)
            m = np.array([self.ident("mat") for _ in range(max(1, n))], np.int32)
            self.note("hit", L.solh_spheres(b, n, c.ctypes.data, rad.ctypes.data, m.ctypes.data), n)
        elif k == 12:
            self.note("hit", L.solh_constant_medium(b, self.ident("hit"), self.num(), self.vec()))
        elif k in (13, 14):
            n = int(r.integers(0, 7))
            ids = (C.c_int * max(1, n))(*[self.ident("hit") for _ in range(max(1, n))])
            self.note("hit", L.solh_bvh(b, n if r.integers(12) else -1, ids))
        elif k == 15:
            self.note("hit", L.solh_bvh_range(b, self.ident("hit"), int(r.integers(-1, 8))))
        else:
            w, h = [(4, 2), (0, 0), (1, 1), (0, 3)][int(r.integers(4))]
            e = np.ascontiguousarray(r.random((max(1, h), max(1, w), 3)), dtype=np.float32)
            rc = L.solh_environment(b, w, h, e.ctypes.data, self.num())
            if rc < 0:
                self.errors += 1

    def finish_and_check(self):
        L, b, r = self.L, self.b, self.rng
        w, h = [(16, 12), (2, 2), (1, 1), (0, 0), (9, 7)][int(r.integers(5))]
        desc = L.solh_finish(b, self.ident("hit"), w, h, int(r.integers(0, 6)), int(r.integers(0, 60)), self.vec(), self.num(), self.num(0.5), self.vec(), self.vec(), self.vec())
        if not desc:
            assert L.solh_last_error()
            return "error"
        hip = _abi.load_hip()
        hnd = C.c_void_p()
        rc = hip.sol_scene_create(desc, 0, C.byref(hnd))
        assert rc in (_abi.SOL_OK, _abi.SOL_EINVAL, _abi.SOL_ENOLIGHT, _abi.SOL_EDEVICE, _abi.SOL_EDEPTH), rc
        if rc == _abi.SOL_OK:  # (a GPU box: whatever the builder made valid is rendered and read back)
            img = np.zeros((h, w, 3), np.float32)
            assert hip.sol_render(hnd, 0, 2, 99) == _abi.SOL_OK and hip.sol_read(hnd, img.ctypes.data_as(C.POINTER(C.c_float))) == _abi.SOL_OK
            hip.sol_scene_destroy(hnd)
        chk = _abi.SolTreeCheck()
        hip.sol_world_tree_check_ex(desc, int(r.integers(0, 3)), C.byref(chk), C.sizeof(chk))
        return "valid" if rc in (_abi.SOL_OK, _abi.SOL_EDEVICE) else "refused"


def _run_sequences():
    rng = np.random.default_rng(41 + 1000 * SEED_SHIFT)
    tally = {"error": 0, "valid": 0, "refused": 0}
    refused_calls = 0
    for _ in range(N_SEQUENCES):
        s = Seq(rng)
        try:
            for _ in range(int(rng.integers(3, 40))):
                s.step()
            tally[s.finish_and_check()] += 1
            if rng.integers(3) == 0:  # (a builder may go on after a finish)
                for _ in range(int(rng.integers(1, 10))):
                    s.step()
                tally[s.finish_and_check()] += 1
            refused_calls += s.errors
        finally:
            s.close()
    print(tally, "refused builder calls:", refused_calls)
    assert tally["valid"] >= N_SEQUENCES // 20 and refused_calls >= N_SEQUENCES


@pytest.mark.timeout(900)
def test_random_builder_sequences_end_in_an_error_or_a_scene():
    from solstrale_amd import device_count
    if device_count() > 0:
        pytest.skip("the GPU form below does the same and renders")
    _run_sequences()


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_random_builder_sequences_render_on_the_device():
    """The same sequences on a GPU box: every scene the builder and the validation accept is created (device tree build included), rendered and read."""
    _run_sequences()
