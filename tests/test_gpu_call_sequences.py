"""Random call sequences against a live handle (the companion of tests/test_desc_mutations.py on the other side of sol_scene_create): every
entry point that takes a `SolScene*` is called in seeded random order with arguments from pools that hold the legal values, the edges
(0, 15 / 16 / 17 samples, rank = world - 1, more ranks than blocks) and the illegal ones (null pointers, rank >= world, unknown options, a
NaN bloom kernel, sample ranges that wrap 32 bits, struct sizes of 0 and 2^32 - 1). Contract (SURVEY 8b "Errors"): every call
returns a code - never an abort, a fault or a hang - and whatever the sequence did, the handle afterwards renders the frame a fresh handle
renders, bit for bit (partition, options, auxiliary planes and post-processing leave nothing behind that a clear does not remove).
Buffers the caller must size are sized as the header says (a too-small caller buffer is a lie no callee can detect)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from solstrale_amd import RenderConfig, _abi, scenes

pytestmark = pytest.mark.gpu

N_CALLS = int(os.environ.get("SOL_TEST_CALLS", "1500"))  # (a longer campaign: SOL_TEST_CALLS=30000 SOL_TEST_CALL_SEED=k)
SEED_SHIFT = int(os.environ.get("SOL_TEST_CALL_SEED", "0"))
TRACE = bool(os.environ.get("SOL_TEST_CALL_TRACE"))
W, H = 104, 67  # (ragged: 13 x 9 blocks, the last row and column partly outside the image)
FIRST = [0, 0, 0, 1, 15, 16, 17, 48, 4096, 0xFFFFFFF0, 0xFFFFFFFF]
COUNT = [0, 1, 5, 15, 16, 16, 17, 33, 64, 100, 0xFFFFFFF0, 0xFFFFFFFF]
PARTITION = [(0, 1), (0, 1), (0, 2), (1, 2), (2, 3), (7, 8), (63, 64), (116, 117), (117, 118), (500, 1000), (0, 0x7FFFFFFF), (0x7FFFFFFE, 0x7FFFFFFF),
             (0, 0), (-1, 2), (2, 2), (3, 2), (0, -1), (-0x80000000, -0x80000000)]
F64 = [float("nan"), float("inf"), float("-inf"), -1.0, 0.0, 1e-300, 0.01, 0.1, 0.25, 0.5, 0.5000001, 1.0, 10.0, 1e300]
SIZES = [0, 4, 7, 8, 16, 48, 64, 4096, 4097, 0xFFFFFFFF]


class Handle:
    def __init__(self, sc):
        self.lib = _abi.load_hip()
        self.sc = sc
        self.h = C.c_void_p()
        assert self.lib.sol_scene_create(sc.desc_ptr, 0, C.byref(self.h)) == _abi.SOL_OK, self.lib.sol_last_error()
        self.img = np.zeros((H, W, 3), np.float32)
        self.img2 = np.zeros((H, W, 3), np.float32)
        self.rgb8 = np.zeros((H, W, 3), np.uint8)
        self.rows = np.zeros((64, 12), np.float32)

    def close(self):
        self.lib.sol_scene_destroy(self.h)

    def fp(self, a):
        return a.ctypes.data_as(C.POINTER(C.c_float))

    def frame(self, spp=16, seed=77):
        L, h = self.lib, self.h
        assert L.sol_scene_bind_accum(h, None, 0) == _abi.SOL_OK
        assert L.sol_scene_set_stream(h, None) == _abi.SOL_OK
        assert L.sol_scene_set_option(h, _abi.OPT_BALANCED_PARTITION, 0) == _abi.SOL_OK
        assert L.sol_scene_set_partition(h, 0, 1) == _abi.SOL_OK, L.sol_last_error()
        assert L.sol_clear(h) == _abi.SOL_OK
        assert L.sol_render(h, 0, spp, seed) == _abi.SOL_OK, L.sol_last_error()
        out = np.zeros((H, W, 3), np.float32)
        assert L.sol_read(h, self.fp(out)) == _abi.SOL_OK
        return out

    def random_call(self, rng):
        """One call: (text, thunk) - the text is known (and traced) BEFORE the call is made."""
        L, h = self.lib, self.h
        pick = lambda pool: pool[int(rng.integers(len(pool)))]
        k = int(rng.integers(24))
        if k < 5:
            f, n, seed = pick(FIRST), pick(COUNT), int(rng.integers(0, 1 << 63))
            fn = ("sol_render", "sol_render_counted", "sol_render_aux")[int(rng.integers(3))] if k == 4 else "sol_render"
            return f"{fn}({f}, {n}, {seed})", lambda: getattr(L, fn)(h, f, n, seed)
        if k == 5:
            which = ("sol_clear", "sol_sync", "sol_clear_aux")[int(rng.integers(3))]
            return which, lambda: getattr(L, which)(h)
        if k == 6:
            null = rng.integers(8) == 0
            return f"sol_read({'NULL' if null else 'buf'})", lambda: L.sol_read(h, None if null else self.fp(self.img))
        if k == 7:
            a, b = (None if rng.integers(3) == 0 else self.fp(self.img)), (None if rng.integers(3) == 0 else self.fp(self.img2))
            return f"sol_read_aux({'NULL' if a is None else 'buf'}, {'NULL' if b is None else 'buf'})", lambda: L.sol_read_aux(h, a, b)
        if k in (8, 9):
            r, w = pick(PARTITION)
            return f"sol_scene_set_partition({r}, {w})", lambda: L.sol_scene_set_partition(h, r, w)
        if k in (10, 11):
            opt = int(rng.integers(0, 10)) if rng.integers(8) else int(rng.integers(-5, 1000))
            val = pick([0, 1, 2, 3, 4, 5, 8, 16, 32, 63, 64, 65, -1, -2, 1 << 40, -(1 << 40)])
            if opt == _abi.OPT_KERNEL and val > 1:
                val = 1  # (the A/B kernels are another library's subject)
            return f"sol_scene_set_option({opt}, {val})", lambda: L.sol_scene_set_option(h, opt, val)
        if k == 12:
            p = C.c_void_p()
            rc = L.sol_resolve_image(h, C.byref(p))
            if rc != _abi.SOL_OK:
                return "sol_resolve_image", lambda: rc
            ns = pick([0, 1, 16, 64, 0xFFFFFFFF])
            null = rng.integers(8) == 0
            return f"sol_resolve_image + sol_tonemap_rgb8(ns {ns}{', NULL' if null else ''})", lambda: L.sol_tonemap_rgb8(h, None if null else p, ns, self.rgb8.ctypes.data_as(C.POINTER(C.c_uint8)))
        if k == 13:
            p = C.c_void_p()
            rc = L.sol_resolve_image(h, C.byref(p))
            if rc != _abi.SOL_OK:
                return "sol_resolve_image", lambda: rc
            ns, frac, thr, mx = pick([0, 1, 16, 0xFFFFFFFF]), pick(F64), pick(F64), pick(F64)
            if rng.integers(2):
                return f"sol_bloom(ns {ns}, {frac}, {thr}, {mx})", lambda: L.sol_bloom(h, p, ns, frac, thr, mx)
            return f"sol_bloom_rgb8(ns {ns}, {frac}, {thr}, {mx})", lambda: L.sol_bloom_rgb8(h, p, ns, frac, thr, mx, self.rgb8.ctypes.data_as(C.POINTER(C.c_uint8)))
        if k == 14:
            x, y = pick([0, 1, W - 1, W, 0xFFFFFFFF]), pick([0, 1, H - 1, H, 0xFFFFFFFF])
            rows = pick([0, 1, 2, 64])
            return f"sol_debug_path({x}, {y}, rows {rows})", lambda: L.sol_debug_path(h, x, y, pick(FIRST), 5, None if rng.integers(8) == 0 else self.fp(self.rows), rows)
        if k == 15:
            info = _abi.SolSceneInfo()
            info.size = pick(SIZES + [C.sizeof(_abi.SolSceneInfo)] * 4)
            if info.size > C.sizeof(info):
                return f"sol_scene_info(size {info.size}: skipped, the caller's struct is smaller)", lambda: 0
            return f"sol_scene_info(size {info.size})", lambda: L.sol_scene_info(h, C.byref(info))
        if k == 16:
            ps = _abi.SolPathStats()
            ps.size = pick([0, 4, 7, 8, 16, 48, C.sizeof(_abi.SolPathStats)])
            st = _abi.SolStats()
            rc = L.sol_stats(h, C.byref(st))
            return f"sol_stats + sol_path_stats(size {ps.size})", lambda: rc or L.sol_path_stats(h, C.byref(ps))
        if k == 17:
            ms, grid = C.c_float(), C.c_uint32()
            if rng.integers(2):
                return "sol_kernel_timing", lambda: L.sol_kernel_timing(h, int(rng.integers(2)))
            return "sol_last_kernel_ms", lambda: L.sol_last_kernel_ms(h, C.byref(ms), C.byref(grid))
        if k == 18:
            n = pick([0, 1, L.sol_accum_floats(h) - 1, L.sol_accum_floats(h), 1 << 40])
            p = None if rng.integers(2) else L.sol_accum_ptr(h)
            return f"sol_scene_bind_accum({'NULL' if p is None else 'own'}, {n})", lambda: L.sol_scene_bind_accum(h, p, n)
        if k == 19:
            world = pick([1, 1, 2, 3, 0, -1, 0x7FFFFFFF])
            p = C.c_void_p()
            L.sol_resolve_image(h, C.byref(p))
            # (gathered = `world` compact buffers back to back: only world 1 - the handle's own accumulator - is a buffer this test owns)
            src = L.sol_accum_ptr(h) if world == 1 and not rng.integers(4) == 0 else None
            if world > 1:
                src = None
            return f"sol_unpermute({'acc' if src else 'NULL'}, world {world})", lambda: L.sol_unpermute(h, src, world, p)
        if k == 20:
            which = ("sol_gather", "sol_comm_destroy", "sol_comm_self_check")[int(rng.integers(3))]
            if which == "sol_gather":
                p = C.c_void_p()
                L.sol_resolve_image(h, C.byref(p))
                return "sol_gather (no communicator)", lambda: L.sol_gather(h, p)
            if which == "sol_comm_self_check":
                return "sol_comm_self_check: skipped (loads RCCL: tests/test_gpu_gather.py)", lambda: 0
            return which, lambda: getattr(L, which)(h)
        if k == 21:
            bt = (C.c_double * 4)()
            return "sol_scene_build_times + sol_max_samples_per_call", lambda: L.sol_scene_build_times(h, bt) or (0 if L.sol_max_samples_per_call(h) > 0 else -1)
        if k == 22:
            ks, sd = pick([0, 1, 2, 3, 5, 64, 255]), pick(F64)
            out = (C.c_double * 256)()
            return f"sol_gaussian_blur_weights({ks}, {sd})", lambda: L.sol_gaussian_blur_weights(ks, sd, None if rng.integers(8) == 0 else out)
        fn, n, si, so = pick([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 100, 0xFFFFFFFF]), pick([0, 1, 7, 64]), pick([0, 1, 8, 16, 32]), pick([0, 1, 8, 16, 32])
        a, b = np.zeros((64, 32), np.float32), np.zeros((64, 32), np.float32)
        return f"sol_eval(fn {fn}, n {n}, strides {si} {so})", lambda: L.sol_eval(0, fn, self.fp(a), n, si, self.fp(b), so)


CODES = (_abi.SOL_OK, _abi.SOL_EINVAL, _abi.SOL_EDEVICE, _abi.SOL_ENOLIGHT, _abi.SOL_EDEPTH, _abi.SOL_ENOMEM)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["test_scene", "cornell"])
def test_random_call_sequences_leave_a_sound_handle(name):
    make = {"test_scene": scenes.create_test_scene, "cornell": scenes.cornell_box}[name]
    sc = make(RenderConfig(W, H, 16))
    hd = Handle(sc)
    try:
        want = hd.frame()
        rng = np.random.default_rng({"test_scene": 21, "cornell": 22}[name] + 1000 * SEED_SHIFT)
        tally = {}
        for i in range(N_CALLS):
            text, call = hd.random_call(rng)
            if TRACE:
                print(f"{name} {i}: {text}", file=sys.stderr, flush=True)
            rc = call()
            assert rc in CODES, (i, text, rc, hd.lib.sol_last_error())
            key = text.split("(")[0].split(":")[0]
            tally[key] = tally.get(key, [0, 0])
            tally[key][0 if rc == _abi.SOL_OK else 1] += 1
            if i % 200 == 199:  # (and along the way, not only at the end)
                assert np.array_equal(hd.frame(), want), (i, text)
        assert np.array_equal(hd.frame(), want)
        print(name, {k: tuple(v) for k, v in sorted(tally.items())})
        assert sum(v[1] for v in tally.values()) > N_CALLS // 20  # (the pools do hold refused arguments)
    finally:
        hd.close()


def test_a_sample_count_in_the_last_fifteen_of_32_bits_is_refused():
    """Found by the sequences above (round 5): the chunk count (n + 15) / 16 wrapped to ZERO for n > 2^32 - 16, the launch went out with no items and a
    chunk count of nothing, and the device faulted. Every such n is now a refused work-item count like its neighbours."""
    hd = Handle(scenes.cornell_box(RenderConfig(W, H, 16)))
    try:
        want = hd.frame()
        for n in (0xFFFFFFFF, 0xFFFFFFF1, 0xFFFFFFF0, 0xFFFFFFEF):
            for fn in ("sol_render", "sol_render_counted", "sol_render_aux"):
                assert getattr(hd.lib, fn)(hd.h, 0, n, 1) == _abi.SOL_EINVAL, (fn, n)
                assert b"work items" in hd.lib.sol_last_error()
        assert np.array_equal(hd.frame(), want)
    finally:
        hd.close()


def test_function_evaluation_refuses_rows_narrower_than_the_function_reads():
    """Found by the sequences above: sol_eval(fn 3, 64 rows, strides 8 / 16) - Onb::new and friends read 7 and WRITE 18 floats per row - wrote past the
    device copy of `out`: a GPU memory fault. Strides below a function's row width, and unknown functions, are now SOL_EINVAL."""
    L = _abi.load_hip()
    a, b = np.zeros((64, 32), np.float32), np.zeros((64, 32), np.float32)
    pa, pb = a.ctypes.data_as(C.POINTER(C.c_float)), b.ctypes.data_as(C.POINTER(C.c_float))
    widths = [(3, 7), (3, 5), (5, 2), (7, 18), (13, 2), (24, 4), (17, 4), (12, 2), (4, 7)]
    for fn, (wi, wo) in enumerate(widths):
        assert L.sol_eval(0, fn, pa, 64, wi, pb, wo) == _abi.SOL_OK, (fn, L.sol_last_error())
        assert L.sol_eval(0, fn, pa, 64, wi - 1, pb, wo) == _abi.SOL_EINVAL and L.sol_eval(0, fn, pa, 64, wi, pb, wo - 1) == _abi.SOL_EINVAL, fn
    assert L.sol_eval(0, 9, pa, 64, 32, pb, 32) == _abi.SOL_EINVAL and L.sol_eval(0, 0xFFFFFFFF, pa, 1, 32, pb, 32) == _abi.SOL_EINVAL
    assert L.sol_eval(0, 0, pa, 0, 3, pb, 7) == _abi.SOL_OK  # (no rows: nothing to do)


def test_destroying_a_handle_returns_its_device_memory():
    """sol_scene_destroy frees everything a handle ever allocated - also what is allocated lazily (auxiliary planes, bloom buffers, the gather buffer of a
    local gather, the balanced partition's tables, staging for the fine tail): device memory free before == free after 30 create / use / destroy cycles
    (hipMemGetInfo; the first cycle is outside the comparison: code objects and the runtime's own pools are loaded once)."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    sc = scenes.create_test_scene(RenderConfig(W, H, 16))

    def cycle():
        a, b = Handle(sc), Handle(sc)
        L = a.lib
        for k, hd in enumerate((a, b)):
            assert L.sol_scene_set_option(hd.h, _abi.OPT_BALANCED_PARTITION, 1) == _abi.SOL_OK
            assert L.sol_scene_set_partition(hd.h, k, 2) == _abi.SOL_OK
            assert L.sol_render(hd.h, 0, 37, 5) == _abi.SOL_OK and L.sol_render_aux(hd.h, 0, 4, 5) == _abi.SOL_OK
        img = C.c_void_p()
        assert L.sol_gather_local((C.c_void_p * 2)(a.h.value, b.h.value), 2, C.byref(img)) == _abi.SOL_OK
        assert L.sol_bloom(a.h, img, 37, 0.1, 1.0, 1e30) == _abi.SOL_OK
        assert L.sol_tonemap_rgb8(a.h, img, 37, a.rgb8.ctypes.data_as(C.POINTER(C.c_uint8))) == _abi.SOL_OK
        assert L.sol_read_aux(a.h, a.fp(a.img), a.fp(a.img2)) == _abi.SOL_OK
        a.close()
        b.close()

    cycle()
    before = free_bytes()
    for _ in range(30):
        cycle()
    after = free_bytes()
    assert abs(before - after) <= (8 << 20), (before, after)  # (the runtime may keep a few pooled megabytes; a leaked handle of this scene is ~3 MB per cycle x 30)
