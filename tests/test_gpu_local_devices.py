"""One process, several GPUs: `sol_gather_local` (csrc/sol_comm.cpp) and the host mirror's `ray_trace(scene, output, abort, devices)` - the shape a caller
of the reference's blocking `ray_trace()` (src/lib.rs:93-99) has. The frame's 8x8 blocks are dealt out over n handles, each renders its blocks, the compact
accumulators go to the first handle's device by peer copies and are un-permuted there. A test box has ONE GPU: the n handles all live on device 0 (the
partition, the concurrent launches on n streams, the gather buffer, the un-permute and the post-processors are the same code; only the copy's two ends
coincide). The picture must not depend on n: byte for byte the single-handle one."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import parity_util as pu
from solstrale_amd import BloomPostProcessor, HostError, RenderConfig, _abi, scenes

pytestmark = pytest.mark.gpu


def _handles(sc, n, balanced=False):
    lib = _abi.load_hip()
    hs = []
    for i in range(n):
        h = C.c_void_p()
        assert lib.sol_scene_create(sc.desc_ptr, 0, C.byref(h)) == _abi.SOL_OK, lib.sol_last_error()
        if balanced:
            assert lib.sol_scene_set_option(h, _abi.OPT_BALANCED_PARTITION, 1) == _abi.SOL_OK
        assert lib.sol_scene_set_partition(h, i, n) == _abi.SOL_OK
        hs.append(h)
    return lib, hs


def _gathered_frame(lib, hs, w, h, spp):
    for hd in hs:
        assert lib.sol_clear(hd) == _abi.SOL_OK and lib.sol_render(hd, 0, spp, pu.SEED) == _abi.SOL_OK  # (asynchronous: all n in flight)
    arr = (C.c_void_p * len(hs))(*[hd.value for hd in hs])
    img = C.c_void_p()
    assert lib.sol_gather_local(arr, len(hs), C.byref(img)) == _abi.SOL_OK, lib.sol_last_error()
    out = np.zeros((h, w, 3), np.float32)
    assert lib.sol_read_image(hs[0], out.ctypes.data_as(C.POINTER(C.c_float))) == _abi.SOL_OK
    return out


@pytest.mark.parametrize("make,size", [(scenes.create_test_scene, (250, 131)), (scenes.sponza_like, (320, 184))], ids=["test_scene_ragged", "c3"])
def test_local_gather_gives_the_single_handle_frame(make, size):
    w, h = size
    sc = make(RenderConfig(w, h, 37))
    lib, one = _handles(sc, 1)
    try:
        want = _gathered_frame(lib, one, w, h, 37)
        direct = np.zeros_like(want)
        assert lib.sol_read(one[0], direct.ctypes.data_as(C.POINTER(C.c_float))) == _abi.SOL_OK
        assert np.array_equal(want, direct)  # (n = 1: the gather is sol_read's un-permute)
    finally:
        lib.sol_scene_destroy(one[0])
    for n, balanced in ((2, False), (3, False), (5, True), (8, False)):
        lib, hs = _handles(sc, n, balanced)
        try:
            got = _gathered_frame(lib, hs, w, h, 37)
            assert np.array_equal(got, want), (n, balanced, int((got != want).any(axis=-1).sum()))
            got = _gathered_frame(lib, hs, w, h, 37)  # (again: the gather buffer is reused)
            assert np.array_equal(got, want)
        finally:
            for hd in hs:
                lib.sol_scene_destroy(hd)


def test_local_gather_refuses_what_is_not_a_partition():
    sc = scenes.cornell_box(RenderConfig(64, 64, 4))
    lib, hs = _handles(sc, 3)
    try:
        img = C.c_void_p()
        arr = (C.c_void_p * 3)(*[h.value for h in hs])
        assert lib.sol_gather_local(arr, 3, C.byref(img)) == _abi.SOL_OK
        for bad, n in (((hs[0], hs[1]), 2), ((hs[1], hs[0], hs[2]), 3), ((hs[0], hs[0], hs[2]), 3), ((hs[0], hs[1], None), 3)):
            arr = (C.c_void_p * len(bad))(*[b.value if b is not None else None for b in bad])
            assert lib.sol_gather_local(arr, n, C.byref(img)) == _abi.SOL_EINVAL, bad
        assert lib.sol_gather_local(None, 3, C.byref(img)) == _abi.SOL_EINVAL and lib.sol_gather_local(arr, 0, C.byref(img)) == _abi.SOL_EINVAL
        assert lib.sol_gather_local((C.c_void_p * 3)(*[h.value for h in hs]), 3, None) == _abi.SOL_EINVAL
    finally:
        for hd in hs:
            lib.sol_scene_destroy(hd)


@pytest.mark.parametrize("strategy", ["only_final", "every_sample"])
def test_ray_trace_on_several_devices_is_ray_trace(strategy):
    """The host mirror's pass loop on 1, 2 and 3 handles: the same progress events, the same RGB8 picture - with the Nop post-processor and with Bloom
    (which runs on the gathered image of the first device)."""
    for pp in ((), (BloomPostProcessor(0.05, None, None),)):
        rc = RenderConfig(200, 104, 24 if strategy == "only_final" else 5)
        if pp:
            rc.post_processors = list(pp)
        sc = scenes.create_test_scene(rc)
        ev1, img1 = sc.ray_trace(strategy=strategy)
        for devices in ([0], [0, 0], [0, 0, 0]):
            ev, img = sc.ray_trace(strategy=strategy, devices=devices)
            assert len(ev) == len(ev1) and [e[3] for e in ev] == [e[3] for e in ev1]
            assert img.shape == img1.shape and np.array_equal(img, img1), (strategy, bool(pp), devices)
    with pytest.raises(HostError):
        sc.ray_trace(devices=[])
    with pytest.raises(HostError):
        sc.ray_trace(devices=[0, 99])


def test_native_program_on_several_devices(tmp_path):
    exe = os.path.join(os.path.dirname(_abi.load_hip()._name), "profiling")
    outs = []
    for k, devices in enumerate(("0", "0,0", "0,0,0,0")):
        out = tmp_path / f"o{k}.ppm"
        r = subprocess.run([exe, "--spp", "48", "--width", "320", "--height", "160", "--devices", devices, "--out", str(out)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        assert json.loads(r.stdout.strip().splitlines()[-1])["devices"] == devices.count(",") + 1
        outs.append(out.read_bytes())
    assert outs[0] == outs[1] == outs[2]
