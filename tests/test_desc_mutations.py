"""`SolSceneDesc` is the caller's memory: every index, reference, count and number in it is checked by `sol_scene_create` (and the two
descriptor-taking diagnostics) BEFORE anything reaches the device - "errors are codes + sol_last_error(), never an abort" (DESIGN.md 1,
SURVEY 8b "Errors"). Here valid descriptors of the reference's scenes are mutated one field at a time - references out of range, to
themselves, of the wrong kind; material and texture ids beyond their tables; counts cut short; NaN, infinite and huge coordinates;
textures outside the texel buffer; null arrays - and every entry point must come back with a code. Without a GPU a descriptor that
survives validation ends in SOL_EDEVICE; tests/tools/sanitize.sh runs the same mutations under AddressSanitizer + UBSan, where an
out-of-range read inside the validation itself would be reported. (Counts are only ever REDUCED: a count beyond the caller's array is a
lie no callee can detect.) The GPU half renders what survives with a short launch: a descriptor the validation accepts must not hang or
fault the device, whatever picture it makes."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from solstrale_amd import RenderConfig, _abi, scenes

F64_POOL = [float("nan"), float("inf"), float("-inf"), 0.0, -0.0, 1e300, -1e300, 1e-300, 3e38, -3e38, 5e-324, 1e15, -1e15, 1.0, -1.0]
U32_POOL = [0, 1, 0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 0x10000000, 0x20000000, 0x30000000, 0x40000000, 0x50000000, 0x0FFFFFFF, 0x1FFFFFFF, 65536]
I32_POOL = [-1, -2, 0, 1, 2, 3, 5, 7, 127, 128, 255, 256, 0x7FFFFFFF, -0x80000000, 1000]
U64_POOL = [0, 1, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 1 << 40, (1 << 32) - 16]
ARRAYS = [("nodes", "n_nodes"), ("spheres", "n_spheres"), ("quads", "n_quads"), ("triangles", "n_triangles"), ("mediums", "n_mediums"),
          ("materials", "n_materials"), ("textures", "n_textures"), ("lights", "n_lights")]
SCALARS = ["abi_version", "width", "height", "shader_kind", "max_depth", "root", "env_scale"]
COUNTS = [c for _, c in ARRAYS] + ["env_width", "env_height"]  # (the environment map's size is the length of the caller's env_texels)
SEEDS = {"test_scene": 11, "cornell": 12, "obj_box": 13, "blend": 14, "environment": 15, "normal_map": 16}
N_MUTATIONS = int(os.environ.get("SOL_TEST_MUTATIONS", "600"))  # (a longer campaign: SOL_TEST_MUTATIONS=20000 SOL_TEST_MUTATION_SEED=k)
SEED_SHIFT = int(os.environ.get("SOL_TEST_MUTATION_SEED", "0"))
TRACE = bool(os.environ.get("SOL_TEST_MUTATION_TRACE"))  # (names every mutation on stderr before it is tried: run with -s)
OK_CODES = (_abi.SOL_EINVAL, _abi.SOL_EDEVICE, _abi.SOL_ENOLIGHT)


def _slots(obj):
    """Every scalar of a ctypes structure / array as (container, key, ctype)."""
    if isinstance(obj, C.Structure):
        for name, tp in obj._fields_:
            if name.startswith("_"):
                continue
            if issubclass(tp, (C.Structure, C.Array)):
                yield from _slots(getattr(obj, name))
            else:
                yield obj, name, tp
    else:
        for i in range(len(obj)):
            if issubclass(obj._type_, (C.Structure, C.Array)):
                yield from _slots(obj[i])
            else:
                yield obj, i, obj._type_


def _get(c, k):
    return getattr(c, k) if isinstance(k, str) else c[k]


def _put(c, k, v):
    if isinstance(k, str):
        setattr(c, k, v)
    else:
        c[k] = v


def _pool(tp, rng, d):
    if tp in (C.c_double, C.c_float):
        return F64_POOL[rng.integers(len(F64_POOL))]
    if tp is C.c_uint64:
        return U64_POOL[rng.integers(len(U64_POOL))]
    if tp is C.c_int32:
        return I32_POOL[rng.integers(len(I32_POOL))]
    r = int(rng.integers(4))
    if r == 0:  # a reference of a valid KIND whose index is just past, or anywhere in, one of the tables
        kind = int(rng.integers(0, 8)) << 28
        n = [d.n_nodes, d.n_spheres, d.n_quads, d.n_triangles, d.n_mediums][int(rng.integers(5))]
        return kind | (n if rng.integers(2) else int(rng.integers(0, max(1, n))))
    if r == 1:
        return int(rng.integers(0, 1 << 32))
    return U32_POOL[rng.integers(len(U32_POOL))]


def _mutations(d, rng, n):
    """n single-field mutations as (describe, apply, undo) over the descriptor's arrays and scalars."""
    out = []
    tables = [(a, c) for a, c in ARRAYS if getattr(d, c) > 0]
    for _ in range(n):
        what = int(rng.integers(10))
        if what < 7 and tables:  # one field of one record
            a, c = tables[rng.integers(len(tables))]
            i = int(rng.integers(getattr(d, c)))
            arr = getattr(d, a)
            if a == "lights":
                cont, key, tp = arr, i, C.c_uint32
            else:
                slots = list(_slots(arr[i]))
                cont, key, tp = slots[rng.integers(len(slots))]
            name = f"{a}[{i}].{key}"
        elif what < 9:  # a top-level scalar, or one of the camera / background numbers
            slots = [(d, s, dict(d._fields_)[s]) for s in SCALARS] + list(_slots(d.camera)) + list(_slots(d.background))
            cont, key, tp = slots[rng.integers(len(slots))]
            name = f"desc.{key}"
        else:  # a count cut short (never raised), or zero
            c = COUNTS[rng.integers(len(COUNTS))]
            cont, key, tp = d, c, None
            name = f"desc.{c}"
        old = _get(cont, key)
        new = (int(rng.integers(0, old + 1)) if old else 0) if tp is None else _pool(tp, rng, d)
        out.append((f"{name}: {old!r} -> {new!r}", cont, key, old, new))
    groups = []  # (a quarter of the cases change two or three fields at once)
    while out:
        k = 1 if rng.integers(4) else int(rng.integers(2, 4))
        groups.append(out[:k])
        out = out[k:]
    return groups


def _scenes(width=16, height=12):
    rc = RenderConfig(width, height, 1)
    return {
        "test_scene": scenes.create_test_scene(rc),  # spheres, quads, triangles, a constant medium, three kinds of light, an image texture
        "cornell": scenes.cornell_box(rc),
        "normal_map": scenes.create_normal_mapping_scene(rc, (30., 30., 30.), True),  # an image used as a normal map
        "obj_box": scenes.create_obj_with_box(rc, "box.obj"),  # a loaded OBJ: triangles only
        "blend": scenes.create_blend_material_scene(rc, 0.5),
        "environment": scenes.create_test_scene_with_environment(rc, size=(32, 16)),
    }


@pytest.fixture(scope="module")
def base():
    return _scenes()


@pytest.fixture(scope="module")
def base_gpu():
    return _scenes(96, 64)


def _call_all(lib, sc):
    h = C.c_void_p()
    rc = lib.sol_scene_create(sc.desc_ptr, 0, C.byref(h))
    codes = [rc]
    if rc == _abi.SOL_OK:
        return codes, h
    chk = _abi.SolTreeCheck()
    codes.append(lib.sol_world_tree_check_ex(sc.desc_ptr, 0, C.byref(chk), C.sizeof(chk)))
    nb = ((sc.desc.width + 7) // 8) * ((sc.desc.height + 7) // 8) if 0 < sc.desc.width < 4096 and 0 < sc.desc.height < 4096 else 1
    flags = (C.c_uint8 * max(1, nb))()
    n = C.c_uint32()
    codes.append(lib.sol_background_blocks(sc.desc_ptr, 0, flags, nb, C.byref(n)))
    return codes, None


@pytest.mark.timeout(600)
@pytest.mark.parametrize("name", list(SEEDS))
def test_mutated_descriptors_come_back_with_a_code(base, name):
    """600 single-field mutations per scene (seeded): sol_scene_create, sol_world_tree_check_ex, sol_background_blocks return a code each
    time - SOL_EINVAL / SOL_ENOLIGHT where the validation refuses, SOL_EDEVICE where a still-valid descriptor meets a box without a GPU -
    and the untouched descriptor is still accepted afterwards (the library keeps nothing of a refused one)."""
    from solstrale_amd import device_count
    if device_count() > 0:
        pytest.skip("the GPU half below creates what survives")
    lib = _abi.load_hip()
    sc = base[name]
    d = sc.desc
    rng = np.random.default_rng(SEEDS[name] + 1000 * SEED_SHIFT)
    refused = 0
    for group in _mutations(d, rng, N_MUTATIONS):
        text = "; ".join(g[0] for g in group)
        for _, cont, key, old, new in group:
            _put(cont, key, new)
        try:
            codes, _ = _call_all(lib, sc)
        finally:
            for _, cont, key, old, new in reversed(group):
                _put(cont, key, old)
        assert codes[0] in OK_CODES, (text, codes)
        assert all(c in (_abi.SOL_OK,) + OK_CODES for c in codes[1:]), (text, codes)
        refused += codes[0] != _abi.SOL_EDEVICE
    codes, _ = _call_all(lib, sc)
    assert codes[0] == _abi.SOL_EDEVICE  # (valid, and no GPU here)
    assert refused >= N_MUTATIONS // 8, refused  # (the mutations do reach the validation: a fifth or more are refused there)


def test_an_image_texture_outside_the_texel_buffer_is_refused_also_where_the_sums_wrap(base):
    """Found by the GPU half below (round 5): `texel_offset + width * height * 3 > n_texel_bytes` wraps for an offset of 2^64 - 1, or a
    2^32 - 1 square image, into a small number; the scene was accepted and the first texel fetch faulted the device."""
    lib = _abi.load_hip()
    sc = base["test_scene"]
    d = sc.desc
    k = next(i for i in range(d.n_textures) if d.textures[i].kind == _abi.TEX_IMAGE)
    t = d.textures[k]
    for field, value in (("texel_offset", 0xFFFFFFFFFFFFFFFF), ("texel_offset", (1 << 64) - t.width * t.height * 3 + 1), ("width", 0xFFFFFFFF),
                         ("height", 0xFFFFFFFF), ("texel_offset", d.n_texel_bytes), ("width", 0)):
        old = getattr(t, field)
        setattr(t, field, value)
        other = None
        if field == "width" and value == 0xFFFFFFFF:  # (2^32 - 1 squared, times three: wraps twice)
            other = t.height
            t.height = 0xFFFFFFFF
        try:
            h = C.c_void_p()
            rc = lib.sol_scene_create(sc.desc_ptr, 0, C.byref(h))
        finally:
            setattr(t, field, old)
            if other is not None:
                t.height = other
        assert rc == _abi.SOL_EINVAL and b"outside texel buffer" in lib.sol_last_error(), (field, value, rc)


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", list(SEEDS))
def test_what_the_validation_accepts_renders_without_a_fault(base_gpu, name):
    """The same mutations on the GPU box: a descriptor that sol_scene_create ACCEPTS (a NaN colour, a camera at 1e15, a light of
    zero area, a shortened primitive table ...) is rendered with one sample and read back; the device must neither fault nor hang,
    and the untouched scene renders the same frame before and after (nothing of a mutated scene survives in the library)."""
    lib = _abi.load_hip()
    sc = base_gpu[name]
    d = sc.desc
    w, h = sc.width, sc.height

    def render(hnd):
        img = np.zeros((h, w, 3), np.float32)
        assert lib.sol_render(hnd, 0, 4, 1234) == _abi.SOL_OK
        assert lib.sol_read(hnd, img.ctypes.data_as(C.POINTER(C.c_float)), None, None) == _abi.SOL_OK
        return img

    codes, hnd = _call_all(lib, sc)
    assert codes[0] == _abi.SOL_OK
    before = render(hnd)
    lib.sol_scene_destroy(hnd)
    rng = np.random.default_rng(SEEDS[name] + 1000 * SEED_SHIFT)
    accepted = 0
    for group in _mutations(d, rng, max(200, N_MUTATIONS // 3)):
        if any(g[2] in ("width", "height", "max_depth") for g in group):
            continue  # (a 2^30-pixel frame or a 2^32-deep path is a legitimate, very long job - not this test's subject)
        text = "; ".join(g[0] for g in group)
        if TRACE:
            print(f"{name}: {text}", file=sys.stderr, flush=True)
        for _, cont, key, old, new in group:
            _put(cont, key, new)
        try:
            codes, hnd = _call_all(lib, sc)
            if codes[0] == _abi.SOL_OK:
                accepted += 1
                render(hnd)
                lib.sol_scene_destroy(hnd)
            else:
                assert codes[0] in OK_CODES, (text, codes)
        finally:
            for _, cont, key, old, new in reversed(group):
                _put(cont, key, old)
    codes, hnd = _call_all(lib, sc)
    assert codes[0] == _abi.SOL_OK
    after = render(hnd)
    lib.sol_scene_destroy(hnd)
    print(f"{name}: {accepted} mutated descriptors accepted and rendered")
    assert accepted >= 20 and np.array_equal(before, after, equal_nan=True)
