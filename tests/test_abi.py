"""The C-ABI libraries load, export every symbol include/*.h declares, the Python struct mirror has the C layout, and the
product has no CPU fallback (without a GPU every compute entry point fails loudly with SOL_EDEVICE)."""
import ctypes as C
import os
import re

import pytest

from solstrale_amd import DeviceError, DeviceScene, RenderConfig, _abi, device_count, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"static inline[^{;]*\{.*?\n\}", "", text, flags=re.S)  # (inline helpers both sides compile - sol_scene_has_needles - are not exports)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    lib = _abi.load_hip()
    names = _declared("solstrale_hip.h", "sol_")
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), f"libsolstrale_hip.so does not export {n}"
    assert sorted(_abi.HIP_SYMBOLS) == names


def test_integration_guide_names_every_entry_point():
    """INTEGRATION.md shows the reference-side binding: every entry point of the device header is declared in its `extern "C"` block
    or named among the diagnostics a binding may leave out - a new entry point must not go undocumented."""
    guide = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in _declared("solstrale_hip.h", "sol_") if not re.search(r"\b" + n + r"\b", guide)]
    assert not missing, f"INTEGRATION.md does not mention {missing}"
    bound = set(re.findall(r"pub fn (sol_[a-z0-9_]+)", guide))
    core = {"sol_scene_create", "sol_scene_destroy", "sol_render", "sol_read", "sol_clear", "sol_sync", "sol_comm_init", "sol_gather", "sol_last_error"}
    assert core <= bound, f"the Rust extern block lacks {sorted(core - bound)}"


def test_host_library_exports_every_declared_symbol():
    lib = _abi.load_host()
    names = [n for n in _declared("solstrale_host.h", "solh_") if not n.endswith("_fn")]
    for n in names:
        assert hasattr(lib, n), f"libsolstrale_host.so does not export {n}"
    assert sorted(_abi.HOST_SYMBOLS) == names


def test_struct_layouts_match_the_c_side():
    lib = _abi.load_host()
    sizes = (C.c_uint32 * 11)()
    lib.solh_abi_sizes(sizes)
    assert [int(x) for x in sizes] == [C.sizeof(s) for s in _abi.ABI_STRUCTS]


def test_device_record_sizes():
    from solstrale_amd import record_sizes
    # node = the 7-wide quantised node the world is searched through (64 B: half a cache line, implicit child addresses)
    assert record_sizes() == {"node": 64, "sphere": 32, "quad": 80, "triangle": 48, "triangle_shade": 64, "material": 48}


def test_hip_library_is_a_gfx950_code_object():
    data = open(_abi.HIP_LIB, "rb").read()
    assert b"gfx950" in data and b"sol_render_kernel" in data


@pytest.mark.skipif(device_count() > 0, reason="only meaningful without a GPU")
def test_no_cpu_fallback_without_a_gpu():
    sc = scenes.cornell_box(RenderConfig(16, 16, 1))
    with pytest.raises(DeviceError) as e:
        DeviceScene(sc)
    assert e.value.code == _abi.SOL_EDEVICE


def test_scene_validation_happens_before_the_device():
    # Renderer::new's light check is part of sol_scene_create (SOL_ENOLIGHT) and does not need a GPU
    sc = scenes.create_simple_test_scene(RenderConfig(20, 10, 1), add_light=False)
    with pytest.raises(DeviceError) as e:
        DeviceScene(sc)
    assert e.value.code == _abi.SOL_ENOLIGHT and e.value.msg == "Scene should have at least one light"
    sc = scenes.cornell_box(RenderConfig(16, 16, 1))
    sc.desc.abi_version = 99
    with pytest.raises(DeviceError) as e:
        DeviceScene(sc)
    assert e.value.code == _abi.SOL_EINVAL
    sc.desc.abi_version = _abi.SOL_ABI_VERSION
    sc.desc.root = (1 << 28) | 4000  # node index out of range
    with pytest.raises(DeviceError) as e:
        DeviceScene(sc)
    assert e.value.code == _abi.SOL_EINVAL


def test_missing_communication_library_is_an_error_code_not_a_crash():
    """A box without librccl.so (a single-GPU host that calls sol_comm_unique_id or bench.py --gpus N): SOL_EDEVICE with the loader's
    message, not a crash (round 3's loader read dlerror() twice: the second read is NULL). In its own process: the loader runs once."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from solstrale_amd import DeviceError, comm_unique_id, _abi\n"
            "try:\n    comm_unique_id()\n    print('LOADED')\n"
            "except DeviceError as e:\n    print('CODE', e.code, e.msg)\n") % os.path.join(ROOT, "solstrale-rust_amd")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SOL_RCCL_LIB="/nonexistent/librccl.so.1"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr[-500:])
    assert r.stdout.startswith(f"CODE {_abi.SOL_EDEVICE} cannot load librccl.so:") and "nonexistent" in r.stdout, r.stdout


def test_scene_extent_is_bounded_before_the_device():
    """The 7-wide node test evaluates plane parameters scaled by up to 1e21 (clamped inverse direction x cull scale): coordinates beyond
    2^38 would overflow them, so such a scene is refused at creation (SOL_EINVAL) instead of losing hits."""
    from solstrale_amd import CameraConfig, SceneBuilder
    b = SceneBuilder()
    m = b.Lambertian(b.SolidColor(.5, .5, .5))
    world = [b.Sphere((0., 0., 0.), 1., m), b.Sphere((3e11, 0., 0.), 1., m), b.Sphere((0., 5., 0.), 1., b.DiffuseLight(1, 1, 1))]
    sc = b.finish(b.Bvh(world), CameraConfig(40., 0., (0., 0., 5.), (0., 0., 0.), (0, 1, 0)), (0., 0., 0.), RenderConfig(8, 8, 1))
    with pytest.raises(DeviceError) as e:
        DeviceScene(sc)
    assert e.value.code == _abi.SOL_EINVAL and "2^38" in e.value.msg


def test_tree_check_keeps_its_first_layout_for_old_bindings():
    """SolTreeCheck has no size field and grew in round 4: the plain entry point writes the 48-byte first layout only (a binding compiled
    against that header is not overrun), sol_world_tree_check_ex fills what the caller's size holds and refuses less than the first layout."""
    from solstrale_amd import RenderConfig, scenes
    lib = _abi.load_hip()
    assert C.sizeof(_abi.SolTreeCheck) == 64 and _abi.SolTreeCheck.n_extra_references.offset == 48
    sc = scenes.cornell_box(RenderConfig(32, 32, 1))
    buf = (C.c_uint8 * 80)(*([0xAB] * 80))
    assert lib.sol_world_tree_check(sc.desc_ptr, 0, C.cast(buf, C.POINTER(_abi.SolTreeCheck))) == 0
    assert all(b == 0xAB for b in buf[48:]) and any(b != 0xAB for b in buf[:48])
    full = _abi.SolTreeCheck()
    assert lib.sol_world_tree_check_ex(sc.desc_ptr, 0, C.byref(full), C.sizeof(full)) == 0
    assert bytes(buf[:48]) == bytes(full)[:48] and full.n_wide >= 1 and full.leaf_mismatches == 0
    assert lib.sol_world_tree_check_ex(sc.desc_ptr, 0, C.byref(full), 40) == _abi.SOL_EINVAL


def test_creation_options_are_checked_before_the_device():
    """SolCreateOptions: an unknown tree choice, a size field no version of the struct had, a pre-splitting budget or a number of reinsertion rounds that
    can only be a typo (a build of hours) are SOL_EINVAL - also without a GPU."""
    import ctypes as C
    lib = _abi.load_hip()
    sc = scenes.cornell_box(RenderConfig(16, 16, 1))
    for field, value in (("world_tree", 99), ("world_tree", -9), ("size", 4), ("size", 1 << 20), ("split_percent", 1001), ("split_percent", 0x7FFFFFFF),
                         ("reinsertion_rounds", 1025), ("reinsertion_rounds", 0x7FFFFFFF)):
        opt = _abi.SolCreateOptions()
        opt.size = C.sizeof(opt)
        setattr(opt, field, value)
        h = C.c_void_p()
        assert lib.sol_scene_create_ex(sc.desc_ptr, 0, C.byref(opt), C.byref(h)) == _abi.SOL_EINVAL, (field, value)
    opt = _abi.SolCreateOptions()
    opt.size = C.sizeof(opt)
    opt.split_percent, opt.reinsertion_rounds = 1000, 1024
    h = C.c_void_p()
    assert lib.sol_scene_create_ex(sc.desc_ptr, 0, C.byref(opt), C.byref(h)) in (_abi.SOL_EDEVICE, _abi.SOL_OK)
    if h:
        lib.sol_scene_destroy(h)
