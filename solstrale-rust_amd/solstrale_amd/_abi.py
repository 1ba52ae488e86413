"""ctypes mirror of include/solstrale_hip.h and include/solstrale_host.h.

This is harness plumbing for tests/ and bench.py: the product is the C-ABI shared library
(`libsolstrale_hip.so`, hand-written HIP for gfx950) and the C++ host above it. Layouts are asserted against
the C side at import time through `solh_abi_sizes` (tests/test_abi.py).
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(PKG_DIR))
BUILD_DIR = os.environ.get("SOLSTRALE_BUILD_DIR") or os.path.join(os.path.dirname(PKG_DIR), "_build")  # env: kernel A/B variants
HIP_LIB = os.path.join(BUILD_DIR, "libsolstrale_hip.so")
HOST_LIB = os.path.join(BUILD_DIR, "libsolstrale_host.so")

SOL_ABI_VERSION = 2
SOL_OK, SOL_EINVAL, SOL_ENOLIGHT, SOL_EDEVICE, SOL_EDEPTH, SOL_ENOMEM = 0, -1, -2, -3, -4, -5
REF_NONE, REF_NODE, REF_SPHERE, REF_QUAD, REF_TRIANGLE, REF_MEDIUM = range(6)
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC, MAT_BLEND = range(6)
TEX_SOLID, TEX_IMAGE = 0, 1
SHADER_PATH_TRACING, SHADER_ALBEDO, SHADER_NORMAL, SHADER_SIMPLE = range(4)


def ref_kind(r):
    return (r >> 28) & 0xF


def ref_index(r):
    return r & 0x0FFFFFFF


class SolAabb(C.Structure):
    _fields_ = [("v", C.c_double * 6)]


class SolBvhNode(C.Structure):
    _fields_ = [("bbox", SolAabb), ("left", C.c_uint32), ("right", C.c_uint32)]


class SolSphere(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("radius", C.c_double), ("bbox", SolAabb),
                ("material", C.c_int32), ("dfs_index", C.c_uint32)]


class SolQuad(C.Structure):
    _fields_ = [("q", C.c_double * 3), ("u", C.c_double * 3), ("v", C.c_double * 3), ("normal", C.c_double * 3),
                ("d", C.c_double), ("w", C.c_double * 3), ("area", C.c_double), ("bbox", SolAabb),
                ("material", C.c_int32), ("dfs_index", C.c_uint32)]


class SolTriangle(C.Structure):
    _fields_ = [("v0", C.c_double * 3), ("v0v1", C.c_double * 3), ("v0v2", C.c_double * 3),
                ("normal", C.c_double * 3), ("tangent", C.c_double * 3), ("bi_tangent", C.c_double * 3),
                ("area", C.c_double), ("uv0", C.c_float * 2), ("uv1", C.c_float * 2), ("uv2", C.c_float * 2),
                ("bbox", SolAabb), ("material", C.c_int32), ("dfs_index", C.c_uint32)]


class SolMedium(C.Structure):
    _fields_ = [("boundary", C.c_uint32), ("material", C.c_int32), ("negative_inverse_density", C.c_double),
                ("bbox", SolAabb), ("dfs_index", C.c_uint32), ("_pad", C.c_uint32)]


class SolMaterial(C.Structure):
    _fields_ = [("kind", C.c_int32), ("albedo_tex", C.c_int32), ("normal_tex", C.c_int32), ("m1", C.c_int32),
                ("m2", C.c_int32), ("_pad", C.c_int32), ("param", C.c_double)]


class SolTexture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32), ("_pad", C.c_uint32),
                ("texel_offset", C.c_uint64), ("rgb", C.c_double * 3)]


class SolCamera(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("lower_left_corner", C.c_double * 3), ("horizontal", C.c_double * 3),
                ("vertical", C.c_double * 3), ("u", C.c_double * 3), ("v", C.c_double * 3),
                ("lens_radius", C.c_double)]


class SolSceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("shader_kind", C.c_uint32), ("max_depth", C.c_uint32), ("root", C.c_uint32),
                ("background", C.c_double * 3), ("camera", SolCamera),
                ("nodes", C.POINTER(SolBvhNode)), ("n_nodes", C.c_uint32),
                ("spheres", C.POINTER(SolSphere)), ("n_spheres", C.c_uint32),
                ("quads", C.POINTER(SolQuad)), ("n_quads", C.c_uint32),
                ("triangles", C.POINTER(SolTriangle)), ("n_triangles", C.c_uint32),
                ("mediums", C.POINTER(SolMedium)), ("n_mediums", C.c_uint32),
                ("materials", C.POINTER(SolMaterial)), ("n_materials", C.c_uint32),
                ("textures", C.POINTER(SolTexture)), ("n_textures", C.c_uint32),
                ("texels", C.POINTER(C.c_uint8)), ("n_texel_bytes", C.c_uint64),
                ("lights", C.POINTER(C.c_uint32)), ("n_lights", C.c_uint32),
                ("env_texels", C.POINTER(C.c_float)), ("env_width", C.c_uint32), ("env_height", C.c_uint32),
                ("env_scale", C.c_double)]


class SolStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "rays", "node_visits", "sphere_tests", "quad_tests",
                                          "triangle_tests", "shades", "texel_fetches", "max_stack")] + \
               [("phase", C.c_uint64 * 6)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "phase"}

    def phases(self):
        p = [int(x) for x in self.phase]
        return {k: (p[2 * i] / p[2 * i + 1] if p[2 * i + 1] else 0.0) for i, k in enumerate(("traverse", "shade", "generate"))}


ABI_STRUCTS = [SolAabb, SolBvhNode, SolSphere, SolQuad, SolTriangle, SolMedium, SolMaterial, SolTexture, SolCamera,
               SolSceneDesc, SolStats]

PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_uint8), C.c_uint32,
                          C.c_uint32)
ABORT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
IMAGE_DECODER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                               C.POINTER(C.POINTER(C.c_uint8)))

_D3 = C.POINTER(C.c_double)
_libs = {}


def _sig(lib, name, res, args):
    f = getattr(lib, name)
    f.restype = res
    f.argtypes = args
    return f


TREE_AUTO, TREE_REF, TREE_SAH8, TREE_SAH16, TREE_SAH64, TREE_DEVICE, TREE_HOST_PROBE = range(7)
OPT_SWITCH_BELOW, OPT_MAX_BLOCKS_PER_CU, OPT_KERNEL, OPT_WORK_ORDER, OPT_FINE_TAIL, OPT_BALANCED_PARTITION, OPT_BACKGROUND_BLOCKS = 1, 2, 3, 4, 5, 6, 7
UNIQUE_ID_BYTES = 128


class SolCreateOptions(C.Structure):
    _fields_ = [("size", C.c_uint32), ("world_tree", C.c_int32), ("no_work_order_probe", C.c_int32), ("split_percent", C.c_int32),
                ("reinsertion_rounds", C.c_int32), ("no_background_blocks", C.c_int32), ("reserved", C.c_int32 * 2)]


class SolSceneInfo(C.Structure):
    _fields_ = [("size", C.c_uint32), ("stack_bound", C.c_uint32), ("lds_stack", C.c_uint32), ("spill_stack", C.c_uint32),
                ("tree_fallback", C.c_uint32), ("tree_name", C.c_char * 32), ("tree_note", C.c_char * 192),
                ("split_references", C.c_uint32), ("split_triangles", C.c_uint32), ("split_area_ratio", C.c_float),
                ("reinsertion_moves", C.c_uint32), ("reinsertion_area_ratio", C.c_float),
                ("partition_table", C.c_uint32), ("partition_crc", C.c_uint32), ("strict_triangles", C.c_uint32),
                ("background_blocks", C.c_uint32), ("background_pixels", C.c_uint32)]


class SolPathStats(C.Structure):
    _fields_ = [("size", C.c_uint32), ("pad", C.c_uint32), ("samples", C.c_uint64), ("primary_hits", C.c_uint64), ("path_len", C.c_uint64 * 6)]


class SolTreeCheck(C.Structure):
    _fields_ = [("n_wide", C.c_uint32), ("n_leaf_refs", C.c_uint32), ("n_primitives", C.c_uint32), ("depth", C.c_uint32),
                ("max_children", C.c_uint32), ("box_violations", C.c_uint32), ("leaf_mismatches", C.c_uint32),
                ("bad_empty_slots", C.c_uint32), ("inner_area", C.c_double), ("leaf_area", C.c_double),
                ("n_extra_references", C.c_uint32), ("n_split_triangles", C.c_uint32), ("split_uncovered", C.c_uint32), ("reserved", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def load_hip():
    """Loads libsolstrale_hip.so (the product). Fails loudly if it has not been built."""
    if "hip" in _libs:
        return _libs["hip"]
    if not os.path.exists(HIP_LIB):
        raise RuntimeError(f"{HIP_LIB} is missing: run `python __graft_entry__.py build` (hipcc --offload-arch=gfx950)")
    lib = C.CDLL(HIP_LIB, mode=C.RTLD_GLOBAL)
    P = C.c_void_p
    _sig(lib, "sol_device_count", C.c_int, [])
    _sig(lib, "sol_scene_create", C.c_int, [C.POINTER(SolSceneDesc), C.c_int, C.POINTER(P)])
    _sig(lib, "sol_scene_destroy", None, [P])
    _sig(lib, "sol_scene_set_partition", C.c_int, [P, C.c_int, C.c_int])
    _sig(lib, "sol_accum_floats", C.c_size_t, [P])
    _sig(lib, "sol_accum_ptr", C.c_void_p, [P])
    _sig(lib, "sol_scene_bind_accum", C.c_int, [P, C.c_void_p, C.c_size_t])
    _sig(lib, "sol_scene_set_stream", C.c_int, [P, C.c_void_p])
    _sig(lib, "sol_clear", C.c_int, [P])
    _sig(lib, "sol_render", C.c_int, [P, C.c_uint32, C.c_uint32, C.c_uint64])
    _sig(lib, "sol_render_counted", C.c_int, [P, C.c_uint32, C.c_uint32, C.c_uint64])
    _sig(lib, "sol_sync", C.c_int, [P])
    _sig(lib, "sol_read", C.c_int, [P, C.POINTER(C.c_float)])
    _sig(lib, "sol_render_aux", C.c_int, [P, C.c_uint32, C.c_uint32, C.c_uint64])
    _sig(lib, "sol_clear_aux", C.c_int, [P])
    _sig(lib, "sol_read_aux", C.c_int, [P, C.POINTER(C.c_float), C.POINTER(C.c_float)])
    _sig(lib, "sol_unpermute", C.c_int, [P, C.c_void_p, C.c_int, C.c_void_p])
    _sig(lib, "sol_tonemap_rgb8", C.c_int, [P, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint8)])
    _sig(lib, "sol_resolve_image", C.c_int, [P, C.POINTER(C.c_void_p)])
    _sig(lib, "sol_bloom", C.c_int, [P, C.c_void_p, C.c_uint32, C.c_double, C.c_double, C.c_double])
    _sig(lib, "sol_bloom_rgb8", C.c_int, [P, C.c_void_p, C.c_uint32, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_uint8)])
    _sig(lib, "sol_gaussian_blur_weights", C.c_int, [C.c_uint32, C.c_double, C.POINTER(C.c_double)])
    _sig(lib, "sol_stats", C.c_int, [P, C.POINTER(SolStats)])
    _sig(lib, "sol_world_tree_check", C.c_int, [C.c_void_p, C.c_int, C.POINTER(SolTreeCheck)])
    _sig(lib, "sol_world_tree_check_ex", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t])
    _sig(lib, "sol_record_sizes", C.c_int, [C.POINTER(C.c_uint32)])
    _sig(lib, "sol_last_error", C.c_char_p, [])
    _sig(lib, "sol_debug_path", C.c_int, [P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32])
    _sig(lib, "sol_kernel_timing", C.c_int, [P, C.c_int])
    _sig(lib, "sol_last_kernel_ms", C.c_int, [P, C.POINTER(C.c_float), C.POINTER(C.c_uint32)])
    _sig(lib, "sol_eval", C.c_int, [C.c_int, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32])
    _sig(lib, "sol_scene_create_ex", C.c_int, [C.POINTER(SolSceneDesc), C.c_int, C.POINTER(SolCreateOptions), C.POINTER(P)])
    _sig(lib, "sol_scene_build_times", C.c_int, [P, C.POINTER(C.c_double)])
    _sig(lib, "sol_scene_set_option", C.c_int, [P, C.c_int, C.c_int64])
    _sig(lib, "sol_background_blocks", C.c_int, [P, C.c_int, P, C.c_size_t, C.POINTER(C.c_uint32)])
    _sig(lib, "sol_scene_info", C.c_int, [P, C.POINTER(SolSceneInfo)])
    _sig(lib, "sol_path_stats", C.c_int, [P, C.POINTER(SolPathStats)])
    _sig(lib, "sol_comm_unique_id", C.c_int, [C.POINTER(C.c_uint8)])
    _sig(lib, "sol_comm_init", C.c_int, [P, C.c_int, C.c_int, C.POINTER(C.c_uint8)])
    _sig(lib, "sol_comm_destroy", C.c_int, [P])
    _sig(lib, "sol_gather", C.c_int, [P, C.c_void_p])
    _sig(lib, "sol_comm_self_check", C.c_int, [P])
    _sig(lib, "sol_gather_local", C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)])
    _sig(lib, "sol_read_image", C.c_int, [P, C.POINTER(C.c_float)])
    _sig(lib, "sol_max_samples_per_call", C.c_uint32, [P])
    _libs["hip"] = lib
    return lib


HIP_SYMBOLS = ["sol_device_count", "sol_scene_create", "sol_scene_destroy", "sol_scene_set_partition",
               "sol_accum_floats", "sol_accum_ptr", "sol_scene_bind_accum", "sol_scene_set_stream", "sol_clear",
               "sol_render", "sol_render_counted", "sol_sync", "sol_read", "sol_unpermute", "sol_tonemap_rgb8",
               "sol_stats", "sol_record_sizes", "sol_last_error", "sol_eval", "sol_kernel_timing", "sol_last_kernel_ms",
               "sol_debug_path", "sol_resolve_image", "sol_bloom", "sol_bloom_rgb8", "sol_gaussian_blur_weights", "sol_world_tree_check", "sol_world_tree_check_ex", "sol_render_aux", "sol_clear_aux", "sol_read_aux",
               "sol_scene_create_ex", "sol_scene_build_times", "sol_scene_set_option", "sol_scene_info", "sol_path_stats", "sol_comm_unique_id", "sol_comm_init",
               "sol_comm_destroy", "sol_gather", "sol_gather_local", "sol_comm_self_check", "sol_read_image", "sol_max_samples_per_call", "sol_background_blocks"]


def load_host():
    if "host" in _libs:
        return _libs["host"]
    load_hip()  # libsolstrale_host.so links against the device library (ray_trace)
    if not os.path.exists(HOST_LIB):
        raise RuntimeError(f"{HOST_LIB} is missing: run `python __graft_entry__.py build`")
    lib = C.CDLL(HOST_LIB)
    B = C.c_void_p
    I = C.c_int
    D = C.c_double
    _sig(lib, "solh_builder_new", B, [])
    _sig(lib, "solh_builder_free", None, [B])
    _sig(lib, "solh_last_error", C.c_char_p, [])
    _sig(lib, "solh_transform", I, [B, I, C.POINTER(C.c_int), _D3])
    _sig(lib, "solh_solid_color", I, [B, D, D, D])
    _sig(lib, "solh_image_map", I, [B, C.c_uint32, C.c_uint32, C.c_void_p])
    _sig(lib, "solh_normal_texture", I, [B, C.c_uint32, C.c_uint32, C.c_void_p])
    _sig(lib, "solh_lambertian", I, [B, I, I])
    _sig(lib, "solh_metal", I, [B, I, I, D])
    _sig(lib, "solh_dielectric", I, [B, I, I, D])
    _sig(lib, "solh_diffuse_light", I, [B, D, D, D, D])
    _sig(lib, "solh_blend", I, [B, I, I, D])
    _sig(lib, "solh_sphere", I, [B, _D3, D, I])
    _sig(lib, "solh_quad", I, [B, _D3, _D3, _D3, I, I])
    _sig(lib, "solh_box", I, [B, _D3, _D3, I, I])
    _sig(lib, "solh_triangle", I, [B, _D3, _D3, _D3, C.POINTER(C.c_float), I, I])
    _sig(lib, "solh_triangles", I, [B, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, I])
    _sig(lib, "solh_spheres", I, [B, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p])
    _sig(lib, "solh_constant_medium", I, [B, I, D, _D3])
    _sig(lib, "solh_bvh", I, [B, I, C.POINTER(C.c_int)])
    _sig(lib, "solh_bvh_range", I, [B, I, I])
    _sig(lib, "solh_finish", C.POINTER(SolSceneDesc),
         [B, I, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _D3, D, D, _D3, _D3, _D3])
    _sig(lib, "solh_environment", I, [B, C.c_uint32, C.c_uint32, C.c_void_p, D])
    _sig(lib, "solh_tree_depth", C.c_uint32, [B])
    _sig(lib, "solh_ray_trace", I, [B, C.c_uint32, C.c_uint64, I, D, I, PROGRESS_FN, ABORT_FN, C.c_void_p])
    _sig(lib, "solh_ray_trace_devices", I, [B, C.c_uint32, C.c_uint64, I, D, I, C.POINTER(C.c_int), PROGRESS_FN, ABORT_FN, C.c_void_p])
    _sig(lib, "solh_load_obj", I, [B, C.c_char_p, C.c_char_p, I, I, IMAGE_DECODER_FN, C.c_void_p])
    _sig(lib, "solh_set_post_processors", I, [B, I, C.POINTER(C.c_int), C.POINTER(C.c_double)])
    _sig(lib, "solh_abi_sizes", None, [C.POINTER(C.c_uint32)])
    _sig(lib, "solh_to_rgb_color", None, [_D3, C.c_uint32, C.POINTER(C.c_uint8)])
    _libs["host"] = lib
    return lib


HOST_SYMBOLS = ["solh_builder_new", "solh_builder_free", "solh_last_error", "solh_transform", "solh_solid_color",
                "solh_image_map", "solh_normal_texture", "solh_lambertian", "solh_metal", "solh_dielectric",
                "solh_diffuse_light", "solh_blend", "solh_sphere", "solh_quad", "solh_box", "solh_triangle",
                "solh_triangles", "solh_spheres", "solh_constant_medium", "solh_bvh", "solh_bvh_range", "solh_finish",
                "solh_tree_depth", "solh_environment", "solh_ray_trace", "solh_ray_trace_devices", "solh_abi_sizes", "solh_to_rgb_color", "solh_set_post_processors", "solh_load_obj"]


def d3(v):
    return (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))
