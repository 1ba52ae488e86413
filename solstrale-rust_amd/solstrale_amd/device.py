"""Thin Python handle over the device C ABI (libsolstrale_hip.so). No compute happens in Python and there is no
fallback: every call goes to the HIP library and raises if it fails (e.g. SOL_EDEVICE without a GPU)."""
import ctypes as C

import numpy as np

from . import _abi


class DeviceError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code
        self.msg = msg


def device_count():
    return int(_abi.load_hip().sol_device_count())


def comm_unique_id():
    """sol_comm_unique_id: 128 bytes rank 0 ships to the other ranks before sol_comm_init."""
    lib = _abi.load_hip()
    buf = (C.c_uint8 * _abi.UNIQUE_ID_BYTES)()
    rc = lib.sol_comm_unique_id(buf)
    if rc != 0:
        raise DeviceError(rc, lib.sol_last_error().decode(errors="replace"))
    return bytes(buf)


def record_sizes():
    out = (C.c_uint32 * 6)()
    _abi.load_hip().sol_record_sizes(out)
    return dict(zip(("node", "sphere", "quad", "triangle", "triangle_shade", "material"), [int(x) for x in out]))


def eval_functions(fn, rows, out_cols, device=0):
    """sol_eval: device functions on rows of fp32 inputs (function-level parity tests)."""
    lib = _abi.load_hip()
    a = np.ascontiguousarray(rows, dtype=np.float32)
    out = np.zeros((a.shape[0], out_cols), dtype=np.float32)
    rc = lib.sol_eval(device, fn, a.ctypes.data, a.shape[0], a.shape[1], out.ctypes.data, out_cols)
    if rc != 0:
        raise DeviceError(rc, lib.sol_last_error().decode(errors="replace"))
    return out


def world_tree_check(scene, use_sah):
    """Host-only structural check of the 8-wide tree sol_scene_create would build for `scene` (no device needed)."""
    lib = _abi.load_hip()
    out = _abi.SolTreeCheck()
    rc = lib.sol_world_tree_check_ex(scene.desc_ptr, int(use_sah), C.byref(out), C.sizeof(out))  # (the size-prefixed form: every field)
    if rc != 0:
        raise DeviceError(rc, lib.sol_last_error().decode(errors="replace"))
    return out.as_dict()


def background_blocks(scene, use_sah=0):
    """Host-only: which 8x8 pixel blocks of `scene` provably see nothing but the background (sol_background_blocks): a bool array
    [blocks_y, blocks_x]."""
    import numpy as np
    lib = _abi.load_hip()
    bx, by = (scene.width + 7) // 8, (scene.height + 7) // 8
    flags = (C.c_uint8 * (bx * by))()
    n = C.c_uint32()
    rc = lib.sol_background_blocks(scene.desc_ptr, int(use_sah), flags, bx * by, C.byref(n))
    if rc != 0:
        raise DeviceError(rc, lib.sol_last_error().decode(errors="replace"))
    out = np.frombuffer(flags, dtype=np.uint8).reshape(by, bx).astype(bool)
    assert int(out.sum()) == n.value
    return out


class DeviceScene:
    """sol_scene_create .. sol_scene_destroy"""

    def __init__(self, scene, device=0, world_tree=None, no_work_order_probe=False, split_percent=0, no_background_blocks=False):
        """split_percent: SolCreateOptions.split_percent (0: the default budget of the device build's triangle pre-splitting, < 0: none);
        no_background_blocks: SolCreateOptions.no_background_blocks (do not look for blocks that provably see only the background)."""
        self.lib = _abi.load_hip()
        self.scene = scene
        self.h = C.c_void_p()
        if world_tree is None and not no_work_order_probe and not split_percent and not no_background_blocks:
            rc = self.lib.sol_scene_create(scene.desc_ptr, device, C.byref(self.h))
        else:
            opt = _abi.SolCreateOptions(size=C.sizeof(_abi.SolCreateOptions), world_tree=int(world_tree or 0),
                                        no_work_order_probe=1 if no_work_order_probe else 0, split_percent=int(split_percent),
                                        no_background_blocks=1 if no_background_blocks else 0)
            rc = self.lib.sol_scene_create_ex(scene.desc_ptr, device, C.byref(opt), C.byref(self.h))
        if rc != 0:
            self.h = None
            raise DeviceError(rc, self.lib.sol_last_error().decode(errors="replace"))
        self.width, self.height = scene.width, scene.height

    def _chk(self, rc):
        if rc != 0:
            raise DeviceError(rc, self.lib.sol_last_error().decode(errors="replace"))

    def close(self):
        if self.h:
            self.lib.sol_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_partition(self, rank, world):
        self._chk(self.lib.sol_scene_set_partition(self.h, rank, world))

    def set_option(self, option, value):
        self._chk(self.lib.sol_scene_set_option(self.h, int(option), int(value)))

    def build_times(self):
        """Seconds sol_scene_create spent in host tree candidates / uploads / device tree build / probe renders."""
        out = (C.c_double * 4)()
        self._chk(self.lib.sol_scene_build_times(self.h, out))
        return dict(zip(("host_trees", "upload", "device_tree", "probes"), [float(x) for x in out]))

    def info(self):
        """sol_scene_info: the world tree in use (and why, when it is not the one asked for), the bound on the traversal stack."""
        r = _abi.SolSceneInfo()
        r.size = C.sizeof(r)
        self._chk(self.lib.sol_scene_info(self.h, C.byref(r)))
        return {"stack_bound": int(r.stack_bound), "lds_stack": int(r.lds_stack), "spill_stack": int(r.spill_stack),
                "tree_fallback": bool(r.tree_fallback), "tree_name": r.tree_name.decode(), "tree_note": r.tree_note.decode(),
                "split_references": int(r.split_references), "split_triangles": int(r.split_triangles), "split_area_ratio": float(r.split_area_ratio),
                "reinsertion_moves": int(r.reinsertion_moves), "reinsertion_area_ratio": float(r.reinsertion_area_ratio),
                "partition_table": int(r.partition_table), "partition_crc": int(r.partition_crc), "strict_triangles": bool(r.strict_triangles),
                "background_blocks": int(r.background_blocks), "background_pixels": int(r.background_pixels)}

    def path_stats(self):
        """sol_path_stats of the last counted render: primary hit fraction and the path-length histogram (shares of the samples)."""
        r = _abi.SolPathStats()
        r.size = C.sizeof(r)
        self._chk(self.lib.sol_path_stats(self.h, C.byref(r)))
        n = max(1, int(r.samples))
        bins = ("1", "2", "3-4", "5-8", "9-16", "17+")
        return {"samples": int(r.samples), "primary_hit_fraction": int(r.primary_hits) / n,
                "rays_per_path_histogram": {b: int(r.path_len[k]) / n for k, b in enumerate(bins)}}

    def max_samples_per_call(self):
        return int(self.lib.sol_max_samples_per_call(self.h))

    def comm_init(self, rank, world, unique_id):
        """sol_comm_init: RCCL communicator of the tile partition (collective over the ranks); also sets the partition."""
        buf = (C.c_uint8 * _abi.UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._chk(self.lib.sol_comm_init(self.h, rank, world, buf))

    def comm_destroy(self):
        self._chk(self.lib.sol_comm_destroy(self.h))

    def gather(self, image_ptr=0):
        """sol_gather: collective; rank 0 gets the row-major image in device memory `image_ptr` (0: the scene's own buffer)."""
        self._chk(self.lib.sol_gather(self.h, C.c_void_p(image_ptr)))

    def comm_self_check(self):
        self._chk(self.lib.sol_comm_self_check(self.h))

    def read_image(self):
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._chk(self.lib.sol_read_image(self.h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def set_stream(self, hip_stream):
        self._chk(self.lib.sol_scene_set_stream(self.h, C.c_void_p(hip_stream)))

    def accum_floats(self):
        return int(self.lib.sol_accum_floats(self.h))

    def accum_ptr(self):
        return int(self.lib.sol_accum_ptr(self.h) or 0)

    def bind_accum(self, device_ptr, n_floats):
        self._chk(self.lib.sol_scene_bind_accum(self.h, C.c_void_p(device_ptr), n_floats))

    def clear(self):
        self._chk(self.lib.sol_clear(self.h))

    def render(self, first_sample, n_samples, seed, counted=False):
        f = self.lib.sol_render_counted if counted else self.lib.sol_render
        self._chk(f(self.h, first_sample, n_samples, seed))

    def sync(self):
        self._chk(self.lib.sol_sync(self.h))

    def read(self):
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._chk(self.lib.sol_read(self.h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def render_aux(self, first_sample, n_samples, seed):
        """Adds the samples' first-hit albedo and normal to the auxiliary accumulators (renderer/mod.rs:175-204)."""
        self._chk(self.lib.sol_render_aux(self.h, first_sample, n_samples, seed))

    def clear_aux(self):
        self._chk(self.lib.sol_clear_aux(self.h))

    def read_aux(self):
        """(albedo sums, normal sums), each (H, W, 3) float32, row 0 = top."""
        a = np.empty((self.height, self.width, 3), dtype=np.float32)
        n = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._chk(self.lib.sol_read_aux(self.h, a.ctypes.data_as(C.POINTER(C.c_float)), n.ctypes.data_as(C.POINTER(C.c_float))))
        return a, n

    def unpermute(self, gathered_ptr, world, image_ptr):
        self._chk(self.lib.sol_unpermute(self.h, C.c_void_p(gathered_ptr), world, C.c_void_p(image_ptr)))

    def tonemap_rgb8(self, image_ptr, num_samples):
        out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        self._chk(self.lib.sol_tonemap_rgb8(self.h, C.c_void_p(image_ptr), num_samples,
                                            out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def resolve_image(self):
        """Device pointer of the scene's own row-major image (W*H*3 floats) after un-permuting its accumulators."""
        p = C.c_void_p()
        self._chk(self.lib.sol_resolve_image(self.h, C.byref(p)))
        return p.value

    BLOOM_DEFAULT_THRESHOLD = 3.0 ** 0.5  # Vec3::new(1., 1., 1.).length() (bloom.rs:39)
    BLOOM_DEFAULT_MAX = 1.7976931348623157e308  # f64::MAX (bloom.rs:40)

    def bloom(self, image_ptr, num_samples, kernel_size_fraction, threshold=None, max_intensity=None):
        """BloomPostProcessor::intermediate_post_process on the device image, in place."""
        self._chk(self.lib.sol_bloom(self.h, C.c_void_p(image_ptr), num_samples, kernel_size_fraction,
                                     self.BLOOM_DEFAULT_THRESHOLD if threshold is None else threshold,
                                     self.BLOOM_DEFAULT_MAX if max_intensity is None else max_intensity))

    def bloom_rgb8(self, image_ptr, num_samples, kernel_size_fraction, threshold=None, max_intensity=None):
        """BloomPostProcessor::post_process of the device image -> RGB8 (H, W, 3)."""
        out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        self._chk(self.lib.sol_bloom_rgb8(self.h, C.c_void_p(image_ptr), num_samples, kernel_size_fraction,
                                          self.BLOOM_DEFAULT_THRESHOLD if threshold is None else threshold,
                                          self.BLOOM_DEFAULT_MAX if max_intensity is None else max_intensity,
                                          out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def debug_path(self, x, y, sample, seed, max_rows=80):
        """Rays of one path: rows of (o xyz, d xyz, t, ref bits, dfs bits, depth, 0, 0); returns (rows, colour)."""
        buf = np.zeros((max_rows, 12), dtype=np.float32)
        self._chk(self.lib.sol_debug_path(self.h, x, y, sample, seed, buf.ctypes.data, max_rows))
        n = int(np.argmax(buf[:, 3] == -1.0)) if (buf[:, 3] == -1.0).any() else max_rows - 1
        return buf[:n], buf[n, :3].copy()

    def kernel_timing(self, enable=True):
        self._chk(self.lib.sol_kernel_timing(self.h, 1 if enable else 0))

    def last_kernel_ms(self):
        ms, grid = C.c_float(), C.c_uint32()
        self._chk(self.lib.sol_last_kernel_ms(self.h, C.byref(ms), C.byref(grid)))
        return float(ms.value), int(grid.value)

    def phase_stats(self):
        st = _abi.SolStats()
        self._chk(self.lib.sol_stats(self.h, C.byref(st)))
        return st.phases()

    def stats(self):
        st = _abi.SolStats()
        self._chk(self.lib.sol_stats(self.h, C.byref(st)))
        return st.as_dict()
