"""Python face of the C++ host mirror (libsolstrale_host.so) with the reference's constructor names.

The reference's tests build scenes with `Quad::new(..)`, `Sphere::new(..)`, `Bvh::new(..)`, `Lambertian::new(..)`
(tests/scenes.rs); the same calls exist here on a `SceneBuilder`, so parity tests read like the reference's own.
All geometry work (transforms, boxes, BVH build, flattening, Camera::new) happens in the C++ host.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _abi


class HostError(RuntimeError):
    pass


# transformation.rs
def Translation(v):
    return (0, (float(v[0]), float(v[1]), float(v[2])))


def RotationX(deg):
    return (1, (float(deg), 0.0, 0.0))


def RotationY(deg):
    return (2, (float(deg), 0.0, 0.0))


def RotationZ(deg):
    return (3, (float(deg), 0.0, 0.0))


def Scale(s):
    return (4, (float(s), 0.0, 0.0))


class CameraConfig:
    """src/camera.rs:8-31"""

    def __init__(self, vertical_fov_degrees=50.0, aperture_size=0.0, look_from=(0, 0, 0), look_at=(0, 0, 0),
                 up=(0, 1, 0)):
        self.vertical_fov_degrees = vertical_fov_degrees
        self.aperture_size = aperture_size
        self.look_from = look_from
        self.look_at = look_at
        self.up = up


def NopPostProcessor():
    """src/post/nop.rs:11-17"""
    return (0, (0., 0., 0.))


def OidnPostProcessor():
    """src/post/oidn.rs:85-128: the crate's default build (no `oidn-postprocessor` feature) makes this the Nop post-processor."""
    return NopPostProcessor()


def BloomPostProcessor(kernel_size_fraction, threshold=None, max_intensity=None):
    """src/post/bloom.rs:27-47 (None = the reference's default; the range check happens where the chain is installed)."""
    nan = float("nan")
    return (1, (float(kernel_size_fraction), nan if threshold is None else float(threshold),
                nan if max_intensity is None else float(max_intensity)))


class RenderConfig:
    """src/renderer/mod.rs:26-52 (seed is the build's addition; OidnPostProcessor is not built)."""

    def __init__(self, width=300, height=200, samples_per_pixel=50, shader=(_abi.SHADER_PATH_TRACING, 50),
                 seed=0x5017A1E, post_processors=None):
        self.width = width
        self.height = height
        self.samples_per_pixel = samples_per_pixel
        self.shader = shader
        self.seed = seed
        self.post_processors = [NopPostProcessor()] if post_processors is None else list(post_processors)


def PathTracingShader(max_depth):
    return (_abi.SHADER_PATH_TRACING, max_depth)


def AlbedoShader():
    return (_abi.SHADER_ALBEDO, 0)


def NormalShader():
    return (_abi.SHADER_NORMAL, 0)


def SimpleShader():
    return (_abi.SHADER_SIMPLE, 0)


class Scene:
    """A finished scene: owns the builder (and therefore the memory behind `desc`)."""

    def __init__(self, builder, desc_ptr, render_config):
        self._builder = builder
        self.desc_ptr = desc_ptr
        self.desc = desc_ptr.contents
        self.render_config = render_config
        self.tree_depth = int(builder.lib.solh_tree_depth(builder.h))

    @property
    def width(self):
        return int(self.desc.width)

    @property
    def height(self):
        return int(self.desc.height)

    def ray_trace(self, strategy="only_final", interval_seconds=0.0, device=0, abort=None, on_progress=None, devices=None):
        """`ray_trace(scene, output, abort)` (src/lib.rs:93-99). Returns (list of progress tuples, last image). `devices` (a list of device
        ordinals, an ordinal may repeat): the same from this one process on several GPUs (solh_ray_trace_devices)."""
        b = self._builder
        rc = self.render_config
        events = []
        last = [None]

        def _progress(_user, progress, fps, eta, img, w, h):
            image = None
            if img:
                image = np.ctypeslib.as_array(img, shape=(h, w, 3)).copy()
                last[0] = image
            events.append((progress, fps, eta, image is not None))
            if on_progress:
                on_progress(progress, fps, eta, image)

        def _abort(_user):
            return 1 if (abort and abort()) else 0

        cb = _abi.PROGRESS_FN(_progress)
        ab = _abi.ABORT_FN(_abort)
        strat = {"every_sample": 0, "interval": 1, "only_final": 2}[strategy]
        pp = rc.post_processors
        kinds = (C.c_int * max(1, len(pp)))(*[k for k, _ in pp])
        params = (C.c_double * max(1, 3 * len(pp)))(*[x for _, prm in pp for x in prm])
        if b.lib.solh_set_post_processors(b.h, len(pp), kinds, params) != 0:
            raise HostError(b.lib.solh_last_error().decode(errors="replace"))
        if devices is None:
            rc_ = b.lib.solh_ray_trace(b.h, rc.samples_per_pixel, rc.seed, strat, interval_seconds, device, cb, ab, None)
        else:
            ids = (C.c_int * max(1, len(devices)))(*devices)
            rc_ = b.lib.solh_ray_trace_devices(b.h, rc.samples_per_pixel, rc.seed, strat, interval_seconds, len(devices), ids, cb, ab, None)
        if rc_ != 0:
            raise HostError(b.lib.solh_last_error().decode(errors="replace"))
        return events, last[0]


class SceneBuilder:
    def __init__(self):
        self.lib = _abi.load_host()
        self.h = self.lib.solh_builder_new()

    def __del__(self):
        try:
            if self.h:
                self.lib.solh_builder_free(self.h)
                self.h = None
        except Exception:
            pass

    def _chk(self, r):
        if r < 0:
            raise HostError(self.lib.solh_last_error().decode(errors="replace"))
        return r

    def _tf(self, ops):
        if not ops:
            return -1
        if isinstance(ops, tuple) and isinstance(ops[0], int):
            ops = [ops]
        kinds = (C.c_int * len(ops))(*[o[0] for o in ops])
        params = (C.c_double * (3 * len(ops)))(*[x for o in ops for x in o[1]])
        return self._chk(self.lib.solh_transform(self.h, len(ops), kinds, params))

    # textures
    def SolidColor(self, r, g, b):
        return self._chk(self.lib.solh_solid_color(self.h, r, g, b))

    def ImageMap(self, rgb8):
        a = np.ascontiguousarray(rgb8, dtype=np.uint8)
        h, w, _ = a.shape
        return self._chk(self.lib.solh_image_map(self.h, w, h, a.ctypes.data))

    def load_normal_texture(self, rgb8):
        a = np.ascontiguousarray(rgb8, dtype=np.uint8)
        h, w, _ = a.shape
        return self._chk(self.lib.solh_normal_texture(self.h, w, h, a.ctypes.data))

    # materials
    def Lambertian(self, albedo, normal=None):
        return self._chk(self.lib.solh_lambertian(self.h, albedo, -1 if normal is None else normal))

    def Metal(self, albedo, normal, fuzz):
        return self._chk(self.lib.solh_metal(self.h, albedo, -1 if normal is None else normal, fuzz))

    def Dielectric(self, albedo, normal, index_of_refraction):
        return self._chk(self.lib.solh_dielectric(self.h, albedo, -1 if normal is None else normal,
                                                  index_of_refraction))

    def DiffuseLight(self, r, g, b, attenuation_half_length=None):
        a = math.nan if attenuation_half_length is None else float(attenuation_half_length)
        return self._chk(self.lib.solh_diffuse_light(self.h, r, g, b, a))

    def Blend(self, m1, m2, blend_factor):
        return self._chk(self.lib.solh_blend(self.h, m1, m2, blend_factor))

    # hittables
    def Sphere(self, center, radius, mat):
        return self._chk(self.lib.solh_sphere(self.h, _abi.d3(center), radius, mat))

    def Quad(self, q, u, v, mat, transformation=None):
        return self._chk(self.lib.solh_quad(self.h, _abi.d3(q), _abi.d3(u), _abi.d3(v), mat, self._tf(transformation)))

    def new_box(self, a, b, mat, transformation=None):
        first = self._chk(self.lib.solh_box(self.h, _abi.d3(a), _abi.d3(b), mat, self._tf(transformation)))
        return list(range(first, first + 6))

    def Triangle(self, v0, v1, v2, mat, transformation=None, uv=None):
        uvp = None
        if uv is not None:
            uvp = (C.c_float * 6)(*[float(x) for p in uv for x in p])
        return self._chk(self.lib.solh_triangle(self.h, _abi.d3(v0), _abi.d3(v1), _abi.d3(v2), uvp, mat,
                                                self._tf(transformation)))

    def triangles(self, vertices, materials, uvs=None, transformation=None):
        """Bulk Triangle::new_with_tex_coords: vertices (n,3,3) f64, materials (n,) int32, uvs (n,3,2) f32."""
        v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 9)
        n = v.shape[0]
        m = np.ascontiguousarray(np.broadcast_to(np.asarray(materials, dtype=np.int32), (n,)))
        up = None
        if uvs is not None:
            u = np.ascontiguousarray(uvs, dtype=np.float32).reshape(n, 6)
            up = u.ctypes.data
        first = self._chk(self.lib.solh_triangles(self.h, n, v.ctypes.data, up, m.ctypes.data, self._tf(transformation)))
        return first, n

    def spheres(self, centers, radii, materials):
        c = np.ascontiguousarray(centers, dtype=np.float64).reshape(-1, 3)
        n = c.shape[0]
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(radii, dtype=np.float64), (n,)))
        m = np.ascontiguousarray(np.broadcast_to(np.asarray(materials, dtype=np.int32), (n,)))
        first = self._chk(self.lib.solh_spheres(self.h, n, c.ctypes.data, r.ctypes.data, m.ctypes.data))
        return first, n

    def load_obj(self, path, filename, transformation=None, default_material=None):
        """`Obj::new(path, filename).load(transformation, default_material)` (src/loader/obj.rs:29-136) -> hittable id of the
        model's Bvh. Image files are decoded here with Pillow (the reference: `image` crate) and handed over as RGB8."""
        from PIL import Image
        keep = []

        def _decode(_user, cpath, pw, ph, pdata):
            try:
                fn = os.fsdecode(cpath)  # (a file name is bytes: an MTL may name one that is not UTF-8)
                if not os.path.isfile(fn):
                    return 1
            except (ValueError, OSError):  # (an embedded NUL, a name longer than the file system takes)
                return 1
            try:
                a = np.ascontiguousarray(np.asarray(Image.open(fn).convert("RGB"), dtype=np.uint8))
            except Exception:
                return 2
            keep.append(a)
            pw[0], ph[0] = a.shape[1], a.shape[0]
            pdata[0] = a.ctypes.data_as(C.POINTER(C.c_uint8))
            return 0

        cb = _abi.IMAGE_DECODER_FN(_decode)
        return self._chk(self.lib.solh_load_obj(self.h, path.encode(), filename.encode(), self._tf(transformation),
                                                -1 if default_material is None else default_material, cb, None))

    def ConstantMedium(self, boundary, density, color):
        return self._chk(self.lib.solh_constant_medium(self.h, boundary, density, _abi.d3(color)))

    def Bvh(self, hittables):
        ids = (C.c_int * len(hittables))(*hittables)
        return self._chk(self.lib.solh_bvh(self.h, len(hittables), ids))

    def Bvh_range(self, first, n):
        return self._chk(self.lib.solh_bvh_range(self.h, first, n))

    def environment(self, rgb32f, scale=1.0):
        """EXTENSION (not in the reference): a latitude-longitude map of linear radiance, (H, W, 3) float32, row 0 = up, that
        rays which hit nothing return instead of the constant background colour. Call before finish()."""
        a = np.ascontiguousarray(rgb32f, dtype=np.float32)
        h, w, _ = a.shape
        self._chk(self.lib.solh_environment(self.h, w, h, a.ctypes.data, float(scale)))

    def finish(self, world, camera, background_color, render_config):
        """`Scene{world, camera, background_color, render_config}` -> flattened description."""
        p = self.lib.solh_finish(self.h, world, render_config.width, render_config.height, render_config.shader[0],
                                 render_config.shader[1], _abi.d3(background_color), camera.vertical_fov_degrees,
                                 camera.aperture_size, _abi.d3(camera.look_from), _abi.d3(camera.look_at),
                                 _abi.d3(camera.up))
        if not p:
            raise HostError(self.lib.solh_last_error().decode(errors="replace"))
        return Scene(self, p, render_config)
