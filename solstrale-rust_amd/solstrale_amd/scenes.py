"""Scene definitions used by tests/ and bench.py.

Two groups:
  * the reference's own test scenes, restated call for call from tests/scenes.rs (citations per function); they need
    the reference's texture files, which live as data fixtures under tests/golden/resources/;
  * the BASELINE.json workloads, which do not exist in the reference and are authored here from the reference's
    constructors: C1 Cornell box, C2 Cornell + 10 000 spheres, C3 "Sponza-class" procedural atrium (the real
    sponza.obj is not available offline), all synthetic and deterministic.
"""
import math
import os

import numpy as np

from . import _abi
from .host import (CameraConfig, RenderConfig, RotationY, SceneBuilder, Translation)

RESOURCES = os.path.join(_abi.ROOT, "tests", "golden", "resources")


def load_image(name):
    from PIL import Image
    return np.asarray(Image.open(os.path.join(RESOURCES, name)).convert("RGB"), dtype=np.uint8)


# ---- deterministic numbers for procedural content (independent of numpy's generators) -----------------------
def _mix32(x):
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x21F0AAAD) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x735A2D97) & 0xFFFFFFFF
    x ^= x >> 15
    return x


def counter_uniform(seed, n, stream=0):
    """n numbers in [0,1): u_i = (mix32(mix32(i + seed*0x9E3779B9) ^ stream*0x85EBCA6B) >> 8) / 2^24."""
    i = np.arange(n, dtype=np.uint64)
    h = _mix32(_mix32((i + seed * 0x9E3779B9) & 0xFFFFFFFF) ^ ((stream * 0x85EBCA6B) & 0xFFFFFFFF))
    return (h >> 8).astype(np.float64) / 16777216.0


# ---------------------------------------------------------------------------------------------------------------
# BASELINE workloads
# ---------------------------------------------------------------------------------------------------------------
def _cornell_world(b):
    """RTIOW 'Cornell box' dimensions with the reference's Quad::new / Quad::new_box / DiffuseLight semantics
    (SURVEY.md 8d): 5 walls + 1 light quad + 2 boxes = 18 quads."""
    red = b.Lambertian(b.SolidColor(.65, .05, .05))
    white = b.Lambertian(b.SolidColor(.73, .73, .73))
    green = b.Lambertian(b.SolidColor(.12, .45, .15))
    light = b.DiffuseLight(15., 15., 15.)
    world = [
        b.Quad((555, 0, 0), (0, 555, 0), (0, 0, 555), green),
        b.Quad((0, 0, 0), (0, 555, 0), (0, 0, 555), red),
        b.Quad((343, 554, 332), (-130, 0, 0), (0, 0, -105), light),
        b.Quad((0, 0, 0), (555, 0, 0), (0, 0, 555), white),
        b.Quad((555, 555, 555), (-555, 0, 0), (0, 0, -555), white),
        b.Quad((0, 0, 555), (555, 0, 0), (0, 555, 0), white),
    ]
    world += b.new_box((0, 0, 0), (165, 330, 165), white, [RotationY(15.), Translation((265, 0, 295))])
    world += b.new_box((0, 0, 0), (165, 165, 165), white, [RotationY(-18.), Translation((130, 0, 65))])
    return world


_CORNELL_CAMERA = dict(vertical_fov_degrees=40., aperture_size=0., look_from=(278, 278, -800), look_at=(278, 278, 0),
                       up=(0, 1, 0))


def cornell_box(render_config=None):
    """C1: Cornell box, 400x400, 50 spp, PathTracingShader(50)."""
    rc = render_config or RenderConfig(width=400, height=400, samples_per_pixel=50)
    b = SceneBuilder()
    world = _cornell_world(b)
    return b.finish(b.Bvh(world), CameraConfig(**_CORNELL_CAMERA), (0., 0., 0.), rc)


def cornell_spheres(render_config=None, n_spheres=10000, seed=1):
    """C2: Cornell box + n Lambertian spheres, radius U[3,8], centres uniform in the box interior, albedo U[.1,.9]^3."""
    rc = render_config or RenderConfig(width=1920, height=1080, samples_per_pixel=256)
    b = SceneBuilder()
    world = _cornell_world(b)
    r = 3. + 5. * counter_uniform(seed, n_spheres, 0)
    c = np.stack([20. + 515. * counter_uniform(seed, n_spheres, 1 + k) for k in range(3)], axis=1)
    alb = np.stack([.1 + .8 * counter_uniform(seed, n_spheres, 4 + k) for k in range(3)], axis=1)
    mats = np.array([b.Lambertian(b.SolidColor(*alb[i])) for i in range(n_spheres)], dtype=np.int32)
    first, n = b.spheres(c, r, mats)
    world += list(range(first, first + n))
    return b.finish(b.Bvh(world), CameraConfig(**_CORNELL_CAMERA), (0., 0., 0.), rc)


# ---- C3: "Sponza-class" procedural atrium ---------------------------------------------------------------------
def _grid(f, nu, nv, tile=(1., 1.)):
    """Tessellates the parametric surface f(u, v) -> (x, y, z), u,v in [0,1], into 2*nu*nv triangles."""
    u = np.linspace(0., 1., nu + 1)
    v = np.linspace(0., 1., nv + 1)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    p = np.stack(f(uu, vv), axis=-1)  # (nu+1, nv+1, 3)
    t = np.stack([uu * tile[0], vv * tile[1]], axis=-1)
    a, bq, c, d = p[:-1, :-1], p[1:, :-1], p[1:, 1:], p[:-1, 1:]
    ta, tb, tc, td = t[:-1, :-1], t[1:, :-1], t[1:, 1:], t[:-1, 1:]
    tri = np.concatenate([np.stack([a, bq, c], axis=2).reshape(-1, 3, 3), np.stack([a, c, d], axis=2).reshape(-1, 3, 3)])
    uv = np.concatenate([np.stack([ta, tb, tc], axis=2).reshape(-1, 3, 2), np.stack([ta, tc, td], axis=2).reshape(-1, 3, 2)])
    return tri, uv.astype(np.float32)


def _procedural_texture(kind, size, seed):
    """RGB8 textures (brick / checker / marble-like noise), deterministic."""
    y, x = np.mgrid[0:size, 0:size].astype(np.float64) / size
    n = counter_uniform(seed, size * size, 9).reshape(size, size)
    if kind == 0:  # brick
        row = np.floor(y * 16)
        xs = (x * 8 + 0.5 * (row % 2)) % 1.0
        mortar = (xs < 0.06) | ((y * 16) % 1.0 < 0.1)
        base = np.stack([0.55 + 0.2 * n, 0.25 + 0.1 * n, 0.18 + 0.08 * n], -1)
        img = np.where(mortar[..., None], 0.75, base)
    elif kind == 1:  # checker
        c = ((np.floor(x * 8) + np.floor(y * 8)) % 2)[..., None]
        img = c * np.array([0.8, 0.78, 0.7]) + (1 - c) * np.array([0.25, 0.22, 0.2]) + 0.04 * (n[..., None] - .5)
    else:  # marble-like
        s = 0.5 + 0.5 * np.sin(40 * x + 12 * np.sin(9 * y + 3 * seed) + 6 * n)
        img = s[..., None] * np.array([0.75, 0.7, 0.62]) + (1 - s[..., None]) * np.array([0.45, 0.4, 0.38])
    return np.clip(img * 255., 0, 255).astype(np.uint8)


SPONZA_TRIANGLES = 262267


def _rot(axis, degrees):
    """Rotation matrix about a coordinate axis (0 = x, 1 = y, 2 = z)."""
    c, s_ = math.cos(math.radians(degrees)), math.sin(math.radians(degrees))
    m = np.eye(3)
    i, j = [(1, 2), (2, 0), (0, 1)][axis]
    m[i, i], m[i, j], m[j, i], m[j, j] = c, -s_, s_, c
    return m


def _placed(f, rot=None, origin=(0., 0., 0.), pivot=(0., 0., 0.)):
    """f(u, v) -> (x, y, z) rotated by `rot` about `pivot`, then moved by `origin`."""
    rot = np.eye(3) if rot is None else rot
    pv, og = np.asarray(pivot, float), np.asarray(origin, float)

    def g(u, v):
        p = np.stack(f(u, v), axis=-1) - pv
        q = p @ rot.T + pv + og
        return q[..., 0], q[..., 1], q[..., 2]
    return g


def _strip(p0, p1, w):
    """A flat strip from p0 to p1, `w` wide (vector): the long thin triangles of rails, cornices, rods."""
    p0, p1, w = np.asarray(p0, float), np.asarray(p1, float), np.asarray(w, float)
    return lambda u, v: tuple(p0[k] + (p1[k] - p0[k]) * u + w[k] * v for k in range(3))


def _heterogeneous_atrium_parts(L, Wd, Hh):
    """The atrium of `sponza_like` with the mesh statistics of a hand-modelled asset instead of uniform grids: a few
    hundred LARGE triangles for floor, walls, galleries and roof; LONG THIN triangles (aspect 60:1 .. 300:1) for rails, balusters,
    cornices and tie rods, the rods running diagonally through the hall; columns of moderate resolution under capitals made of
    thousands of tiny leaves; arches, drapes and banners ROTATED off the coordinate axes; ornament clusters (lion heads, vases with
    plants) whose triangles are > 1000 times smaller than the wall triangles. Returns [(surface f(u, v), nu, nv, material, uv tiling)]."""
    parts = []

    def plane(o, du, dv):
        o, du, dv = np.array(o, float), np.array(du, float), np.array(dv, float)
        return lambda u, v: tuple(o[k] + du[k] * u + dv[k] * v for k in range(3))

    # ---- 1. large surfaces: 384 triangles of 3 x 3 units and the like ----
    parts.append((plane((-L, 0, -Wd), (2 * L, 0, 0), (0, 0, 2 * Wd)), 10, 4, 0, (10, 4)))           # floor
    parts.append((plane((-L, 0, -Wd), (2 * L, 0, 0), (0, Hh, 0)), 10, 4, 1, (10, 4)))                # wall z=-W
    parts.append((plane((-L, 0, Wd), (2 * L, 0, 0), (0, Hh, 0)), 10, 4, 1, (10, 4)))                 # wall z=+W
    parts.append((plane((-L, 0, -Wd), (0, 0, 2 * Wd), (0, Hh, 0)), 4, 4, 2, (4, 4)))                 # end wall x=-L
    parts.append((plane((L, 0, -Wd), (0, 0, 2 * Wd), (0, Hh, 0)), 4, 4, 2, (4, 4)))                  # end wall x=+L
    parts.append((plane((-L, 5.5, -Wd), (2 * L, 0, 0), (0, 0, 2.0)), 10, 1, 3, (10, 1)))             # gallery floor -z
    parts.append((plane((-L, 5.5, Wd - 2.0), (2 * L, 0, 0), (0, 0, 2.0)), 10, 1, 3, (10, 1)))        # gallery floor +z
    parts.append((plane((-L, Hh, -Wd), (2 * L, 0, 0), (0, 0, 2.5)), 10, 1, 4, (10, 1)))              # roof strip -z
    parts.append((plane((-L, Hh, Wd - 2.5), (2 * L, 0, 0), (0, 0, 2.5)), 10, 1, 4, (10, 1)))         # roof strip +z
    # ---- 2. long thin triangles ----
    for side in (-1., 1.):
        zr = side * (Wd - 2.0)
        for y in (6.0, 6.2, 6.4, 6.6):                                                               # gallery rails: 7.5 x 0.05 (150:1)
            parts.append((_strip((-L, y, zr), (L, y, zr), (0, 0.05, 0)), 4, 1, 16, (4, 1)))
            parts.append((_strip((-L, y + 0.05, zr), (L, y + 0.05, zr), (0, 0, side * 0.05)), 4, 1, 16, (4, 1)))
        for i in range(120):                                                                         # balusters: 1.1 x 0.018 (61:1)
            x = -L + 0.125 + i * 0.25
            parts.append((_strip((x, 5.5, zr), (x, 6.6, zr), (0.018, 0, 0)), 1, 1, 17, (1, 1)))
        zw = side * (Wd - 0.02)
        for y in (3.0, 5.3, 8.6, 11.6):                                                              # cornices on the long walls: three faces, 5 x 0.07 (71:1)
            zf = zw - side * 0.05
            parts.append((_strip((-L, y, zf), (L, y, zf), (0, 0.07, 0)), 6, 1, 18, (6, 1)))
            parts.append((_strip((-L, y + 0.07, zf), (L, y + 0.07, zf), (0, 0, side * 0.05)), 6, 1, 18, (6, 1)))
            parts.append((_strip((-L, y, zf), (L, y, zf), (0, 0, side * 0.05)), 6, 1, 18, (6, 1)))
    for i in range(14):                                                                              # tie rods: diagonal through the hall, 12.5 x 0.04 (310:1)
        x0 = -L + 1.0 + i * 2.0
        a, bq = np.array((x0, 11.6, -Wd + 0.1)), np.array((x0 + 2.6, 10.4, Wd - 0.1))
        for w in ((0, 0.04, 0), (0.04, 0, 0), (0.03, 0.03, 0)):
            parts.append((_strip(a, bq, w), 1, 1, 19, (8, 1)))
    for i in range(10):                                                                              # banner ropes from gallery to gallery, sagging
        x0 = -L + 2.0 + i * 2.8

        def rope(u, v, x0=x0, i=i):
            return (x0 + 1.5 * u + 0.02 * v, 7.4 - 1.6 * np.sin(np.pi * u) + 0.0 * v, -(Wd - 2.0) + 2 * (Wd - 2.0) * u)
        parts.append((rope, 12, 1, 19, (12, 1)))
    # ---- 3. columns (moderate) with capitals made of tiny leaves ----
    col_x = np.linspace(-L + 1.5, L - 1.5, 10)
    for side in (-1., 1.):
        for storey in range(2):
            y0 = 0.0 if storey == 0 else 5.6
            for i, cx in enumerate(col_x):
                cz = side * (Wd - 2.0)

                def column(u, v, cx=cx, cz=cz, y0=y0):
                    r = 0.35 * (1.0 + 0.08 * np.sin(12 * np.pi * u)) * (1.0 - 0.1 * v)
                    return cx + r * np.cos(2 * np.pi * u), y0 + 3.9 * v, cz + r * np.sin(2 * np.pi * u)
                parts.append((column, 32, 16, 5 + (i % 3), (2, 3)))
                for k in range(24):                                                                   # acanthus leaves: 6 x 6 cells of ~0.02
                    ang = 2 * np.pi * k / 24

                    def leaf(u, v, cx=cx, cz=cz, y0=y0, ang=ang, k=k):
                        r = 0.33 + 0.16 * v + 0.05 * np.sin(np.pi * v) * (1 + 0.3 * np.sin(6 * np.pi * u))
                        a2 = ang + (u - 0.5) * 0.24
                        return cx + r * np.cos(a2), y0 + 3.9 + 0.3 * v - 0.05 * np.sin(np.pi * v) ** 2, cz + r * np.sin(a2)
                    parts.append((leaf, 6, 6, 8 + (k % 3), (1, 1)))
                if i + 1 < len(col_x):                                                                # arches, each yawed about its own vertical axis
                    x1 = col_x[i + 1]

                    def arch(u, v, x0=cx, x1=x1, cz=cz, y0=y0):
                        ang = np.pi * u
                        xm, rad = 0.5 * (x0 + x1), 0.5 * (x1 - x0)
                        return xm - rad * np.cos(ang), y0 + 4.2 + 0.9 * np.sin(ang), cz - 0.3 + 0.6 * v
                    yaw = (7.0 if i % 2 == 0 else -9.0) * side
                    parts.append((_placed(arch, _rot(1, yaw), pivot=(0.5 * (cx + x1), 0, cz)), 24, 4, 8 + (i % 4), (3, 1)))
    # ---- 4. drapes: dense cloth, hung askew (yawed and tilted) ----
    for i in range(6):
        x0 = -L + 3.0 + i * 4.6
        side = -1. if i % 2 == 0 else 1.

        def drape(u, v, i=i):
            return (2.2 * u, -3.0 * v + 0.1 * np.sin(6 * np.pi * u), 0.35 * np.sin(10 * np.pi * u + i) * (0.3 + v))
        r = _rot(1, 18.0 + 7.0 * i) @ _rot(0, -8.0 + 3.0 * i)
        parts.append((_placed(drape, r, origin=(x0, 5.4, side * (Wd - 2.6))), 96, 66, 12 + i, (2, 2)))
    # ---- 5. ornament clusters: lion heads on the walls, vases with plants on the floor ----
    for i in range(12):
        side = -1. if i % 2 == 0 else 1.
        cxh, cyh, czh = -L + 2.5 + (i // 2) * 5.0, 4.4 + 3.4 * ((i // 2) % 2), side * (Wd - 0.25)

        def head(u, v, c=(cxh, cyh, czh), i=i):
            th = np.pi * (0.02 + 0.96 * v)
            r = 0.25 * (1.0 + 0.18 * np.sin(7 * 2 * np.pi * u + i) * np.sin(5 * th) + 0.08 * np.sin(23 * 2 * np.pi * u) * np.sin(17 * th))
            return c[0] + r * np.sin(th) * np.cos(2 * np.pi * u), c[1] - r * np.cos(th), c[2] + r * np.sin(th) * np.sin(2 * np.pi * u)
        parts.append((head, 64, 32, 20 + (i % 4), (1, 1)))
    for i in range(8):
        cxv, czv = -L + 3.7 + i * 3.3, (-1. if i % 2 else 1.) * 1.2

        def vase(u, v, c=(cxv, czv)):
            r = 0.12 + 0.22 * np.sin(np.pi * (0.15 + 0.8 * v)) ** 2
            return c[0] + r * np.cos(2 * np.pi * u), 0.9 * v, c[1] + r * np.sin(2 * np.pi * u)
        parts.append((vase, 32, 16, 22, (1, 1)))
        for k in range(60):                                                                           # leaves: 0.9 x 0.03 strips in every direction
            az, el = 360.0 * ((k * 0.618034) % 1.0), 25.0 + 50.0 * ((k * 0.754878) % 1.0)
            d = _rot(1, az) @ _rot(2, el) @ np.array((1.0, 0.0, 0.0))

            def leafstrip(u, v, c=(cxv, czv), d=d):
                bend = -0.35 * u * u
                w = np.cross(d, (0., 1., 0.))
                w = w / (np.linalg.norm(w) + 1e-12) * 0.03
                t = (1.0 - 0.8 * u) * (v - 0.5)
                return (c[0] + 0.9 * d[0] * u + w[0] * t, 0.9 + 0.9 * d[1] * u + bend + w[1] * t, c[1] + 0.9 * d[2] * u + w[2] * t)
            parts.append((leafstrip, 4, 1, 23, (1, 1)))
    return parts


def sponza_like(render_config=None, n_triangles=SPONZA_TRIANGLES, texture_size=1024, n_materials=24, camera="default", mesh="regular"):
    """C3 stand-in: an atrium (floor, two-storey walls, galleries, 2x10x2 columns with arches, hanging drapes, partial
    roof) tessellated to exactly `n_triangles` Lambertian triangles, `n_materials` materials of which 8 carry image
    textures, one Quad light above the roof opening and a constant sky background.
    camera: "default" looks across the atrium with the roof opening and the sky in view (paths end early: ~3 rays per sample);
    "interior" stands under the -z gallery and looks along the colonnade - floor, wall, gallery floor overhead and the columns fill
    the frame, every camera ray hits, light arrives by bounces through the arches (the STRESS variant of the same geometry).
    mesh: "regular" - every surface a uniform grid of near-equal small triangles (the headline workload since round 1; the
    friendliest input a BVH builder can get); "heterogeneous" - the same hall with the triangle statistics of a hand-modelled asset
    (_heterogeneous_atrium_parts: 2-triangle-per-3-metres walls beside leaves 20 000 times smaller, rails and rods of aspect 60:1 to
    300:1, rotated drapes and arches): the STRESS mesh for the tree builder, reported beside the headline."""
    rc = render_config or RenderConfig(width=1920, height=1080, samples_per_pixel=512)
    b = SceneBuilder()
    mats = []
    for kind, value in atrium_materials(n_materials, texture_size):
        mats.append(b.Lambertian(b.ImageMap(value)) if kind == "image" else b.Lambertian(b.SolidColor(*value)))
    tri, slots, uv = atrium_mesh(n_triangles, n_materials, mesh)
    first, n = b.triangles(tri, np.asarray(mats, dtype=np.int32)[slots], uv)
    model = b.Bvh_range(first, n)  # like Obj::load -> Bvh::new(triangles) (src/loader/obj.rs:135)
    Hh = ATRIUM_HEIGHT
    light = b.Quad((-6., Hh + 1.5, -2.0), (12., 0, 0), (0, 0, 4.0), b.DiffuseLight(18., 17., 15.))
    world = b.Bvh([model, light])
    if camera == "interior":
        cam = CameraConfig(vertical_fov_degrees=60., aperture_size=0., look_from=(-14.2, 1.7, -5.0), look_at=(13.0, 2.3, -4.9), up=(0, 1, 0))
    elif camera == "default":
        cam = CameraConfig(vertical_fov_degrees=55., aperture_size=0., look_from=(-13.0, 2.2, 0.6), look_at=(6.0, 4.5, -0.4),
                           up=(0, 1, 0))
    elif camera == "far":  # a long lens 190 units above the hall, looking down through the roof opening past the light: floor, galleries, drapes
        cam = CameraConfig(vertical_fov_degrees=7.5, aperture_size=0., look_from=(47.0, 181.0, 34.0), look_at=(0.0, 3.0, 0.0), up=(0, 1, 0))
    else:
        raise ValueError(f"unknown camera preset {camera!r}")
    return b.finish(world, cam, (0.35, 0.5, 0.75), rc)


ATRIUM_HEIGHT = 12.0


def atrium_materials(n_materials=24, texture_size=1024):
    """The atrium's material table: ("image", (H, W, 3) uint8) for the first 8 slots - procedural textures -, ("solid", rgb) for the rest.
    (tests/tools/export_obj.py writes the same table as an MTL file with the textures beside it.)"""
    out = []
    for i in range(n_materials):
        if i < 8:
            out.append(("image", _procedural_texture(i % 3, texture_size, 100 + i)))
        else:
            out.append(("solid", tuple(float(x) for x in 0.25 + 0.6 * counter_uniform(77, 3, i))))
    return out


def atrium_mesh(n_triangles=SPONZA_TRIANGLES, n_materials=24, mesh="regular"):
    """The atrium's triangles: (N, 3, 3) vertices, (N,) material SLOTS (indices into atrium_materials), (N, 3, 2) float32 texture coordinates."""
    L, Wd, Hh = 15.0, 6.0, ATRIUM_HEIGHT  # half length (x), half width (z), height (y)
    parts = []  # (weight, surface, material index, uv tiling)
    tris, uvs, mids = [], [], []
    made = 0
    if mesh == "heterogeneous":
        for f, nu, nv, m, tile in _heterogeneous_atrium_parts(L, Wd, Hh):
            t, uv = _grid(f, nu, nv, tile)
            tris.append(t), uvs.append(uv), mids.append(np.full(len(t), m % n_materials, dtype=np.int32))
            made += len(t)
    elif mesh != "regular":
        raise ValueError(f"unknown mesh preset {mesh!r}")
    else:

        def plane(o, du, dv):
            o, du, dv = np.array(o, float), np.array(du, float), np.array(dv, float)
            return lambda u, v: tuple(o[k] + du[k] * u + dv[k] * v for k in range(3))

        parts.append((10., plane((-L, 0, -Wd), (2 * L, 0, 0), (0, 0, 2 * Wd)), 0, (10, 4)))          # floor
        parts.append((8., plane((-L, 0, -Wd), (2 * L, 0, 0), (0, Hh, 0)), 1, (10, 4)))               # wall z=-W
        parts.append((8., plane((-L, 0, Wd), (2 * L, 0, 0), (0, Hh, 0)), 1, (10, 4)))                # wall z=+W
        parts.append((3., plane((-L, 0, -Wd), (0, 0, 2 * Wd), (0, Hh, 0)), 2, (4, 4)))               # end wall x=-L
        parts.append((3., plane((L, 0, -Wd), (0, 0, 2 * Wd), (0, Hh, 0)), 2, (4, 4)))                # end wall x=+L
        parts.append((4., plane((-L, 5.5, -Wd), (2 * L, 0, 0), (0, 0, 2.0)), 3, (10, 1)))            # gallery floor -z
        parts.append((4., plane((-L, 5.5, Wd - 2.0), (2 * L, 0, 0), (0, 0, 2.0)), 3, (10, 1)))       # gallery floor +z
        parts.append((3., plane((-L, Hh, -Wd), (2 * L, 0, 0), (0, 0, 2.5)), 4, (10, 1)))             # roof strip -z
        parts.append((3., plane((-L, Hh, Wd - 2.5), (2 * L, 0, 0), (0, 0, 2.5)), 4, (10, 1)))        # roof strip +z
        col_x = np.linspace(-L + 1.5, L - 1.5, 10)
        for side in (-1., 1.):
            for storey in range(2):
                y0 = 0.0 if storey == 0 else 5.6
                for i, cx in enumerate(col_x):
                    cz = side * (Wd - 2.0)

                    def column(u, v, cx=cx, cz=cz, y0=y0):
                        r = 0.35 * (1.0 + 0.08 * np.sin(12 * np.pi * u)) * (1.0 - 0.1 * v)
                        return cx + r * np.cos(2 * np.pi * u), y0 + 4.2 * v, cz + r * np.sin(2 * np.pi * u)

                    parts.append((2.2, column, 5 + (i % 3), (2, 3)))
                    if i + 1 < len(col_x):
                        x1 = col_x[i + 1]

                        def arch(u, v, x0=cx, x1=x1, cz=cz, y0=y0):
                            ang = np.pi * u
                            xm, rad = 0.5 * (x0 + x1), 0.5 * (x1 - x0)
                            return xm - rad * np.cos(ang), y0 + 4.2 + 0.9 * np.sin(ang), cz - 0.3 + 0.6 * v

                        parts.append((1.2, arch, 8 + (i % 4), (3, 1)))
        for i in range(6):  # drapes hanging from the galleries
            x0 = -L + 3.0 + i * 4.6
            side = -1. if i % 2 == 0 else 1.

            def drape(u, v, x0=x0, side=side, i=i):
                zc = side * (Wd - 2.3)
                return (x0 + 2.2 * u, 5.4 - 3.0 * v + 0.1 * np.sin(6 * np.pi * u),
                        zc + 0.35 * np.sin(10 * np.pi * u + i) * (0.3 + v))

            parts.append((6., drape, 12 + i, (2, 2)))
        total_w = sum(p[0] for p in parts)
        budget = n_triangles - 1  # one pennant triangle closes odd totals
        for k, (w, f, m, tile) in enumerate(parts):
            share = budget * w / total_w if k + 1 < len(parts) else budget - made
            cells = max(1, int(share // 2))
            nu = max(1, int(round(math.sqrt(cells * 2.0))))
            nv = max(1, cells // nu)
            t, uv = _grid(f, nu, nv, tile)
            tris.append(t), uvs.append(uv), mids.append(np.full(len(t), m % n_materials, dtype=np.int32))
            made += len(t)
    # filler: a strip of small pennants along the -z gallery rail until the exact count is reached
    missing = n_triangles - made
    if missing < 0:
        raise ValueError("triangle budget too small for the atrium layout")
    if missing:
        i = np.arange(missing, dtype=np.float64)
        x = -L + 0.5 + (2 * L - 1.0) * (i + 0.5) / missing
        w_ = min(0.2, (2 * L - 1.0) / missing * 0.45)
        p0 = np.stack([x - w_, np.full_like(x, 6.6), np.full_like(x, -Wd + 2.0)], 1)
        p1 = np.stack([x + w_, np.full_like(x, 6.6), np.full_like(x, -Wd + 2.0)], 1)
        p2 = np.stack([x, np.full_like(x, 6.2), np.full_like(x, -Wd + 2.05)], 1)
        tris.append(np.stack([p0, p1, p2], 1))
        uvs.append(np.tile(np.array([[0, 0], [1, 0], [.5, 1]], np.float32), (missing, 1, 1)))
        mids.append(np.full(missing, n_materials - 1, dtype=np.int32))
    tri = np.concatenate(tris)
    assert len(tri) == n_triangles, (len(tri), n_triangles)
    return tri, np.concatenate(mids), np.concatenate(uvs)


def obj_bounds(path):
    """Axis-aligned bounds of the `v` lines of an OBJ file (for placing a default camera and light)."""
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    with open(path, "r", errors="replace") as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                if len(p) >= 4:
                    v = np.array([float(p[1]), float(p[2]), float(p[3])])
                    lo, hi = np.minimum(lo, v), np.maximum(hi, v)
    if not np.isfinite(lo).all():
        raise ValueError(f"{path}: no vertices")
    return lo, hi


def obj_file_scene(path, render_config=None, camera=None, light=None, background=(0.35, 0.5, 0.75)):
    """A user-supplied OBJ (+MTL, textures beside it) in place of a stand-in, e.g. the real Crytek sponza.obj for C3
    ("if a real sponza.obj is supplied at run time, use it and say so"). Loaded by the C++ host's OBJ loader
    with the reference's material mapping (src/loader/obj.rs:38-136). camera = (look_from, look_at, vfov) and
    light = (q, u, v, rgb) override the defaults: an eye at 15 % of the long horizontal axis, 30 % up, looking down that axis,
    and a Quad light over the middle third of the model's top (y up), as in the atrium stand-in."""
    rc = render_config or RenderConfig(width=1920, height=1080, samples_per_pixel=512)
    d, fn = os.path.split(os.path.abspath(path))
    if light is None or camera is None:  # (a scan of the file's vertices in Python: seconds on a million-face file, so only when a default is asked for)
        lo, hi = obj_bounds(path)
        ext = hi - lo
        ax = 0 if ext[0] >= ext[2] else 2  # long horizontal axis
        ox = 2 - ax
    b = SceneBuilder()
    model = b.load_obj(d + os.sep, fn, None, b.Lambertian(b.SolidColor(0.7, 0.7, 0.7)))
    if light is None:
        q = lo.copy()
        q[1] = hi[1] + 0.02 * ext[1]
        q[ax] = lo[ax] + ext[ax] / 3.0
        q[ox] = lo[ox] + ext[ox] / 3.0
        u, v = np.zeros(3), np.zeros(3)
        u[ax], v[ox] = ext[ax] / 3.0, ext[ox] / 3.0
        light = (q, u, v, (18., 17., 15.))
    lq = b.Quad(tuple(light[0]), tuple(light[1]), tuple(light[2]), b.DiffuseLight(*light[3]))
    if camera is None:
        eye, at = lo + 0.5 * ext, lo + 0.5 * ext
        eye[ax] = lo[ax] + 0.15 * ext[ax]
        eye[1] = lo[1] + 0.3 * ext[1]
        at[ax] = lo[ax] + 0.8 * ext[ax]
        at[1] = lo[1] + 0.4 * ext[1]
        camera = (eye, at, 55.)
    cam = CameraConfig(vertical_fov_degrees=float(camera[2]), aperture_size=0., look_from=tuple(camera[0]), look_at=tuple(camera[1]),
                       up=(0, 1, 0))
    return b.finish(b.Bvh([model, lq]), cam, background, rc)


def procedural_sky(width=2048, height=1024, sun_dir=(0.45, 0.7, -0.55), sun_radiance=60.0):
    """A latitude-longitude HDR environment (H, W, 3) float32, row 0 = up, in the direction convention of SolSceneDesc::env_*
    (the reference's sphere mapping, src/hittable/sphere.rs:134-140): horizon-to-zenith gradient, a darker ground half and a
    small bright sun. Stand-in for the "HDRI env light" BASELINE.json's config 5 names (no HDR files offline)."""
    v = (np.arange(height, dtype=np.float64) + 0.5) / height          # 0 = up .. 1 = down  (row = (1 - v_ref) * (h - 1))
    u = (np.arange(width, dtype=np.float64) + 0.5) / width
    theta = (1.0 - v) * np.pi                                          # v_ref = theta / pi, theta = acos(-y)
    y = -np.cos(theta)[:, None]
    phi = u * 2.0 * np.pi                                              # u = phi / 2pi, phi = -atan2(z, x) + pi
    r = np.sqrt(np.maximum(0.0, 1.0 - y * y))
    x = r * np.cos(np.pi - phi)[None, :]
    z = r * np.sin(np.pi - phi)[None, :]
    up = np.clip(y, 0.0, 1.0)
    sky = np.stack([0.45 - 0.30 * up, 0.60 - 0.25 * up, 0.85 - 0.10 * up], -1) * np.ones((1, width, 1))
    ground = np.array([0.10, 0.09, 0.08])
    img = np.where((y > 0.0)[..., None] * np.ones((1, width, 1), bool), sky, ground)
    sd = np.asarray(sun_dir, np.float64)
    sd = sd / np.linalg.norm(sd)
    cosang = x * sd[0] + y * sd[1] + z * sd[2]
    img = img + (cosang > np.cos(np.radians(2.5)))[..., None] * sun_radiance
    return img.astype(np.float32)


STATUE_TRIANGLES = 1_090_000


def statue_like(render_config=None, n_triangles=STATUE_TRIANGLES, environment=False, camera="default"):
    """C5 stand-in: a procedurally displaced "statue" of about 1.09 M triangles - a noise-displaced body of
    Metal(fuzz 0.1), a Dielectric(1.5) head and a glass orb, a Lambertian plinth and drapery - on a floor quad under one
    quad light, constant background. (The Happy Buddha mesh and an HDRI light are not available / not in the reference.)"""
    rc = render_config or RenderConfig(width=1920, height=1080, samples_per_pixel=2048)
    b = SceneBuilder()
    metal = b.Metal(b.SolidColor(.85, .7, .45), None, 0.1)
    glass = b.Dielectric(b.SolidColor(1., 1., 1.), None, 1.5)
    stone = b.Lambertian(b.SolidColor(.55, .52, .5))
    cloth = b.Lambertian(b.SolidColor(.6, .15, .12))
    tri, slots, uv = statue_mesh(n_triangles)
    first, n = b.triangles(tri, np.asarray([metal, glass, stone, cloth], dtype=np.int32)[slots], uv)
    model = b.Bvh_range(first, n)
    orb = b.Sphere((2.2, 0.7, 1.2), 0.7, glass)
    floor = b.Quad((-12., 0., -12.), (24., 0, 0), (0, 0, 24.), b.Lambertian(b.SolidColor(.4, .42, .45)))
    light = b.Quad((-2.5, 8.5, -1.0), (5., 0, 0), (0, 0, 4.), b.DiffuseLight(20., 19., 17.))
    if camera == "closeup":  # STRESS variant: the statue fills the frame, seen from above so that the glass head lies in front of the metal body
        cam = CameraConfig(vertical_fov_degrees=40., aperture_size=0., look_from=(1.6, 7.2, 3.4), look_at=(0.0, 3.1, 0.0), up=(0, 1, 0))
    elif camera == "default":
        cam = CameraConfig(vertical_fov_degrees=38., aperture_size=0., look_from=(4.5, 3.6, 8.0), look_at=(0.2, 2.6, 0.0), up=(0, 1, 0))
    elif camera == "far":  # the default view from ten times the distance through a lens ten times as long (the regime in which fp32 loses a distant origin's digits)
        cam = CameraConfig(vertical_fov_degrees=3.94, aperture_size=0., look_from=(43.2, 12.6, 80.0), look_at=(0.2, 2.6, 0.0), up=(0, 1, 0))
    else:
        raise ValueError(f"unknown camera preset {camera!r}")
    if environment:  # EXTENSION (not in the reference): the "HDRI env light" of BASELINE.json's config 5, as a procedural HDR sky
        b.environment(procedural_sky(), 1.0)
    return b.finish(b.Bvh([model, orb, floor, light]), cam, (0.25, 0.3, 0.4), rc)


STATUE_MATERIALS = (("metal body", (.85, .7, .45)), ("glass head", (1., 1., 1.)), ("stone plinth", (.55, .52, .5)), ("cloth drape", (.6, .15, .12)))


def statue_mesh(n_triangles=STATUE_TRIANGLES):
    """The statue's triangles: (N, 3, 3) vertices, (N,) material slots (0 metal body, 1 glass head, 2 stone plinth, 3 cloth drape), (N, 3, 2) texture coordinates."""
    metal, glass, stone, cloth = 0, 1, 2, 3

    def noise(u, v, k, amp):  # smooth, periodic in u
        return amp * (np.sin(2 * np.pi * (k * u) + 7.0 * v) * np.cos(2 * np.pi * (0.5 * k * v) + 3.0 * u) +
                      0.5 * np.sin(2 * np.pi * (3 * k * u) + 1.3) * np.sin(2 * np.pi * (2 * k * v)))

    def body(u, v):  # torso: a lathe profile with folds
        y = 1.0 + 3.2 * v
        r = (0.95 - 0.35 * v + 0.25 * np.sin(np.pi * v) ** 2) * (1.0 + noise(u, v, 6, 0.06)) + 0.05 * np.sin(40 * np.pi * v)
        return r * np.cos(2 * np.pi * u), y, r * np.sin(2 * np.pi * u)

    def head(u, v):
        th = np.pi * (0.02 + 0.96 * v)
        r = 0.55 * (1.0 + noise(u, v, 4, 0.05))
        return r * np.sin(th) * np.cos(2 * np.pi * u), 4.75 - r * np.cos(th), r * np.sin(th) * np.sin(2 * np.pi * u)

    def plinth(u, v):
        y = 1.0 * v
        r = 1.5 * (1.0 - 0.15 * v) * (1.0 + 0.03 * np.sign(np.sin(16 * np.pi * u)))
        return r * np.cos(2 * np.pi * u), y, r * np.sin(2 * np.pi * u)

    def drape(u, v):
        ang = 2 * np.pi * (0.15 + 0.5 * u)
        r = 1.25 + 0.12 * np.sin(18 * np.pi * u) * (0.3 + v)
        return r * np.cos(ang), 3.4 - 2.3 * v + 0.05 * np.sin(9 * np.pi * u), r * np.sin(ang)

    parts = [(0.60, body, metal), (0.16, head, glass), (0.12, plinth, stone), (0.12, drape, cloth)]
    tris, uvs, mids = [], [], []
    made = 0
    for k, (w, f, m) in enumerate(parts):
        share = n_triangles * w if k + 1 < len(parts) else n_triangles - made
        cells = max(1, int(share // 2))
        nu = max(1, int(round(math.sqrt(cells * 2.0))))
        nv = max(1, cells // nu)
        t, uv = _grid(f, nu, nv)
        tris.append(t), uvs.append(uv), mids.append(np.full(len(t), m, dtype=np.int32))
        made += len(t)
    return np.concatenate(tris), np.concatenate(mids), np.concatenate(uvs)


# ---------------------------------------------------------------------------------------------------------------
# The reference's test scenes (tests/scenes.rs)
# ---------------------------------------------------------------------------------------------------------------
def create_test_scene(render_config, environment=None):
    """tests/scenes.rs:17-122 (environment = (map, scale): the extension of create_test_scene_with_environment)"""
    b = SceneBuilder()
    if environment is not None:
        b.environment(*environment)
    cam = CameraConfig(20., 0.1, (-5., 3., 6.), (.25, 1., 0.), (0., 1., 0.))
    ground = b.Lambertian(b.ImageMap(load_image("textures/tex.jpg")))
    glass = b.Dielectric(b.SolidColor(1., 1., 1.), None, 1.5)
    light = b.DiffuseLight(10., 10., 10.)
    red = b.Lambertian(b.SolidColor(1., 0., 0.))
    world = [b.Quad((-5., 0., -15.), (20., 0., 0.), (0., 0., 20.), ground),
             b.Sphere((-1., 1., 0.), 1., glass)]
    world += b.new_box((0., 0., -.5), (1., 2., .5), red, RotationY(15.))
    world.append(b.ConstantMedium(b.Bvh(b.new_box((0., 0., -.5), (1., 2., .5), red, Translation((0., 0., 1.)))), 0.1,
                                  (1., 1., 1.)))
    world += b.new_box((-1., 2., 0.), (-.5, 2.5, .5), red)
    balls = []
    for ii in range(0, 10, 2):
        i = ii * 0.1
        for jj in range(0, 10, 2):
            j = jj * 0.1
            for kk in range(0, 10, 2):
                k = kk * 0.1
                balls.append(b.Triangle((i, j + .05, k + .8), (i, j, k + .8), (i, j + .05, k), red))
    world.append(b.Bvh(balls))
    world.append(b.Triangle((1., .1, 2.), (3., .1, 2.), (2., .1, 1.), red))
    world.append(b.Sphere((10., 5., 10.), 10., light))
    world.append(b.Quad((0., 0., 0.), (2., 0., 0.), (0., 0., 2.), light, [RotationY(45.), Translation((-1., 10., -1.))]))
    world.append(b.Triangle((-2., 1., -3.), (0., 1., -3.), (-1., 2., -3.), light))
    return b.finish(b.Bvh(world), cam, (.2, .3, .5), render_config)


def create_test_scene_with_environment(render_config, size=(256, 128)):
    """The reference's test scene under an environment map (EXTENSION, SolSceneDesc::env_*) instead of its constant background."""
    return create_test_scene(render_config, environment=(procedural_sky(size[0], size[1]), 0.8))


def create_simple_test_scene(render_config, add_light):
    """tests/scenes.rs:170-193"""
    b = SceneBuilder()
    cam = CameraConfig(20., 0.1, (0., 0., 4.), (0., 0., 0.), (0., 1., 0.))
    yellow = b.Lambertian(b.SolidColor(1., 1., 0.))
    light = b.DiffuseLight(10., 10., 10.)
    world = []
    if add_light:
        world.append(b.Sphere((0., 100., 0.), 20., light))
    world.append(b.Sphere((0., 0., 0.), .5, yellow))
    return b.finish(b.Bvh(world), cam, (.2, .3, .5), render_config)


def create_uv_scene(render_config):
    """tests/scenes.rs:196-230"""
    b = SceneBuilder()
    cam = CameraConfig(20., 0., (0., 1., 5.), (0., 1., 0.), (0., 1., 0.))
    light = b.DiffuseLight(10., 10., 10.)
    checker = b.Lambertian(b.ImageMap(load_image("textures/checker.jpg")))
    world = [b.Sphere((50., 50., 50.), 20., light),
             b.Triangle((-1., 0., 0.), (1., 0., 0.), (0., 2., 0.), checker, None, uv=((-1., -1.), (2., -1.), (0., 2.)))]
    return b.finish(b.Bvh(world), cam, (.2, .3, .5), render_config)


def _resource_dir(name):
    """The reference's `resources/<name>/` prefix (its loaders concatenate path + filename, src/loader/obj.rs:50)."""
    return os.path.join(RESOURCES, name) + os.sep


def create_obj_scene(render_config):
    """tests/scenes.rs:318-352: the spider model (resources/spider/spider.obj, 5 textured materials) over a textured ground."""
    b = SceneBuilder()
    cam = CameraConfig(30., 20., (-250., 30., 150.), (-50., 0., 0.), (0., 1., 0.))
    light = b.DiffuseLight(15., 15., 15.)
    world = [b.Sphere((-100., 100., 40.), 35., light), b.load_obj(_resource_dir("spider"), "spider.obj")]
    ground = b.Lambertian(b.ImageMap(load_image("textures/tex.jpg")))
    world.append(b.Quad((-200., -30., -200.), (400., 0., 0.), (0., 0., 400.), ground))
    return b.finish(b.Bvh(world), cam, (.2, .3, .5), render_config)


def create_obj_with_box(render_config, filename, path=None):
    """tests/scenes.rs:355-381 (default material: red Lambertian)."""
    b = SceneBuilder()
    cam = CameraConfig(30., 0., (2., 1., 3.), (0., 0., 0.), (0., 1., 0.))
    light = b.DiffuseLight(15., 15., 15.)
    red = b.Lambertian(b.SolidColor(1., 0., 0.))
    world = [b.Sphere((-100., 100., 40.), 35., light), b.load_obj(path or _resource_dir("obj"), filename, None, red)]
    return b.finish(b.Bvh(world), cam, (.2, .3, .5), render_config)


def create_obj_with_triangle(render_config, filename, path=None):
    """tests/scenes.rs:384-409"""
    b = SceneBuilder()
    cam = CameraConfig(30., 0., (0., 0., 2.), (0., 0., 0.), (0., 1., 0.))
    light = b.DiffuseLight(15., 15., 15.)
    world = [b.Sphere((100., 0., 100.), 35., light), b.load_obj(path or _resource_dir("obj"), filename)]
    return b.finish(b.Bvh(world), cam, (0., 0., 0.), render_config)


def create_normal_mapping_scene(render_config, light_pos, normal_mapping_enabled):
    """tests/scenes.rs:233-280"""
    b = SceneBuilder()
    cam = CameraConfig(40., 0., (.2, .2, 2.), (0., 0., 0.), (0., 1., 0.))
    light = b.DiffuseLight(45., 45., 45.)
    world = [b.Sphere(light_pos, 5., light)]
    ntex = b.load_normal_texture(load_image("textures/normal.png")) if normal_mapping_enabled else None
    mat = b.Lambertian(b.SolidColor(.8, .8, .8), ntex)
    red = b.Lambertian(b.SolidColor(1., 0., 0.))
    world += b.new_box((-.1, -.1, 0.), (.1, .1, 1.), red)
    world.append(b.Quad((-1., -1., 0.), (2., 0., 0.), (0., 2., 0.), mat))
    return b.finish(b.Bvh(world), cam, (0., 0., 0.), render_config)


def create_normal_mapping_sphere_scene(render_config, light_pos):
    """tests/scenes.rs:283-315"""
    b = SceneBuilder()
    cam = CameraConfig(40., 0., (.2, .2, 2.), (0., 0., 0.), (0., 1., 0.))
    light = b.DiffuseLight(45., 45., 45.)
    ntex = b.load_normal_texture(load_image("textures/earth_height.jpg"))
    mat = b.Lambertian(b.SolidColor(.8, .8, .8), ntex)
    world = [b.Sphere(light_pos, 5., light), b.Sphere((0., 0., 0.), .6, mat)]
    return b.finish(b.Bvh(world), cam, (0., 0., 0.), render_config)


def create_light_attenuation_scene(render_config, attenuation_half_length):
    """tests/scenes.rs:412-449"""
    b = SceneBuilder()
    cam = CameraConfig(20., 0., (0., 1., 2.), (0., .2, 0.), (0., 1., 0.))
    light = b.DiffuseLight(25., 25., 25., attenuation_half_length)
    red = b.Lambertian(b.SolidColor(1., 0., 0.))
    green = b.Lambertian(b.SolidColor(0., 1., 0.))
    blue = b.Lambertian(b.SolidColor(0., 0., 1.))
    glass = b.Dielectric(b.SolidColor(.8, .8, .8), None, 1.5)
    world = [b.Sphere((0., .2, 0.), .03, light), b.Sphere((.25, .1, .25), .1, green), b.Sphere((.25, .1, -.5), .1, blue),
             b.Sphere((-.1, .1, -.1), .1, glass), b.Quad((-1., 0., -1.), (2., 0., 0.), (0., 0., 2.), red)]
    return b.finish(b.Bvh(world), cam, (0., 0., 0.), render_config)


def create_quad_rotation_scene(render_config, rotation):
    """tests/scenes.rs:452-479"""
    b = SceneBuilder()
    world = [b.Quad((-100., 0., -100.), (200., 0., 0.), (0., 0., 200.), b.Lambertian(b.SolidColor(0., 1., 0.)), rotation),
             b.Sphere((100., 300., -500.), 50., b.DiffuseLight(15., 15., 15.))]
    cam = CameraConfig(vertical_fov_degrees=35., look_from=(0., 200., -500.))
    return b.finish(b.Bvh(world), cam, (0., 0., 0.), render_config)


def create_blend_material_scene(render_config, blend_factor):
    """tests/scenes.rs:482-513"""
    b = SceneBuilder()
    m = b.Blend(b.Lambertian(b.ImageMap(load_image("textures/checker.jpg"))), b.Lambertian(b.SolidColor(0., 1., 0.)),
                blend_factor)
    world = [b.Quad((-100., 0., -100.), (200., 0., 0.), (0., 0., 200.), m),
             b.Sphere((0., 500., -200.), 50., b.DiffuseLight(15., 15., 15.))]
    cam = CameraConfig(vertical_fov_degrees=35., look_from=(0., 400., -100.))
    return b.finish(b.Bvh(world), cam, (0., 0., 0.), render_config)


def new_bvh_test_scene(render_config, use_bvh, num_triangles):
    """tests/scenes.rs:125-167 (bench-only scene in the reference)"""
    b = SceneBuilder()
    cam = CameraConfig(20., 0.1, (-.5, 0., 4.), (-.5, 0., 0.), (0., 1., 0.))
    yellow = b.Lambertian(b.SolidColor(1., 1., 0.))
    world = [b.Sphere((0., 4., 10.), 4., b.DiffuseLight(10., 10., 10.))]
    tris = []
    for x in range(num_triangles):
        cx = x - num_triangles / 2.
        tris.append(b.Triangle((cx, -.5, 0.), (cx + 1., -.5, 0.), (cx + .5, .5, 0.), yellow))
    if use_bvh:
        world.append(b.Bvh(tris))
    else:
        world += tris
    return b.finish(b.Bvh(world), cam, (.2, .3, .5), render_config)
