"""Writes the stand-in meshes as Wavefront OBJ + MTL (+ texture images), at the sizes BASELINE.json names, so that the OBJ ingest path
(host/solstrale_obj.cpp = src/loader/obj.rs:38-136 restated) runs on files of the size configs 3 - 5 name and not only on the reference's
1 368-face spider. TEST INFRASTRUCTURE (tests/test_obj_scale.py, bench.py --obj <file>); not a pytest.
  atrium  the heterogeneous atrium: 262 267 triangles, 24 materials - 8 with `map_Kd` image textures, one of those also with a `map_bump`
          height map (the loader's normal-vs-height detection, texture.rs:53-97) -, 16 with `Kd` colours
  statue  the 1.09 M-triangle statue (4 `Kd` materials: the reference's loader makes every OBJ material Lambertian, obj.rs:57-76)
Vertices are written per triangle (no sharing), `v` / `vt` with 9 significant digits (tobj parses them as f32), faces `f a/a b/b c/c`,
one `usemtl` group per material in first-use order (each is a tobj model; triangle order = file order = what Bvh::new sees).
Usage: python export_obj.py atrium|statue OUT_DIR [n_triangles] [texture_size]   -> OUT_DIR/<name>.obj, <name>.mtl, <name>_tex<k>.png"""
import _paths  # noqa: F401
import os
import sys
import time

import numpy as np
from PIL import Image

from solstrale_amd import scenes


def _write_mesh(path, mtl_name, tri, slots, uv, material_names):
    order = np.argsort(slots, kind="stable")
    tri, slots, uv = tri[order], slots[order], uv[order]
    n = len(tri)
    with open(path, "w") as f:
        f.write(f"# {n} triangles\nmtllib {mtl_name}\n")
        v = tri.reshape(-1, 3)
        f.write("".join("v %.9g %.9g %.9g\n" % (a, b, c) for a, b, c in v.astype(np.float32).tolist()))
        t = uv.reshape(-1, 2)
        f.write("".join("vt %.9g %.9g\n" % (a, b) for a, b in t.astype(np.float32).tolist()))
        start = 0
        while start < n:
            end = start + int(np.searchsorted(slots[start:], slots[start], side="right"))
            f.write(f"g part_{int(slots[start])}\nusemtl {material_names[int(slots[start])]}\n")
            idx = np.arange(start * 3 + 1, end * 3 + 1).reshape(-1, 3)
            f.write("".join("f %d/%d %d/%d %d/%d\n" % (a, a, b, b, c, c) for a, b, c in idx.tolist()))
            start = end


def export_atrium(out_dir, n_triangles=scenes.SPONZA_TRIANGLES, texture_size=1024, mesh="heterogeneous", name="atrium"):
    os.makedirs(out_dir, exist_ok=True)
    mats = scenes.atrium_materials(24, texture_size)
    names = [f"m{k:02d}" for k in range(len(mats))]
    with open(os.path.join(out_dir, name + ".mtl"), "w") as f:
        for k, (kind, value) in enumerate(mats):
            f.write(f"newmtl {names[k]}\n")
            if kind == "image":
                fn = f"{name}_tex{k}.png"
                Image.fromarray(value).save(os.path.join(out_dir, fn))
                f.write(f"Kd 1 1 1\nmap_Kd {fn}\n")
                if k == 3:  # one bump map: a grey HEIGHT map (r = g = b), which the loader turns into normals (height_map.rs:68-86)
                    yy, xx = np.mgrid[0:256, 0:256]
                    h = (127.5 + 127.5 * np.sin(xx * 2 * np.pi / 32.0) * np.cos(yy * 2 * np.pi / 48.0)).astype(np.uint8)
                    Image.fromarray(np.stack([h, h, h], -1)).save(os.path.join(out_dir, f"{name}_bump{k}.png"))
                    f.write(f"map_bump -bm 1 {name}_bump{k}.png\n")
            else:
                f.write("Kd %.9g %.9g %.9g\n" % tuple(np.float32(value).tolist()))
    tri, slots, uv = scenes.atrium_mesh(n_triangles, 24, mesh)
    _write_mesh(os.path.join(out_dir, name + ".obj"), name + ".mtl", tri, slots, uv, names)
    return os.path.join(out_dir, name + ".obj")


def export_statue(out_dir, n_triangles=scenes.STATUE_TRIANGLES, name="statue"):
    os.makedirs(out_dir, exist_ok=True)
    names = [f"m{k}" for k in range(len(scenes.STATUE_MATERIALS))]
    with open(os.path.join(out_dir, name + ".mtl"), "w") as f:
        for k, (what, rgb) in enumerate(scenes.STATUE_MATERIALS):
            f.write(f"# {what}\nnewmtl {names[k]}\nKd %.9g %.9g %.9g\n" % rgb)
    tri, slots, uv = scenes.statue_mesh(n_triangles)
    _write_mesh(os.path.join(out_dir, name + ".obj"), name + ".mtl", tri, slots, uv, names)
    return os.path.join(out_dir, name + ".obj")


if __name__ == "__main__":
    which, out_dir = sys.argv[1], sys.argv[2]
    t0 = time.time()
    if which == "atrium":
        p = export_atrium(out_dir, int(sys.argv[3]) if len(sys.argv) > 3 else scenes.SPONZA_TRIANGLES, int(sys.argv[4]) if len(sys.argv) > 4 else 1024)
    else:
        p = export_statue(out_dir, int(sys.argv[3]) if len(sys.argv) > 3 else scenes.STATUE_TRIANGLES)
    print(f"{p}: {os.path.getsize(p) / 1e6:.1f} MB in {time.time() - t0:.1f} s")
