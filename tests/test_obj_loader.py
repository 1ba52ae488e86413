"""The OBJ + MTL loader of the C++ host (solstrale-rust_amd/host/solstrale_obj.cpp) against the reference's loader tests
(src/loader/obj.rs:148-185) and an independent pure-Python reading of the same files. The golden-image cases obj, obj_default,
obj_diffuse, obj_normal_map, obj_height_map run with the other reference cases (tests/ref_cases.py)."""
import os
import shutil

import numpy as np
import pytest

from solstrale_amd import CameraConfig, HostError, RenderConfig, SceneBuilder, scenes

OBJ_DIR = scenes._resource_dir("obj")
SPIDER_DIR = scenes._resource_dir("spider")


def _scene_of(hittable_builder):
    """Flattens a world holding a light and the loaded model; returns the description."""
    b = SceneBuilder()
    light = b.Sphere((0., 50., 0.), 1., b.DiffuseLight(1., 1., 1.))
    model = hittable_builder(b)
    return b.finish(b.Bvh([light, model]), CameraConfig(30., 0., (0., 0., 5.), (0., 0., 0.), (0., 1., 0.)), (0., 0., 0.),
                    RenderConfig(8, 8, 1))


def _triangles(sc):
    d = sc.desc
    out = []
    for i in range(d.n_triangles):
        t = d.triangles[i]
        v0 = np.array(t.v0[:])
        out.append((v0, v0 + np.array(t.v0v1[:]), v0 + np.array(t.v0v2[:])))
    return out


def _py_obj_triangles(path):
    """Independent reading: f32 vertices, fan triangulation, file order."""
    pos, tris = [], []
    for line in open(path):
        t = line.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] == "v":
            pos.append([np.float32(x) for x in t[1:4]])
        elif t[0] == "f":
            idx = []
            for tok in t[1:]:
                v = int(tok.split("/")[0])
                idx.append(v - 1 if v > 0 else len(pos) + v)
            for k in range(1, len(idx) - 1):
                tris.append((idx[0], idx[k], idx[k + 1]))
    p = np.array(pos, dtype=np.float32).astype(np.float64)
    return [(p[a], p[b], p[c]) for a, b, c in tris]


def _same_triangle_set(got, want):
    key = lambda tri: tuple(np.round(np.concatenate(tri), 9))
    return sorted(map(key, got)) == sorted(map(key, want))


def test_missing_file():
    with pytest.raises(HostError) as e:
        SceneBuilder().load_obj("resources/obj/", "missing.obj")
    assert str(e.value) == "failed to load obj model from resources/obj/missing.obj"  # obj.rs:152-158


def test_missing_material_file():
    with pytest.raises(HostError) as e:
        SceneBuilder().load_obj(OBJ_DIR, "missingMaterialLib.obj")
    assert str(e.value) == f"failed to load MTL file for {OBJ_DIR}missingMaterialLib.obj"  # obj.rs:160-168


def test_missing_image_file():
    with pytest.raises(HostError) as e:
        SceneBuilder().load_obj(OBJ_DIR, "missingImage.obj")
    assert f"Failed to open image texture {OBJ_DIR}missing.jpg" in str(e.value)  # obj.rs:170-175


def test_invalid_image_file():
    with pytest.raises(HostError) as e:
        SceneBuilder().load_obj(OBJ_DIR, "invalidImage.obj")
    assert f"Failed to decode image texture {OBJ_DIR}invalidImage.mtl" in str(e.value)  # obj.rs:177-183


def test_box_quads_are_fanned_and_take_the_default_material():
    sc = _scene_of(lambda b: b.load_obj(OBJ_DIR, "box.obj", None, b.Lambertian(b.SolidColor(1., 0., 0.))))
    d = sc.desc
    assert d.n_triangles == 12
    assert _same_triangle_set(_triangles(sc), _py_obj_triangles(OBJ_DIR + "box.obj"))
    mats = {d.triangles[i].material for i in range(12)}
    assert len(mats) == 1  # `usemtl Default` names no loaded material -> material id None -> the default material
    m = d.materials[mats.pop()]
    assert tuple(d.textures[m.albedo_tex].rgb) == (1., 0., 0.)


def test_kd_becomes_a_solid_lambertian():
    sc = _scene_of(lambda b: b.load_obj(OBJ_DIR, "boxWithMat.obj", None, b.Lambertian(b.SolidColor(1., 0., 0.))))
    d = sc.desc
    mats = {d.triangles[i].material for i in range(d.n_triangles)}
    assert d.n_triangles == 12 and len(mats) == 1
    m = d.materials[mats.pop()]
    assert tuple(d.textures[m.albedo_tex].rgb) == (0., 0., 1.) and m.normal_tex < 0  # Kd 0 0 1, no bump map


def test_triangle_with_uvs_and_bump_maps():
    for name, is_height in (("triWithNormalMap.obj", False), ("triWithHeightMap.obj", True)):
        sc = _scene_of(lambda b: b.load_obj(OBJ_DIR, name))
        d = sc.desc
        assert d.n_triangles == 1
        t = d.triangles[0]
        assert tuple(t.uv0) == (0., 0.) and tuple(t.uv1) == (1., 0.) and tuple(t.uv2) == (.5, 1.)
        m = d.materials[t.material]
        assert m.normal_tex >= 0 and d.textures[m.normal_tex].kind != 0  # an image texture
        assert tuple(d.textures[m.albedo_tex].rgb) == ((1., 1., 0.) if is_height else (0., 0., 1.))


def test_spider_counts_materials_and_file_order():
    sc = _scene_of(lambda b: b.load_obj(SPIDER_DIR, "spider.obj"))
    d = sc.desc
    want = _py_obj_triangles(SPIDER_DIR + "spider.obj")
    assert d.n_triangles == len(want) == 1368
    assert _same_triangle_set(_triangles(sc), want)
    used = {d.triangles[i].material for i in range(d.n_triangles)}
    assert len(used) == 4  # usemtl Skin, HLeibTex, BeinTex, Augentex (Brusttex is defined but unused): image-textured Lambertians
    assert all(d.textures[d.materials[m].albedo_tex].kind != 0 for m in used)


def test_negative_indices_polygons_and_transformation(tmp_path):
    p = tmp_path / "m"
    p.mkdir()
    (p / "pent.obj").write_text("v 0 0 0\nv 1 0 0\nv 1.5 1 0\nv 0.5 2 0\nv -0.5 1 0\n\nf -5 -4 -3 -2 -1\nl 1 2\np 1\n")
    sc = _scene_of(lambda b: b.load_obj(str(p) + os.sep, "pent.obj", (0, (1., 2., 3.))))  # Translation(1, 2, 3)
    got = _triangles(sc)
    want = [tuple(v + np.array([1., 2., 3.]) for v in tri) for tri in _py_obj_triangles(str(p / "pent.obj"))]
    assert len(got) == 3 and _same_triangle_set(got, want)


def test_relative_texture_path_like_the_reference(tmp_path):
    # the reference concatenates path + name (obj.rs:66,72): "../textures/.." in triWithHeightMap.mtl resolves against path
    root = tmp_path / "resources"
    shutil.copytree(os.path.dirname(OBJ_DIR.rstrip(os.sep)), root)
    sc = _scene_of(lambda b: b.load_obj(str(root / "obj") + os.sep, "triWithHeightMap.obj"))
    assert sc.desc.n_triangles == 1


MALFORMED = {
    "empty": "",
    "only_comments": "# nothing\n\n#\n",
    "face_before_vertices": "f 1 2 3\nv 0 0 0\nv 1 0 0\nv 0 1 0\n",
    "index_zero": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n",
    "index_out_of_range": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7\n",
    "negative_out_of_range": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf -1 -2 -9\n",
    "huge_index": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 99999999999999999999\n",
    "two_vertex_face": "v 0 0 0\nv 1 0 0\nf 1 2\n",
    "non_numeric": "v a b c\nv 1 0 0\nv 0 1 0\nf 1 2 3\n",
    "nan_and_inf": "v nan 0 0\nv inf 0 0\nv 0 -inf 0\nf 1 2 3\n",
    "short_vertex": "v 1\nv 1 2\nv\nf 1 2 3\n",
    "slashes": "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nf 1/ 2// 3/9/9\nf /1 //2 ///\n",
    "texcoord_out_of_range": "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nf 1/5 2/-7 3/0\n",
    "unknown_material": "v 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl nope\nf 1 2 3\n",
    "mtllib_garbage": "mtllib garbage.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\n",
    "long_line": "v " + "1 " * 20000 + "\nv 1 0 0\nv 0 1 0\nf 1 2 3\n",
    "binary_noise": "".join(chr((i * 37 + 11) % 256) for i in range(4096)),
    "no_trailing_newline": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3",
    "crlf": "v 0 0 0\r\nv 1 0 0\r\nv 0 1 0\r\nf 1 2 3\r\n",
}


@pytest.mark.parametrize("name", list(MALFORMED))
def test_malformed_obj_and_mtl_are_errors_or_models_never_crashes(name, tmp_path):
    """Untrusted text in, an error string or a model out (tobj 4.0.2 behind src/loader/obj.rs:51 returns Err for what it cannot read;
    the reference maps that to an error): no crash, no out-of-range triangle, whatever the file holds. Also run under
    -fsanitize=address,undefined by tests/tools/sanitize.sh."""
    p = tmp_path / "m"
    p.mkdir()
    (p / "x.obj").write_bytes(MALFORMED[name].encode("latin-1"))
    (p / "garbage.mtl").write_bytes(b"newmtl a\nKd 1 x\nmap_Kd\nKd\nnewmtl\nmap_bump missing.png\nNs nan\n\xff\xfe\x00newmtl b\nKd 0.1 0.2 0.3\n")
    try:
        sc = _scene_of(lambda b: b.load_obj(str(p) + os.sep, "x.obj"))
    except HostError as e:
        assert str(e)  # an error message, like the reference's Result::Err
        return
    d = sc.desc
    for i in range(d.n_triangles):
        t = d.triangles[i]
        assert 0 <= t.material < d.n_materials


TOKENS = ["nan", "inf", "-inf", "1e999", "-1e999", "1e-999", "0", "-0", "-1", "1", "2", "99999999999999999999", "-99999999999999999999", "4294967296", "2147483648",
          "/", "//", "1/", "/1", "1//1", "1/1/1", "-1/-1/-1", "0/0/0", "", " ", "\t", "#", "v", "vt", "vn", "f", "usemtl", "mtllib", "newmtl", "Kd", "map_Kd", "map_bump",
          "-bm", "g", "o", "s", "l", "p", "\x00", "\xff\xfe", "a" * 300, "../../../etc/passwd", "spider.mtl", "x.mtl", "SpiderTex.jpg", "1.0.0", "1e", "+", "-", "."]
N_OBJ_MUTATIONS = int(os.environ.get("SOL_TEST_OBJ_MUTATIONS", "250"))  # (a longer campaign: SOL_TEST_OBJ_MUTATIONS=20000 SOL_TEST_OBJ_SEED=k)


def _mutate_text(text, rng):
    """One to four edits of a text file: a token replaced, a line dropped / doubled / moved, bytes inserted, the file cut short."""
    lines = text.split("\n")
    for _ in range(int(rng.integers(1, 5))):
        what = int(rng.integers(7))
        i = int(rng.integers(len(lines))) if lines else 0
        if not lines:
            break
        if what <= 2:
            toks = lines[i].split(" ")
            toks[int(rng.integers(len(toks)))] = TOKENS[int(rng.integers(len(TOKENS)))]
            lines[i] = " ".join(toks)
        elif what == 3:
            del lines[i]
        elif what == 4:
            lines.insert(int(rng.integers(len(lines) + 1)), lines[i])
        elif what == 5:
            k = int(rng.integers(len(lines[i]) + 1))
            lines[i] = lines[i][:k] + "".join(chr(int(c)) for c in rng.integers(0, 256, int(rng.integers(1, 9)))) + lines[i][k:]
        else:
            lines = lines[: int(rng.integers(len(lines) + 1))]
            if lines:
                lines[-1] = lines[-1][: int(rng.integers(len(lines[-1]) + 1))]
    return "\n".join(lines)


@pytest.mark.parametrize("base", ["boxWithMat", "triWithHeightMap", "spider"])
def test_mutated_obj_and_mtl_files_are_errors_or_models(base, tmp_path):
    """The hand-written malformed files above, made many: the reference's own OBJ + MTL files with one to four random edits each (tokens replaced by
    numbers no type holds, indices of every sign and size, stray slashes, keywords in the wrong place, raw bytes, lines dropped, doubled and cut),
    alternately in the .obj and the .mtl. An error string or a model whose triangles name existing materials; never a crash (also under ASan + UBSan:
    tests/tools/sanitize.sh). Image files a mutated `map_Kd` names are the directory's real ones, a missing one, or not an image at all."""
    src = SPIDER_DIR if base == "spider" else OBJ_DIR
    root = tmp_path / "m"
    shutil.copytree(src, root)
    obj0 = open(os.path.join(src, base + ".obj"), encoding="latin-1").read()
    mtl0 = open(os.path.join(src, base + ".mtl"), encoding="latin-1").read()
    (root / "x.mtl").write_bytes(b"newmtl a\nKd 1 0 0\n")
    rng = np.random.default_rng({"boxWithMat": 31, "triWithHeightMap": 32, "spider": 33}[base] + 1000 * int(os.environ.get("SOL_TEST_OBJ_SEED", "0")))
    n = N_OBJ_MUTATIONS if base != "spider" else max(40, N_OBJ_MUTATIONS // 6)  # (the spider is 106 kB)
    models = 0
    for k in range(n):
        obj, mtl = (_mutate_text(obj0, rng), mtl0) if k % 2 == 0 else (obj0, _mutate_text(mtl0, rng))
        (root / (base + ".obj")).write_bytes(obj.encode("latin-1"))
        (root / (base + ".mtl")).write_bytes(mtl.encode("latin-1"))
        try:
            sc = _scene_of(lambda b: b.load_obj(str(root) + os.sep, base + ".obj"))
        except HostError as e:
            assert str(e)
            continue
        models += 1
        d = sc.desc
        for i in range(d.n_triangles):
            assert 0 <= d.triangles[i].material < d.n_materials
    assert models > n // 10  # (most edits leave a readable file)


def test_vertex_numbers_follow_rusts_f32_grammar(tmp_path):
    """tobj reads `v` / `vt` with `str::parse::<f32>`: an optional sign, digits with an optional fraction and exponent, or inf / infinity / nan - the whole
    token, nothing around it; values beyond f32 round to +-inf / 0. The loader's scan (std::from_chars, no allocation per line) is held to the same grammar:
    a C hex float or a trailing unit is an error, as it is for the reference."""
    f32 = lambda x: float(np.float32(x))
    good = {"1.5": 1.5, "+1.5": 1.5, "-2": -2.0, ".5": 0.5, "5.": 5.0, "1e3": 1000.0, "+.5E-3": f32(0.0005), "1e999": float("inf"), "-1e999": float("-inf"),
            "1e-999": 0.0, "3.4028234e38": f32(3.4028234e38), "0.1": f32(0.1), "16777217": 16777216.0}
    for tok, want in good.items():
        (tmp_path / "g.obj").write_text(f"v {tok} 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
        sc = _scene_of(lambda b: b.load_obj(str(tmp_path) + os.sep, "g.obj"))
        assert sc.desc.n_triangles == 1
        xs = sorted({float(v[0]) for tri in _triangles(sc) for v in tri} - {0.0, 1.0}) or [0.0 if want == 0.0 else 1.0]
        if np.isfinite(want) and want not in (0.0, 1.0):
            assert xs == [want], (tok, xs, want)
    for tok in ("0x1p3", "1.5abc", "1,5", "--1", "+-1", "++1", "+", "-", ".", "e5", "1e", "1e+", "1_000", ""):
        (tmp_path / "b.obj").write_text(f"v {tok} 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n" if tok else "v  \nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
        with pytest.raises(HostError):
            _scene_of(lambda b: b.load_obj(str(tmp_path) + os.sep, "b.obj"))
