// solstrale_obj.cpp -- scene ingest before the path: Wavefront OBJ + MTL -> triangles, the reference's loader
// (src/loader/obj.rs:38-136). SURVEY.md 8f rank 2.
//
// The reference parses with the third-party crate tobj 4.0.2 (Cargo.toml:27; not under /root/reference), called as
// tobj::load_obj(path, &LoadOptions{triangulate: true, ..Default}) (obj.rs:45-53). What is restated here is the part of
// tobj's published behaviour that reaches the triangles:
//   * `v x y z [w]`, `vt u [v [w]]` parsed as f32 (tobj's `Mesh::positions: Vec<f32>`), later widened to f64 (obj.rs:139-145);
//   * `f` entries `v`, `v/vt`, `v//vn`, `v/vt/vn`, 1-based, negative = relative to the end of the list read so far;
//   * triangulate: a polygon a b c d .. becomes the fan (a,b,c), (a,c,d), ..; points and lines are dropped;
//   * a new model (mesh with ONE material id) starts at `o`, `g`, and at a `usemtl` that follows faces; `usemtl` of an unknown
//     name leaves the material id unset (None);
//   * `mtllib` files are read relative to the OBJ's directory; a missing one makes the material result an error while the
//     models still load; MTL: `newmtl`, `Kd r g b` (Option<[f32;3]>), `map_Kd file`, `map_Bump|map_bump|bump file`
//     (normal_texture; options such as `-bm 1` before the file name are skipped); everything else is ignored.
// Triangles keep file order, which is what Bvh::new sees.
#include <cstdlib>
#include <cstring>
#include <charconv>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "solstrale.hpp"

namespace solstrale {

namespace {

struct MtlMaterial {
  std::string name;
  bool has_diffuse = false;
  float diffuse[3] = {0.f, 0.f, 0.f};
  std::string diffuse_texture, normal_texture;  // empty = None
};

struct ObjMesh {
  std::vector<float> positions, texcoords;                  // as referenced: flat xyz / uv
  std::vector<uint32_t> indices, texcoord_indices;          // 3 per triangle
  bool has_material = false;
  size_t material_id = 0;
};

// (a plain scan: through an istringstream the tokens of a 1.1 M-face file cost more than everything after them)
std::vector<std::string> split_ws(const std::string& s) {
  std::vector<std::string> out;
  const char* p = s.data();
  const char* const end = p + s.size();
  auto is_ws = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; };
  while (p < end) {
    while (p < end && is_ws(*p)) ++p;
    const char* q = p;
    while (q < end && !is_ws(*q)) ++q;
    if (q > p) out.emplace_back(p, (size_t)(q - p));
    p = q;
  }
  return out;
}

// ---- the OBJ scan: no allocation per line (a 263 MB / 7.6 M-line file spent 2.6 s in per-token strings; 0.9 s this way) ----
struct Tok {
  const char* p;
  const char* e;
  size_t size() const { return (size_t)(e - p); }
  bool is(const char* word) const { return size() == std::strlen(word) && std::memcmp(p, word, size()) == 0; }
};

inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

void tokens_of(const char* p, const char* end, std::vector<Tok>& out) {
  out.clear();
  while (p < end) {
    while (p < end && is_ws(*p)) ++p;
    const char* q = p;
    while (q < end && !is_ws(*q)) ++q;
    if (q > p) out.push_back(Tok{p, q});
    p = q;
  }
}

// A whole token as f32, the way Rust's `str::parse::<f32>` reads it (tobj): an optional sign, decimal digits with an optional fraction and exponent, or
// inf / infinity / nan; nothing before, nothing after. Values beyond f32 go to +-inf / 0 as both Rust and strtof round them (std::from_chars only
// reports them: that rare token takes the slow road).
bool parse_f32(Tok t, float& out) {
  const char* p = t.p;
  if (p < t.e && *p == '+') {  // (from_chars takes no '+'; "+-1" and "++1" are not numbers)
    ++p;
    if (p < t.e && (*p == '+' || *p == '-')) return false;
  }
  if (p >= t.e) return false;
  const std::from_chars_result r = std::from_chars(p, t.e, out);
  if (r.ec == std::errc() && r.ptr == t.e) return true;
  if (r.ec == std::errc::result_out_of_range && r.ptr == t.e) {
    const std::string z(p, t.e);
    char* end = nullptr;
    out = std::strtof(z.c_str(), &end);
    return end != z.c_str() && *end == '\0';
  }
  return false;
}


std::string dir_of(const std::string& filepath) {
  size_t p = filepath.find_last_of("/\\");
  return p == std::string::npos ? std::string() : filepath.substr(0, p + 1);
}

// tobj::load_mtl. Returns false when the file cannot be opened.
bool load_mtl(const std::string& path, std::vector<MtlMaterial>& mats, std::map<std::string, size_t>& by_name) {
  std::ifstream f(path);
  if (!f) return false;
  std::string line;
  MtlMaterial* cur = nullptr;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::vector<std::string> t = split_ws(line);
    if (t.empty() || t[0][0] == '#') continue;
    if (t[0] == "newmtl") {
      MtlMaterial m;
      m.name = t.size() > 1 ? line.substr(line.find(t[1], line.find(t[0]) + t[0].size())) : std::string();  // (the rest of the line after the keyword)
      while (!m.name.empty() && (m.name.back() == ' ' || m.name.back() == '\t')) m.name.pop_back();
      by_name[m.name] = mats.size();
      mats.push_back(m);
      cur = &mats.back();
    } else if (cur && t[0] == "Kd" && t.size() >= 4) {
      auto num = [](const std::string& z, float& out) { return parse_f32(Tok{z.data(), z.data() + z.size()}, out); };
      if (num(t[1], cur->diffuse[0]) && num(t[2], cur->diffuse[1]) && num(t[3], cur->diffuse[2])) cur->has_diffuse = true;
    } else if (cur && t[0] == "map_Kd" && t.size() >= 2) {
      cur->diffuse_texture = t.back();
    } else if (cur && (t[0] == "map_Bump" || t[0] == "map_bump" || t[0] == "bump") && t.size() >= 2) {
      cur->normal_texture = t.back();
    }
  }
  return true;
}

// One `f` vertex: v[/vt[/vn]] -> zero-based indices (-1 = absent). Throws on malformed or out-of-range input.
void parse_face_vertex(Tok tok, size_t n_pos, size_t n_tex, long& vi, long& ti) {
  vi = -1; ti = -1;
  Tok parts[3] = {{tok.p, tok.p}, {nullptr, nullptr}, {nullptr, nullptr}};
  int k = 0;
  for (const char* c = tok.p; c < tok.e; ++c) {
    if (*c == '/') {
      if (++k > 2) throw std::runtime_error("face");
      parts[k] = Tok{c + 1, c + 1};
    } else {
      parts[k].e = c + 1;
    }
  }
  auto idx = [](Tok s, size_t n) -> long {
    const char* p = s.p;
    if (p < s.e && *p == '+') { ++p; if (p < s.e && (*p == '+' || *p == '-')) throw std::runtime_error("face index"); }
    long v = 0;
    const std::from_chars_result r = std::from_chars(p, s.e, v, 10);
    if (p >= s.e || r.ec != std::errc() || r.ptr != s.e || v == 0) throw std::runtime_error("face index");
    long z = v > 0 ? v - 1 : (long)n + v;
    if (z < 0 || (size_t)z >= n) throw std::runtime_error("face index out of range");
    return z;
  };
  vi = idx(parts[0], n_pos);
  if (parts[1].p && parts[1].size() > 0) ti = idx(parts[1], n_tex);
}

}  // namespace

Hittables Obj::load(const Transformer& transformation, Materials default_material, const ImageDecoder& decode) const {
  const bool verbose = std::getenv("SOL_VERBOSE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  auto stamp = [&](const char* what) {
    if (verbose) std::fprintf(stderr, "[solstrale] obj: %s at %.2f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
  };
  if (!default_material) default_material = Lambertian::create(SolidColor::create(1., 1., 1.), nullptr);  // obj.rs:43-44
  const std::string filepath = path + filename;                                                            // obj.rs:50

  // ---- tobj::load_obj ----
  std::ifstream f(filepath);
  if (!f) throw std::runtime_error("failed to load obj model from " + filepath);  // obj.rs:51-53
  std::vector<float> pos, tex;
  std::vector<ObjMesh> models;
  std::vector<MtlMaterial> mtl;
  std::map<std::string, size_t> mtl_by_name;
  bool mtl_error = false;
  ObjMesh cur;
  bool cur_has_mat = false;
  size_t cur_mat = 0;
  auto flush = [&]() {
    if (!cur.indices.empty()) models.push_back(std::move(cur));
    cur = ObjMesh{};
    cur.has_material = cur_has_mat;
    cur.material_id = cur_mat;
  };
  try {
    std::string text;
    {
      f.seekg(0, std::ios::end);
      const std::streamoff size = f.tellg();
      f.seekg(0, std::ios::beg);
      if (size > 0) {
        text.resize((size_t)size);
        f.read(&text[0], size);
        text.resize((size_t)f.gcount());
      }
    }
    stamp("read");
    std::vector<Tok> t;
    std::vector<long> vi, ti;
    const char* p = text.data();
    const char* const text_end = p + text.size();
    while (p < text_end) {
      const char* eol = (const char*)std::memchr(p, '\n', (size_t)(text_end - p));
      const char* const next = eol ? eol + 1 : text_end;
      const char* line_end = eol ? eol : text_end;
      if (line_end > p && line_end[-1] == '\r') --line_end;
      tokens_of(p, line_end, t);
      p = next;
      if (t.empty() || t[0].p[0] == '#') continue;
      if (t[0].is("v")) {
        float x, y, z;
        if (t.size() < 4 || !parse_f32(t[1], x) || !parse_f32(t[2], y) || !parse_f32(t[3], z)) throw std::runtime_error("v");
        pos.push_back(x); pos.push_back(y); pos.push_back(z);
      } else if (t[0].is("vt")) {
        float u, v = 0.f;
        if (t.size() < 2 || !parse_f32(t[1], u)) throw std::runtime_error("vt");
        if (t.size() >= 3 && !parse_f32(t[2], v)) throw std::runtime_error("vt");
        tex.push_back(u); tex.push_back(v);
      } else if (t[0].is("f")) {
        const size_t n = t.size() - 1;
        if (n < 3) continue;  // points and lines: dropped under triangulate
        vi.resize(n); ti.resize(n);
        for (size_t k = 0; k < n; ++k) parse_face_vertex(t[k + 1], pos.size() / 3, tex.size() / 2, vi[k], ti[k]);
        for (size_t k = 1; k + 1 < n; ++k) {  // fan (a, b, c), (a, c, d), ...
          const size_t tri[3] = {0, k, k + 1};
          for (size_t c : tri) {
            cur.indices.push_back((uint32_t)(cur.positions.size() / 3));
            cur.positions.push_back(pos[vi[c] * 3]); cur.positions.push_back(pos[vi[c] * 3 + 1]); cur.positions.push_back(pos[vi[c] * 3 + 2]);
            if (ti[c] >= 0) {
              cur.texcoord_indices.push_back((uint32_t)(cur.texcoords.size() / 2));
              cur.texcoords.push_back(tex[ti[c] * 2]); cur.texcoords.push_back(tex[ti[c] * 2 + 1]);
            }
          }
        }
      } else if (t[0].is("o") || t[0].is("g")) {
        flush();
      } else if (t[0].is("usemtl")) {
        std::string name = t.size() > 1 ? std::string(t[1].p, line_end) : std::string();  // (the rest of the line: a material's name may hold spaces)
        while (!name.empty() && (name.back() == ' ' || name.back() == '\t')) name.pop_back();
        auto it = mtl_by_name.find(name);
        const bool has = it != mtl_by_name.end();
        const size_t id = has ? it->second : 0;
        if (has != cur_has_mat || (has && id != cur_mat)) {
          cur_has_mat = has; cur_mat = id;
          if (!cur.indices.empty()) flush();
          cur.has_material = has; cur.material_id = id;
        }
      } else if (t[0].is("mtllib")) {
        for (size_t k = 1; k < t.size(); ++k)
          if (!load_mtl(dir_of(filepath) + std::string(t[k].p, t[k].e), mtl, mtl_by_name)) mtl_error = true;
      }
    }
    flush();
  } catch (const std::exception&) {
    throw std::runtime_error("failed to load obj model from " + filepath);
  }
  if (mtl_error) throw std::runtime_error("failed to load MTL file for " + filepath);  // obj.rs:54-55
  stamp("parsed");

  // ---- materials: everything Lambertian (obj.rs:57-77); keys are `i as i8`, key -1 = the default material ----
  std::map<int8_t, Materials> mat_map;
  mat_map[-1] = default_material;
  for (size_t i = 0; i < mtl.size(); ++i) {
    const MtlMaterial& m = mtl[i];
    Textures albedo;
    if (m.diffuse_texture.empty())
      albedo = m.has_diffuse ? SolidColor::create((double)m.diffuse[0], (double)m.diffuse[1], (double)m.diffuse[2])  // new_from_f32_array
                             : SolidColor::create(1., 1., 1.);
    else
      albedo = ImageMap::create(decode(path + m.diffuse_texture, "image"));  // ImageMap::load (texture.rs:137-153)
    Textures normal;
    if (!m.normal_texture.empty()) normal = normal_texture_from_bump_map(decode(path + m.normal_texture, "bump"));  // texture.rs:53-97
    mat_map[(int8_t)i] = Lambertian::create(albedo, normal);
  }

  // ---- triangles (obj.rs:79-133) ----
  std::vector<Hittables> triangles;
  for (const ObjMesh& mesh : models) {
    const int8_t material_id = mesh.has_material ? (int8_t)mesh.material_id : (int8_t)-1;
    auto it = mat_map.find(material_id);
    const Materials material = it == mat_map.end() ? default_material : it->second;
    const bool has_tex = !mesh.texcoords.empty() && mesh.texcoord_indices.size() == mesh.indices.size();
    for (size_t i = 0; i + 2 < mesh.indices.size(); i += 3) {
      auto vtx = [&](size_t k) {
        const size_t o = (size_t)mesh.indices[k] * 3;
        return Vec3{(double)mesh.positions[o], (double)mesh.positions[o + 1], (double)mesh.positions[o + 2]};
      };
      auto uv = [&](size_t k) {
        if (!has_tex) return Uv{0.f, 0.f};
        const size_t o = (size_t)mesh.texcoord_indices[k] * 2;
        return Uv{mesh.texcoords[o], mesh.texcoords[o + 1]};
      };
      triangles.push_back(Triangle::new_with_tex_coords(vtx(i), vtx(i + 1), vtx(i + 2), uv(i), uv(i + 1), uv(i + 2), material, transformation));
    }
  }
  stamp("triangles made");
  Hittables bvh = Bvh::create(std::move(triangles));  // obj.rs:135
  stamp("Bvh::new done");
  return bvh;
}

}  // namespace solstrale
