// sol_shade.h -- surface reconstruction, textures, materials and pdfs on the device.
// Mirrors src/material/mod.rs, src/material/texture.rs, src/pdf.rs and the pdf_value / random_direction methods of
// src/hittable/{quad,triangle,sphere}.rs, formula by formula (citations at each function).
#pragma once
#include "sol_trace.h"

struct Surface {
  f3 p;          // hit_point
  Onb onb;       // tangent frame handed to RayHit::new, normal already flipped to face the ray
  f3 normal;     // RayHit.normal = material.get_transformed_normal(onb, uv)
  float u, v;    // RayHit.uv
  float t;       // RayHit.ray_length
  int mat;
  bool front;
};

// Texture::color (src/material/texture.rs:121-123,170-179) + rgb_to_vec3 (src/util/rgb_color.rs:37-43)
template <bool COUNT>
DEV f3 tex_color(const DevScene& S, int id, float u, float v, Counters& cnt) {
  const DTex T = ldg_rec(S.texs + id);
  if (T.kind == SOL_TEX_SOLID) return mk3(T.r, T.g, T.b);
  if (COUNT) cnt.texel_fetches++;
  float au = fabsf(u), av = fabsf(v);
  float fu = au - floorf(au);
  float fv = 1.0f - (av - floorf(av));
  float x = fu * ((float)T.w - 1.0f), y = fv * ((float)T.h - 1.0f);
  uint32_t xi = x >= 0.0f ? (x < 4294967296.0f ? (uint32_t)x : 0xFFFFFFFFu) : 0u;
  uint32_t yi = y >= 0.0f ? (y < 4294967296.0f ? (uint32_t)y : 0xFFFFFFFFu) : 0u;
  xi = min(xi, T.w - 1u);
  yi = min(yi, T.h - 1u);
  const uint8_t* px = S.texels + T.offset + ((size_t)yi * T.w + xi) * 3;
  const float s = (float)(1.0 / 255.);
  return mk3((float)ldg_u8(px) * s, (float)ldg_u8(px + 1) * s, (float)ldg_u8(px + 2) * s);
}

// The albedo colour of material record `m`: a solid colour sits in the record, an image goes through the texture record.
template <bool COUNT>
DEV f3 albedo_color(const DevScene& S, const DMat& m, float u, float v, Counters& cnt) {
  if (m.flags & DMAT_ALBEDO_SOLID) return mk3(m.ar, m.ag, m.ab);
  return tex_color<COUNT>(S, m.albedo, u, v, cnt);
}

// Geometry of the closest hit, recomputed from (ray, t, u, v) with the formulas of the primitives' `hit`.
template <bool COUNT>
DEV void build_surface(const DevScene& S, f3 o, f3 d, const Hit& h, const Rng& rng, uint32_t depth, Surface& sf) {
  const uint32_t kind = SOL_REF_KIND(h.ref), idx = SOL_REF_INDEX(h.ref);
  sf.t = h.t;
  sf.p = o + d * h.t;  // Ray::at (geo/mod.rs:288-290)
  if (kind == SOL_REF_TRIANGLE) {  // triangle.rs:145-172
    const float4* sp = reinterpret_cast<const float4*>(S.tri_shade + idx);
    const float4 a = ldg_f4(sp), b = ldg_f4(sp + 1), c = ldg_f4(sp + 2), e = ldg_f4(sp + 3);
    float uv0 = 1.0f - h.u - h.v;
    sf.u = uv0 * b.w + h.u * e.x + h.v * e.z;
    sf.v = uv0 * c.w + h.u * e.y + h.v * e.w;
    f3 n = mk3(a.x, a.y, a.z);
    sf.front = dot3(d, n) < 0.0f;
    if (!sf.front) n = neg3(n);
    sf.onb = Onb{mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), n};
    sf.mat = (int32_t)__float_as_uint(a.w);  // DTriShade::mat
  } else if (kind == SOL_REF_QUAD) {  // quad.rs:175-193
    const DQuad Q = ldg_rec(S.quads + idx);
    f3 n = mk3(Q.nx, Q.ny, Q.nz);
    // fp32 contract, seventh rule (oracle.cpp hit_quad): the hit point goes back onto the quad's plane n . x = d - `o + t d` from a distant origin lands
    // ~1e-4 beside it, and a grazing ray scattered from a point behind the plane re-hits the same quad
    sf.p = sf.p + n * (Q.d - dot3(n, sf.p));
    sf.front = dot3(d, n) < 0.0f;
    if (!sf.front) n = neg3(n);
    sf.onb = Onb{unit3(mk3(Q.ux, Q.uy, Q.uz)), unit3(mk3(Q.vx, Q.vy, Q.vz)), n};
    sf.u = h.u; sf.v = h.v;
    sf.mat = Q.mat;
  } else if (kind == SOL_REF_SPHERE) {  // sphere.rs:83-107,134-140
    const DSphere Sp = ldg_rec(S.spheres + idx);
    f3 n = sf.p - mk3(Sp.cx, Sp.cy, Sp.cz);
    f3 normal = unit3(n);
    // fp32 contract, fifth rule (oracle.cpp hit_sphere): the hit point goes back onto the sphere - t of the fp32 quadratic is off by
    // thousandths for distant origins, the direction centre -> point is not; normal, uv and tangents stay as computed from the raw point
    sf.p = mk3(Sp.cx, Sp.cy, Sp.cz) + n * (Sp.radius / len3(n));
    sf.mat = Sp.mat;
    if (ldg_u32(&S.mats[Sp.mat].flags) & DMAT_NEEDS_UV) {  // uv and tangents are consumed only by image textures
      float theta = acos_r(-normal.y);
      float phi = -atan2_r(normal.z, normal.x) + SOL_PI;
      sf.u = phi / (2.0f * SOL_PI);
      sf.v = theta / SOL_PI;
      f3 tangent = unit3(cross3(mk3(0.f, 1.f, 0.f), n));
      sf.onb.tangent = tangent;
      sf.onb.bi_tangent = cross3(n, tangent);
    } else {
      sf.u = sf.v = 0.0f;
      sf.onb.tangent = sf.onb.bi_tangent = mk3(0.f, 0.f, 0.f);
    }
    sf.front = dot3(d, normal) < 0.0f;
    if (!sf.front) normal = neg3(normal);
    sf.onb.normal = normal;
  } else {  // SOL_REF_MEDIUM: constant_medium.rs:63-77
    sf.onb = Onb{mk3(1.f, 1.f, 1.f), mk3(1.f, 1.f, 1.f), medium_normal(rng, idx, depth)};
    sf.u = sf.v = 0.0f;
    sf.front = false;
    sf.mat = ldg_i32(&S.mediums[idx].mat);
  }
}

// Material::get_transformed_normal (material/mod.rs:108-110,209-213,251-255,304-308,438-444) with
// transform_normal_by_map (:386-389); applied to the closest hit only (DESIGN.md "Deviations").
template <bool COUNT>
DEV f3 transformed_normal(const DevScene& S, int mid, const Surface& sf, Rng& rng, Counters& cnt) {
  DMat m = ldg_rec(S.mats + mid);
  for (int guard = 0; guard < 16 && m.kind == SOL_MAT_BLEND; ++guard) m = ldg_rec(S.mats + (rnd(rng) > m.param ? m.m1 : m.m2));
  if (m.kind <= SOL_MAT_DIELECTRIC && m.normal >= 0) {
    f3 n = tex_color<COUNT>(S, m.normal, sf.u, sf.v, cnt) * 2.0f - mk3(1.f, 1.f, 1.f);
    return onb_local(sf.onb, n);
  }
  return sf.onb.normal;
}

// ---- light sampling (pdf.rs:75-102) ------------------------------------------------------------------------------
// Hittable::pdf_value of quad (quad.rs:132-143), triangle (triangle.rs:100-112), sphere (sphere.rs:40-56)
template <bool COUNT, bool STRICT = false>
DEV float light_pdf_value(const DevScene& S, uint32_t ref, f3 origin, f3 dir, Counters& cnt) {
  const uint32_t kind = SOL_REF_KIND(ref), idx = SOL_REF_INDEX(ref);
  const float inf = __builtin_huge_valf();
  if (kind == SOL_REF_QUAD) {
    const DQuad Q = ldg_rec(S.quads + idx);
    if (COUNT) cnt.quad_tests++;
    float t, u, v;
    if (!quad_test(Q, origin, dir, RAY_MIN_F, inf, t, u, v)) return 0.0f;
    f3 n = mk3(Q.nx, Q.ny, Q.nz);
    if (!(dot3(dir, n) < 0.0f)) n = neg3(n);
    float ds = t * t * len2(dir);
    float cosine = fabsf(dot3(dir, n) / len3(dir));
    return ds / (cosine * Q.area);
  }
  if (kind == SOL_REF_TRIANGLE) {
    const DTri T = ldg_rec(S.tris + idx);
    const DTriShade Ts = ldg_rec(S.tri_shade + idx);
    if (COUNT) cnt.triangle_tests++;
    float t, u, v;
    if (!tri_test(T, origin, dir, RAY_MIN_F, inf, t, u, v)) return 0.0f;
    // (no consistency rule here, in scenes with needle triangles either: the rule makes SEARCH results independent of the tree, and this
    // is a one-primitive test; a refused hit would report pdf 0 for a direction that light_random_direction generates densely - oracle.cpp)
    f3 n = mk3(Ts.nx, Ts.ny, Ts.nz);
    if (!(dot3(dir, n) < 0.0f)) n = neg3(n);
    float ds = t * t * len2(dir);
    float cosine = fabsf(dot3(dir, n) / len3(dir));
    return ds / (cosine * T.area);
  }
  if (kind == SOL_REF_SPHERE) {
    const DSphere Sp = ldg_rec(S.spheres + idx);
    if (COUNT) cnt.sphere_tests++;
    float t;
    if (!sphere_test(Sp, origin, dir, RAY_MIN_F, inf, S.sphere_slack, t)) return 0.0f;
    float cos_theta_max = sol_sqrt(1.0f - Sp.radius * Sp.radius / len2(mk3(Sp.cx, Sp.cy, Sp.cz) - origin));
    float solid_angle = 2.0f * SOL_PI * (1.0f - cos_theta_max);
    return 1.0f / solid_angle;
  }
  return 0.0f;
}
// Hittable::random_direction of quad (quad.rs:145-148), triangle (triangle.rs:114-117), sphere (sphere.rs:58-62,142-153)
DEV f3 light_random_direction(const DevScene& S, uint32_t ref, uint32_t light_index, f3 origin, Rng& rng) {
  const uint32_t kind = SOL_REF_KIND(ref), idx = SOL_REF_INDEX(ref);
  if (kind == SOL_REF_QUAD) {
    const DQuad Q = ldg_rec(S.quads + idx);
    float r1 = rnd(rng), r2 = rnd(rng);
    return mk3(Q.qx, Q.qy, Q.qz) + mk3(Q.ux, Q.uy, Q.uz) * r1 + mk3(Q.vx, Q.vy, Q.vz) * r2 - origin;
  }
  if (kind == SOL_REF_TRIANGLE) {
    const DTri T = ldg_rec(S.light_tri + light_index);  // the reference's frame, not the rotated intersect record (sol_types.h)
    float r1 = rnd(rng), r2 = rnd(rng);
    return mk3(T.v0x, T.v0y, T.v0z) + mk3(T.e1x, T.e1y, T.e1z) * r1 + mk3(T.e2x, T.e2y, T.e2z) * r2 - origin;
  }
  const DSphere Sp = ldg_rec(S.spheres + idx);
  f3 direction = mk3(Sp.cx, Sp.cy, Sp.cz) - origin;
  Onb uvw = onb_new(direction);
  float ds = len2(direction);
  float r1 = rnd(rng), r2 = rnd(rng);
  float z = 1.0f + r2 * (sol_sqrt(1.0f - Sp.radius * Sp.radius / ds) - 1.0f);
  float c, s;
  sincos2pi(r1, c, s);
  float zz = sol_sqrt(1.0f - z * z);
  return onb_local(uvw, mk3(c * zz, s * zz, z));
}
template <bool COUNT, bool STRICT = false>
DEV float container_pdf_value(const DevScene& S, f3 origin, f3 dir, Counters& cnt) {  // pdf.rs:89-96
  float sum = 0.0f;
  if (S.n_lights == 1u) return light_pdf_value<COUNT, STRICT>(S, S.light0, origin, dir, cnt) / 1.0f;  // (x / 1 == x: same value as the loop)
  for (uint32_t i = 0; i < S.n_lights; ++i) sum += light_pdf_value<COUNT, STRICT>(S, ldg_u32(S.lights + i), origin, dir, cnt);
  return sum / (float)S.n_lights;
}
DEV f3 container_pdf_generate(const DevScene& S, f3 origin, Rng& rng) {  // pdf.rs:98-101
  uint32_t i = rnd_index(rng, S.n_lights);  // (the draw is consumed also when there is one light: same stream as the oracle)
  return light_random_direction(S, S.n_lights == 1u ? S.light0 : ldg_u32(S.lights + i), S.n_lights == 1u ? 0u : i, origin, rng);
}
DEV f3 random_in_unit_sphere(Rng& rng) {  // vec3.rs:380-392 (bound never reached: (1-pi/6)^80)
  f3 p = mk3(0.f, 0.f, 0.f);
  for (int it = 0; it < 80; ++it) {
    p.x = rnd_range(rng, -1.0f, 1.0f);
    p.y = rnd_range(rng, -1.0f, 1.0f);
    p.z = rnd_range(rng, -1.0f, 1.0f);
    if (len2(p) < 1.0f) break;
  }
  return p;
}
DEV f3 random_cosine_direction(Rng& rng) {  // vec3.rs:417-428
  float r1 = rnd(rng), r2 = rnd(rng);
  float r2_sqrt = sol_sqrt(r2);
  float c, s;
  sincos2pi(r1, c, s);
  return mk3(c * r2_sqrt, s * r2_sqrt, sol_sqrt(1.0f - r2));
}

// RayScatter (material/mod.rs:60-93)
#define SCATTER_PDF 0
#define SCATTER_BASIC 1
#define SCATTER_EMISSION 2
struct Scatter {
  int type;
  f3 color;
  f3 dir;            // direction of the scattered ray (origin = hit point)
  float probability; // ScatterPdf
  float af;          // ScatterEmission.attenuation_factor
  bool has_af;
};

// Materials::scatter (material/mod.rs:191-207 Lambertian, :239-249 Metal, :279-302 Dielectric, :359-368 DiffuseLight,
// :396-410 Isotropic, :430-436 Blend)
template <bool COUNT, bool STRICT = false>
DEV void scatter(const DevScene& S, f3 ray_dir, const Surface& sf, Rng& rng, Scatter& sc, Counters& cnt) {
  DMat m = ldg_rec(S.mats + sf.mat);
  for (int guard = 0; guard < 16 && m.kind == SOL_MAT_BLEND; ++guard) m = ldg_rec(S.mats + (rnd(rng) > m.param ? m.m1 : m.m2));
  if (COUNT) cnt.shades++;
  sc.has_af = false; sc.af = 0.0f; sc.probability = 0.0f; sc.dir = mk3(0.f, 0.f, 0.f);
  if (m.kind == SOL_MAT_LAMBERTIAN) {
    sc.type = SCATTER_PDF;
    sc.color = albedo_color<COUNT>(S, m, sf.u, sf.v, cnt);
    Onb uvw = onb_new(sf.normal);  // CosinePdf::new (pdf.rs:58)
    f3 dir;
    if (rnd(rng) < 0.5f) dir = container_pdf_generate(S, sf.p, rng);  // mix_generate (pdf.rs:42-48)
    else dir = onb_local(uvw, random_cosine_direction(rng));
    f3 udir = unit3(dir);
    float cos_pdf = fmaxf(dot3(udir, uvw.normal) / SOL_PI, 0.0f);                                 // CosinePdf::value
    float mix = 0.5f * container_pdf_value<COUNT, STRICT>(S, sf.p, dir, cnt) + 0.5f * cos_pdf;            // mix_value
    float cos_theta = dot3(sf.normal, udir);                                                      // scattering_pdf_value
    float scattering = cos_theta < 0.0f ? 0.0f : cos_theta / SOL_PI;
    sc.dir = dir;
    sc.probability = scattering / mix;
  } else if (m.kind == SOL_MAT_METAL) {
    sc.type = SCATTER_BASIC;
    f3 reflected = reflect3(unit3(ray_dir), sf.normal);
    sc.color = albedo_color<COUNT>(S, m, sf.u, sf.v, cnt);
    sc.dir = reflected + random_in_unit_sphere(rng) * m.param;
  } else if (m.kind == SOL_MAT_DIELECTRIC) {
    sc.type = SCATTER_BASIC;
    float ratio = sf.front ? 1.0f / m.param : m.param;
    f3 ud = unit3(ray_dir);
    float cos_theta = fminf(dot3(neg3(ud), sf.normal), 1.0f);
    float sin_theta = sol_sqrt(1.0f - cos_theta * cos_theta);
    bool refl = ratio * sin_theta > 1.0f;
    if (!refl) {  // reflectance (mod.rs:312-316); the draw happens only when refraction is possible
      float r0 = (1.0f - ratio) / (1.0f + ratio);
      r0 = r0 * r0;
      float x = 1.0f - cos_theta, x2 = x * x, x4 = x2 * x2;
      refl = r0 + (1.0f - r0) * (x4 * x) > rnd(rng);
    }
    sc.dir = refl ? reflect3(ud, sf.normal) : refract3(ud, sf.normal, ratio);
    sc.color = albedo_color<COUNT>(S, m, sf.u, sf.v, cnt);
  } else if (m.kind == SOL_MAT_DIFFUSE_LIGHT) {
    sc.type = SCATTER_EMISSION;
    sc.color = sf.front ? albedo_color<COUNT>(S, m, sf.u, sf.v, cnt) : mk3(0.f, 0.f, 0.f);
    sc.has_af = !(m.flags & DMAT_PARAM_NONE);
    sc.af = m.param;
  } else {  // SOL_MAT_ISOTROPIC
    sc.type = SCATTER_PDF;
    sc.color = albedo_color<COUNT>(S, m, sf.u, sf.v, cnt);
    f3 dir;
    if (rnd(rng) < 0.5f) dir = container_pdf_generate(S, sf.p, rng);
    else dir = unit3(random_in_unit_sphere(rng));  // SpherePdf::generate (pdf.rs:121-124)
    const float sphere_pdf = (float)(1. / (4. * 3.14159265358979323846));
    float mix = 0.5f * container_pdf_value<COUNT, STRICT>(S, sf.p, dir, cnt) + 0.5f * sphere_pdf;
    sc.dir = dir;
    sc.probability = sphere_pdf / mix;
  }
}
