// sol_types.h -- fp32 device records (data layout in HBM; DESIGN.md "Data layout in HBM").
// Every field is the plain (float) cast of the f64 field of the same name in include/solstrale_hip.h; nothing is
// re-derived on upload except the child-box placement of DNode (boxes move from a node to its parent).
#pragma once
#include <stdint.h>

#define SOL_WG 256          // threads per workgroup = 4 wave64
#ifndef SOL_LDS_STACK
#define SOL_LDS_STACK 32    // traversal stack entries per lane kept in LDS (u32 each -> 32 KiB per workgroup)
#endif
#ifndef SOL_V1_MIN_WAVES
#define SOL_V1_MIN_WAVES 4  // waves per SIMD the one-path-per-lane kernel is compiled for (register budget 512 / waves).
                            // Throughput still grows with residency (C3, SOL_MAX_BPC: 1 wave 492, 2: 862, 3: 1141, 4: 1347
                            // Msamples/s), but 5 and 6 waves (with a 24-entry LDS stack) need 96 / 80 VGPRs and the 58 / 95
                            // spilled registers cost what the residency buys (1351 / 1375 on C3, C2 slower).
#endif
#define SOL_SPILL_STACK 480 // further entries per lane in a global spill area (deep trees, nested medium search)
#ifndef SOL_PRIM_MIN
#define SOL_PRIM_MIN 8      // trav_step: primitive tests wait until this many lanes of the wave hold one (1 = never wait).
                            // MI355X, 1080p x 128 spp, C3 / C2 ms: 1: 199.8 / 128.4, 4: 196.6 / 126.5, 8: 193.1 / 122.5,
                            // 12: 193.0 / 122.6, 16: 199.1 / 129.7, 24: 215.7 / 148.0
#endif
#ifndef SOL_FETCH_PRIO
#define SOL_FETCH_PRIO 33        // s_setprio around the loads of a search step (units: node fetch, tens: triangle fetch; 0 off): the wave
                                 // about to fetch issues its addresses and loads ahead of the waves in their arithmetic
#endif
#ifndef SOL_LOOP_PRIO
#define SOL_LOOP_PRIO 1          // priority of a wave inside the search loop (the service block runs at 0): a search is a chain of
                                 // dependent fetches, shading is throughput work. MI355X, 64 spp, ms with (loop, fetch) = (0, 0) /
                                 // (0, 1) / (1, 3): C3 70.6 / 70.0 / 69.4, C2 44.6 / 44.3 / 43.8, C1 10.56 / - / 10.39; the service block at 1 or 2 as
                                 // well: C3 70.3 (slower)
#endif
#define SOL_PACK_MAX_NODES (1u << 17)  // trees below this many wide nodes can keep their node groups as one stack dword (sol_trace.h, wide_visit<PACK>)
#define SOL_CHUNK 16        // samples per work item; fixed so that summation order never depends on the partition
#define SOL_TILE 8          // 8x8-pixel blocks = one wave's worth of adjacent work items
#define SOL_POOL_MAX 1024   // path slots per wave in the pool kernel (u16 queue entries: 2 KiB of LDS per wave)

// 64-byte BVH node: the boxes of BOTH children plus their references, so one fetch (4 x dwordx4 per lane)
// decides both children. Mirrors `Bvh{left,right,b_box}` (src/hittable/bvh.rs:14-18) with b_box hoisted.
struct __attribute__((aligned(16))) DNode {
  float lxmin, lxmax, lymin, lymax;
  float lzmin, lzmax, rxmin, rxmax;
  float rymin, rymax, rzmin, rzmax;
  uint32_t left, right, pad0, pad1;  // SOL_MAKE_REF encoding, NODE index = device node index
};
static_assert(sizeof(DNode) == 64, "DNode");

// 64-byte 7-wide node with 8-bit quantised child boxes and IMPLICIT child addresses, built at upload from a binary tree
// (sol_tree.h: WideBuilder + WideLayout). Half a 128-byte cache line, never straddling one: a visit costs 4 dwordx4 accesses per
// lane and at most one line fill. Why so small: the render kernel's time follows its vector-memory work 1:1 (measured with
// probe builds) - every lane visits another node, each 16-byte access is its own L1 look-up (~0.55 clk/lane on
// MI355X, tests/tools/micro/l1_gather.hip), and the first touch of a node is an L2 fill of its whole line. The first layout
// (96 B: the same boxes plus eight explicit 32-bit child references, 6 accesses, half of the nodes straddling two lines) took
// 1.5x the accesses; padding it to 128 B to avoid the straddling was 4 % SLOWER (footprint, L2 hit rate). Narrower trees were
// measured earlier with explicit references: 13.0 node visits per ray on C3 at width 8, 14.9 at 6, 18.1 at 4, 34.5 at 2.
//   child i of a node sits in SLOT s(i) in 0..6 (slot 7 does not exist: its byte in each of the six plane arrays carries the
//   two base indices). Slots are octants of the node (x<<2 | y<<1 | z), so that visiting slots in the order (s ^ ray_octant)
//   is front to back. Child box: lo = origin + q_lo[s] * scale, hi = origin + q_hi[s] * scale, scale_axis = 2^(e_axis +
//   DevScene::wide_emin - 127); it CONTAINS the child's padded fp32 box (the builder checks the decoded values): a pure cull.
//   imask bit s: slot s holds an inner node, namely node  base_inner + popcount(imask & ((1 << s) - 1));
//   lmask bit s: slot s holds a primitive, number          base_prim + popcount(lmask & ((1 << s) - 1))  of
//     kind 1: the triangle array, 2: the sphere array, 3: the quad array (the arrays are permuted at upload so that the
//     primitives of a node are consecutive), 0: DevScene::leaf_refs (full references; nodes with mixed kinds or a medium).
#define SOL_REF_WIDE 6u  // device-only reference kind
#define SOL_WIDE_CHILDREN 7
#define SOL_WIDE_MAX_INDEX 0x00FFFFFFu  // 24-bit base indices
struct __attribute__((aligned(64))) DWide {
  float ox, oy, oz;
  uint32_t meta;      // ex | ey << 5 | ez << 10 (exponents - wide_emin) | imask << 15 | lmask << 22 | leaf kind << 29
  uint32_t q[12];     // q_lo_x[8], q_lo_y[8], q_lo_z[8], q_hi_x[8], q_hi_y[8], q_hi_z[8]: one byte per slot; byte 7 of the three
                      // lo arrays = base_inner (bits 0-7, 8-15, 16-23), of the three hi arrays = base_prim
};
static_assert(sizeof(DWide) == 64, "DWide");
#define SOL_LEAF_REFS 0u
#define SOL_LEAF_TRIANGLES 1u
#define SOL_LEAF_SPHERES 2u
#define SOL_LEAF_QUADS 3u

// 48-byte triangle intersect record (src/hittable/triangle.rs:14-17 v0, v0v1, v0v2)
struct __attribute__((aligned(16))) DTri {
  float v0x, v0y, v0z, e1x;
  float e1y, e1z, e2x, e2y;
  float e2z;
  uint32_t dfs;
  int32_t mat;
  float area;         // (triangle.rs:27; read only by the light pdf of a triangle light)
};
static_assert(sizeof(DTri) == 48, "DTri");

// 64-byte triangle shading record (triangle.rs:18-27 uv0..2, normal, tangent, bi_tangent) + the material index, so that shading
// a hit touches this one record (half a cache line) and not the intersect record again
struct __attribute__((aligned(16))) DTriShade {
  float nx, ny, nz;
  int32_t mat;
  float tx, ty, tz, u0;
  float bx, by, bz, v0;
  float u1, v1, u2, v2;
};
static_assert(sizeof(DTriShade) == 64, "DTriShade");

// 80-byte quad record (src/hittable/quad.rs:19-29)
struct __attribute__((aligned(16))) DQuad {
  float nx, ny, nz, d;
  float qx, qy, qz;
  uint32_t dfs;
  float wx, wy, wz;
  int32_t mat;
  float ux, uy, uz, area;
  float vx, vy, vz, pad;
};
static_assert(sizeof(DQuad) == 80, "DQuad");

// 32-byte sphere record (src/hittable/sphere.rs:15-20)
struct __attribute__((aligned(16))) DSphere {
  float cx, cy, cz, radius;
  uint32_t dfs;
  int32_t mat;
  uint32_t pad0, pad1;
};
static_assert(sizeof(DSphere) == 32, "DSphere");

// constant medium (src/hittable/constant_medium.rs:14-19)
struct __attribute__((aligned(16))) DMedium {
  uint32_t boundary;  // device reference of the boundary sub-tree
  int32_t mat;
  float nid;  // negative_inverse_density
  uint32_t dfs;
  float bxmin, bxmax, bymin, bymax, bzmin, bzmax;  // box of the boundary root (Bvh::hit tests it first)
  uint32_t pad0, pad1;
};
static_assert(sizeof(DMedium) == 48, "DMedium");

// 48-byte material record (src/material/mod.rs:134-150). A SolidColor albedo texture (texture.rs:101) is copied into the record:
// shading a hit is a chain of dependent fetches (shading record -> material -> texture -> texel -> light), and for a solid
// colour the texture record is one link less.
struct __attribute__((aligned(16))) DMat {
  int32_t kind, albedo, normal, m1;
  int32_t m2;
  float param;
  uint32_t flags;  // bit0: param is None (DiffuseLight.attenuation_factor); bit1: some texture below is an image; bit2: the albedo
  uint32_t pad;    //       texture is a solid colour, held in ar / ag / ab
  float ar, ag, ab;
  uint32_t pad2;
};
static_assert(sizeof(DMat) == 48, "DMat");
#define DMAT_PARAM_NONE 1u
#define DMAT_NEEDS_UV 2u
#define DMAT_ALBEDO_SOLID 4u

// 32-byte texture record (src/material/texture.rs:101,128-133)
struct __attribute__((aligned(16))) DTex {
  int32_t kind;
  uint32_t w, h, offset;  // offset: byte offset into texels (RGB8)
  float r, g, b, pad;
};
static_assert(sizeof(DTex) == 32, "DTex");

struct DCamera {
  float ox, oy, oz, llx, lly, llz, hx, hy, hz, vx, vy, vz, ux, uy, uz, wx, wy, wz, lens_radius;
};

struct DevScene {
  const DNode* nodes;
  const DWide* wides;   // 7-wide tree of the world (searches with t >= 0); binary nodes remain for medium boundaries
  const uint32_t* leaf_refs;  // references of the primitives of wide nodes with leaf kind 0
  uint32_t wroot;       // index of the world's root in `wides`
  uint32_t wide_emin;   // biased exponent that a DWide exponent field of 0 stands for
  const DTri* tris;
  const DTriShade* tri_shade;
  const DQuad* quads;
  const DSphere* spheres;
  const DMedium* mediums;
  const DMat* mats;
  const DTex* texs;
  const uint8_t* texels;
  const uint32_t* lights;
  uint32_t n_lights;
  uint32_t light0;      // the light's reference when there is exactly one (saves the dependent fetch of lights[0] per bounce)
  uint32_t root;                                          // device reference of the world
  float rxmin, rxmax, rymin, rymax, rzmin, rzmax;         // world root box (Bvh::hit's own b_box test)
  uint32_t width, height, shader, max_depth;
  float sphere_slack;  // half the fp32 box pad: tolerance of the sphere hit-point-in-own-box rule (sol_trace.h)
  float bgx, bgy, bgz;
  const float* env;      // EXTENSION (SolSceneDesc::env_*): latitude-longitude radiance map for rays that hit nothing; null = background
  uint32_t env_w, env_h;
  float env_scale;
  DCamera cam;
  // Work order of the one-path-per-lane kernel (sol_path.h, decode_item_ordered): block_order[k] = local block taken k-th, the
  // first n_first of them ("heavy": long paths, found by a counted probe at scene creation) with all their chunks up front;
  // null = identity. block_cost: where a counted render records, per local block, the ray count of its longest item (null otherwise).
  const uint32_t* block_order;
  uint32_t n_first;
  uint32_t* block_cost;
  // Tile partition: the image block (row-major over 8x8 blocks) behind local block lb of this rank. null: lb * world + rank (block b
  // belongs to rank b mod world); a table: the balanced partition (sol_scene_set_option SOL_OPT_BALANCED_PARTITION), the blocks dealt
  // out in the order of their cost in the creation probe. block_work: where a counted render adds up, per local block, its rays.
  const uint32_t* block_of_local;
  uint32_t* block_work;
  // Scenes with needle triangles (solstrale_hip.h, sol_scene_has_needles): tolerance of the triangle hit's consistency rule, 0.8
  // box pads; 0: the scene has none and the rule is off. (Last: the fields above keep the offsets the scalar loads were tuned around.)
  float tri_delta;
  // Triangle LIGHTS are sampled in the reference's own frame - Triangle::random_direction draws from the parallelogram at the FIRST
  // vertex (triangle.rs:114-117), which is not the vertex the intersect record starts at (solstrale_hip.h, sol_triangle_rotation):
  // light_tri[i] = (v0, v0v1, v0v2) of light i in the reference's order when light i is a triangle (a DTri whose other fields are unused).
  const DTri* light_tri;
};

struct RenderParams {
  uint32_t first_sample, n_samples, n_chunks;
  uint32_t rank, world;
  uint32_t n_local_blocks;   // 8x8 blocks owned by this rank
  uint32_t n_traced_blocks;  // v1: the first so many blocks of the work order are traced; the rest are background blocks (proved at scene
                             // creation to see nothing but the background), whose sums sol_fill_background_kernel writes
  uint32_t blocks_x;         // blocks per image row
  uint32_t seed_lo, seed_hi;
  uint32_t n_items;          // n_chunks * n_traced_blocks * 64
  uint32_t total_threads;    // grid * SOL_WG (spill stack stride)
  uint32_t pool_slots;       // pool kernel: path slots per wave (multiple of 64, <= SOL_POOL_MAX)
  uint32_t swap_min;         // pool kernel (sol_pool.hip): lanes whose search is over exchange it for a ready ray of the pool once this many wait
  uint32_t switch_below;     // v1: a wave leaves the search loop for shading once fewer than this many of its live lanes
                             // (in 64ths) are still searching and some lane waits; 0 = search until every lane is done
  // v1 writes every item's sum into the `partial` buffer: chunk sums at [chunk][slot], and behind them (from RGB triple stage_at)
  // the sample colours of the fine tail. Items [0, n_coarse) are whole 16-sample chunks; the items from
  // n_coarse on are the LAST pairs (block, chunk) of the work order handed out one sample at a time - item
  // n_coarse + ((j * 16 + sub) * 64 + pin) is sample `sub` of pixel `pin` of pair n_coarse / 64 + j - and their colours go to triple
  // stage_at + (j * 64 + pin) * 16 + sub, which sol_stage_resolve_kernel adds up in sample order. n_coarse == n_items: no fine tail.
  uint32_t n_coarse;
  uint32_t fine_count;       // samples of the last chunk (1 .. 16): fine items with sub >= fine_count do not exist
  uint32_t stage_at;
};

struct DevCounters {
  unsigned long long samples, rays, node_visits, sphere_tests, quad_tests, triangle_tests, shades, texel_fetches, max_stack;
  unsigned long long phase[6];
  unsigned long long primary_hits, path_len[6];
};
