/*
 * solstrale_hip.h -- C ABI of the MI355X (gfx950) path-tracing library `libsolstrale_hip.so`.
 *
 * This is the drop-in boundary for ONE hot path of DanielPettersson/Solstrale-Rust: the body of the
 * per-pass row loop of `Renderer::render` (reference src/renderer/mod.rs:241-291), i.e. everything under
 * `Renderer::ray_color` (src/renderer/mod.rs:164-206): BVH closest hit (src/hittable/), material scatter
 * and pdf importance sampling (src/material/mod.rs, src/pdf.rs), camera rays (src/camera.rs:77-89) and the
 * per-pixel accumulation (src/renderer/mod.rs:268,361-365).
 *
 * The reference has no FFI of its own on this path: the host (Rust `Renderer`, or the C++
 * mirror in solstrale-rust_amd/host/) keeps scene loading, the BVH builder (src/hittable/bvh.rs:61-162),
 * Camera::new (src/camera.rs:47-74), the pass loop, progress reporting and post-processing; it flattens its
 * `Hittables` tree into the POD arrays below and calls sol_scene_create / sol_render / sol_read.
 *
 * Every struct mirrors the fields of a reference type, in f64 exactly as the reference holds them; the
 * library converts to its fp32 device layout on upload (DESIGN.md "Data layout in HBM").
 *
 * All functions return 0 on success, a negative SOL_E* code otherwise; sol_last_error() gives the
 * thread-local message. Nothing here aborts the process. No torch types appear in any signature.
 */
#ifndef SOLSTRALE_HIP_H
#define SOLSTRALE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOL_ABI_VERSION 2  /* 2: SolSceneDesc ends with the optional environment map; a version-1 description (without those
                           * fields) is still accepted */

/* ---- error codes ---------------------------------------------------------------------------------- */
#define SOL_OK 0
#define SOL_EINVAL (-1)    /* malformed description / argument                                         */
#define SOL_ENOLIGHT (-2)  /* "Scene should have at least one light" (src/renderer/mod.rs:143-147)      */
#define SOL_EDEVICE (-3)   /* HIP runtime error / no GPU                                               */
#define SOL_EDEPTH (-4)    /* BVH deeper than the traversal stack supports                             */
#define SOL_ENOMEM (-5)

/* ---- child / primitive references ------------------------------------------------------------------
 * A 32-bit reference: kind in bits 31..28, index into the array of that kind in bits 27..0.
 * Mirrors `BvhItem::{Node, Leaf(Box<Hittables>), None}` (src/hittable/bvh.rs:21-25) with the leaf's
 * `Hittables` variant (src/hittable/mod.rs:47-61) made explicit. A nested `Bvh` held in a Leaf is
 * inlined by the flattener as SOL_REF_NODE (Bvh::hit is the same function, so this is exact). */
#define SOL_REF_NONE 0u
#define SOL_REF_NODE 1u
#define SOL_REF_SPHERE 2u
#define SOL_REF_QUAD 3u
#define SOL_REF_TRIANGLE 4u
#define SOL_REF_MEDIUM 5u
#define SOL_REF_KIND(r) ((uint32_t)(r) >> 28)
#define SOL_REF_INDEX(r) ((uint32_t)(r) & 0x0FFFFFFFu)
#define SOL_MAKE_REF(kind, index) ((((uint32_t)(kind)) << 28) | ((uint32_t)(index) & 0x0FFFFFFFu))

/* Axis-aligned box, `Aabb{x,y,z: Interval{min,max}}` (src/geo/mod.rs:39-46): xmin,xmax,ymin,ymax,zmin,zmax */
typedef struct SolAabb {
  double v[6];
} SolAabb;

/* `Bvh{left,right,b_box}` (src/hittable/bvh.rs:14-18) */
typedef struct SolBvhNode {
  SolAabb bbox;
  uint32_t left;  /* SOL_MAKE_REF(...) */
  uint32_t right; /* SOL_MAKE_REF(...) */
} SolBvhNode;

/* `Sphere{center,radius,mat,b_box}` (src/hittable/sphere.rs:15-20) */
typedef struct SolSphere {
  double center[3];
  double radius;
  SolAabb bbox;
  int32_t material;
  uint32_t dfs_index; /* position in the depth-first leaf order of the world tree (tie rule) */
} SolSphere;

/* `Quad{q,u,v,normal,d,w,mat,b_box,area}` (src/hittable/quad.rs:19-29) */
typedef struct SolQuad {
  double q[3], u[3], v[3], normal[3];
  double d;
  double w[3];
  double area;
  SolAabb bbox;
  int32_t material;
  uint32_t dfs_index;
} SolQuad;

/* `Triangle{v0,v0v1,v0v2,uv0..2,normal,tangent,bi_tangent,mat,b_box,area}` (src/hittable/triangle.rs:14-27) */
typedef struct SolTriangle {
  double v0[3], v0v1[3], v0v2[3];
  double normal[3], tangent[3], bi_tangent[3];
  double area;
  float uv0[2], uv1[2], uv2[2]; /* `Uv{f32,f32}` (src/geo/mod.rs:15-20) */
  SolAabb bbox;
  int32_t material;
  uint32_t dfs_index;
} SolTriangle;

/* `ConstantMedium{boundary,negative_inverse_density,phase_function}` (src/hittable/constant_medium.rs:14-19) */
typedef struct SolMedium {
  uint32_t boundary; /* reference to the boundary hittable (its own sub-tree of nodes/prims)            */
  int32_t material;  /* the Isotropic phase function                                                    */
  double negative_inverse_density;
  SolAabb bbox;
  uint32_t dfs_index;
  uint32_t _pad;
} SolMedium;

/* `Materials` (src/material/mod.rs:134-150) */
#define SOL_MAT_LAMBERTIAN 0
#define SOL_MAT_METAL 1
#define SOL_MAT_DIELECTRIC 2
#define SOL_MAT_DIFFUSE_LIGHT 3
#define SOL_MAT_ISOTROPIC 4
#define SOL_MAT_BLEND 5
typedef struct SolMaterial {
  int32_t kind;
  int32_t albedo_tex; /* Lambertian/Metal/Dielectric.albedo, DiffuseLight.tex, Isotropic.tex; -1 for Blend */
  int32_t normal_tex; /* `normal: Option<Textures>`; -1 = None                                          */
  int32_t m1, m2;     /* Blend.material_1 / material_2 (indices into materials), else -1                 */
  int32_t _pad;
  /* Metal.fuzz | Dielectric.index_of_refraction | Blend.blend_factor |
   * DiffuseLight.attenuation_factor with NaN standing for `None` (src/material/mod.rs:320-340) */
  double param;
} SolMaterial;

/* `Textures::{SolidColor(Vec3), ImageMap{image,max_x,max_y}}` (src/material/texture.rs:26-33,101,128-133) */
#define SOL_TEX_SOLID 0
#define SOL_TEX_IMAGE 1
typedef struct SolTexture {
  int32_t kind;
  uint32_t width, height; /* image only                                                                */
  uint32_t _pad;
  uint64_t texel_offset; /* byte offset of the first RGB8 texel in SolSceneDesc.texels                 */
  double rgb[3];         /* solid colour                                                               */
} SolTexture;

/* `Camera{origin,lower_left_corner,horizontal,vertical,u,v,lens_radius}` (src/camera.rs:35-43),
 * produced on the host by Camera::new (src/camera.rs:47-74). */
typedef struct SolCamera {
  double origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3];
  double lens_radius;
} SolCamera;

/* `Shaders` (src/renderer/shader.rs:33-44) */
#define SOL_SHADER_PATH_TRACING 0
#define SOL_SHADER_ALBEDO 1
#define SOL_SHADER_NORMAL 2
#define SOL_SHADER_SIMPLE 3

typedef struct SolSceneDesc {
  uint32_t abi_version; /* SOL_ABI_VERSION */
  uint32_t width, height; /* RenderConfig.width/height (src/renderer/mod.rs:26-30)                      */
  uint32_t shader_kind;   /* SOL_SHADER_*                                                               */
  uint32_t max_depth;     /* PathTracingShader.max_depth (src/renderer/shader.rs:48-50)                 */
  uint32_t root;          /* reference to `Scene.world` (src/renderer/mod.rs:63-72)                     */
  double background[3];   /* Scene.background_color                                                     */
  SolCamera camera;

  const SolBvhNode* nodes;      uint32_t n_nodes;
  const SolSphere* spheres;     uint32_t n_spheres;
  const SolQuad* quads;         uint32_t n_quads;
  const SolTriangle* triangles; uint32_t n_triangles;
  const SolMedium* mediums;     uint32_t n_mediums;
  const SolMaterial* materials; uint32_t n_materials;
  const SolTexture* textures;   uint32_t n_textures;
  const uint8_t* texels;        uint64_t n_texel_bytes;
  /* `Renderer.lights` = world.get_lights() in depth-first order (src/renderer/mod.rs:126,141;
   * src/hittable/bvh.rs:186-193): references to light primitives */
  const uint32_t* lights;       uint32_t n_lights;
  /* ---- abi_version >= 2: EXTENSION, not in the reference (which only has the constant `background_color`,
   * src/renderer/mod.rs:197-204; BASELINE.json config 5 names an "HDRI env light"). A latitude-longitude map of linear RGB
   * radiance, fp32, row 0 = up (+y): a ray that hits nothing returns env_scale * texel(direction) instead of `background`.
   * The direction is mapped like a point on the reference's unit sphere (calculate_sphere_uv, src/hittable/sphere.rs:134-140)
   * and the texel picked like an ImageMap's (nearest, src/material/texture.rs:170-179). It is NOT importance-sampled: the
   * scene still needs a light (Renderer::new). env_texels == NULL (or width/height 0) = no environment. */
  const float* env_texels;      uint32_t env_width, env_height;
  double env_scale;
} SolSceneDesc;

/* fp32 arithmetic contract, vertex order of a triangle's fp32 record. Moller-Trumbore works in the frame (v0; e1 = v1 - v0, e2 = v2 - v0);
 * its rounding error grows with |e1| |e2| / sin(angle between them), i.e. - the area being what it is - with the product of the two
 * edge lengths at v0. The fp32 records (device, and the oracle's float instantiation) therefore start at the vertex OPPOSITE THE LONGEST
 * EDGE: a cyclic rotation (v0, v1, v2) -> (v_k, v_k+1, v_k+2) - same winding, same normal, same set of points; the texture
 * coordinates rotate along, so the interpolated values are the same numbers up to rounding. k = 0 unless another start is strictly
 * better. A triangle that is a LIGHT is INTERSECTED through its rotated record like any other and SAMPLED in the reference's own
 * frame (Triangle::random_direction draws from the parallelogram at the first vertex, triangle.rs:114-117 - a set that depends on the
 * start): both sides keep the unrotated (v0, v0v1, v0v2) of every triangle light for that. For a strip-shaped needle of aspect 300:1
 * the rotation takes the test's noise down by that factor. f64 is untouched. */
static inline int sol_triangle_rotation(const SolTriangle* t) {
  const double a[3] = {t->v0v1[0], t->v0v1[1], t->v0v1[2]}, b[3] = {t->v0v2[0], t->v0v2[1], t->v0v2[2]};
  const double c[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
  const double l01 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2], l02 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
  const double l12 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
  /* start k: the edge opposite v_k is v_k+1 v_k+2: k = 0: v1v2 (l12), k = 1: v2v0 (l02), k = 2: v0v1 (l01) */
  int k = 0;
  double best = l12;
  if (l02 > best) { best = l02; k = 1; }
  if (l01 > best) { k = 2; }
  return k;
}
/* The rotated record: vertex, two edges and the order of the three texture coordinates (index into {uv0, uv1, uv2}). */
static inline void sol_triangle_rotated(const SolTriangle* t, int k, double v0[3], double e1[3], double e2[3], int uv_of[3]) {
  int i;
  for (i = 0; i < 3; ++i) {
    const double p0 = t->v0[i], a = t->v0v1[i], b = t->v0v2[i];
    if (k == 1) { v0[i] = p0 + a; e1[i] = b - a; e2[i] = -a; }
    else if (k == 2) { v0[i] = p0 + b; e1[i] = -b; e2[i] = a - b; }
    else { v0[i] = p0; e1[i] = a; e2[i] = b; }
  }
  uv_of[0] = k % 3; uv_of[1] = (k + 1) % 3; uv_of[2] = (k + 2) % 3;
}

/* fp32 arithmetic contract, needle triangles. fp32 Moller-Trumbore (src/hittable/triangle.rs:119-140 in single
 * precision) is ill-conditioned for triangles of extreme aspect: it accepts rays that pass many box pads beside the triangle (hundreds, in the
 * reference's vertex order; the rotation above leaves about one candidate in 10^5 beyond ONE thin pad on a mesh with 300:1 rods),
 * and whether such a phantom is seen would depend on which boxes a traversal tested. A scene HAS NEEDLES when some triangle's
 * longest edge squared is at least 2 * 32 times its area (aspect >= 32:1). For such scenes the fp32 contract - the device and the
 * oracle's float instantiation alike; in f64 nothing changes - (i) pads every box by SOL_NEEDLE_PAD * S * 2^-20 instead of S * 2^-20 and (ii)
 * counts a triangle hit only if the ray's point o + t*d and the triangle's point v0 + u*e1 + v*e2 agree within 0.8 pads in
 * every coordinate: an accepted hit then lies inside every box around its part of the triangle, whatever the tree. Both sides decide
 * with THIS function. */
#ifndef SOL_NEEDLE_ASPECT
#define SOL_NEEDLE_ASPECT 32.0
#endif
#ifndef SOL_NEEDLE_PAD
#define SOL_NEEDLE_PAD 4.0f /* such scenes' box pad, in thin pads (S * 2^-20); the rule's tolerance is 0.8 of it */
#endif
static inline int sol_scene_has_needles(const SolSceneDesc* d) {
  uint32_t i;
  for (i = 0; i < d->n_triangles; ++i) {
    const SolTriangle* t = &d->triangles[i];
    const double a[3] = {t->v0v1[0], t->v0v1[1], t->v0v1[2]}, b[3] = {t->v0v2[0], t->v0v2[1], t->v0v2[2]};
    const double c[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    double l2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    const double lb = b[0] * b[0] + b[1] * b[1] + b[2] * b[2], lc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    if (lb > l2) l2 = lb;
    if (lc > l2) l2 = lc;
    if (!(l2 < 2.0 * SOL_NEEDLE_ASPECT * t->area)) return 1; /* (a degenerate or NaN triangle counts as a needle) */
  }
  return 0;
}

/* Counters of the last instrumented render (sol_render_counted); zero otherwise. Definitions are the
 * ones SURVEY.md 8d / DESIGN.md use for algorithmic bytes. */
typedef struct SolStats {
  uint64_t samples;       /* ray_color(primary,0,0) evaluations                                         */
  uint64_t rays;          /* world closest-hit queries (src/renderer/mod.rs:165), any depth             */
  uint64_t node_visits;   /* device BVH nodes fetched (sol_record_sizes: 64-B 7-wide nodes of the world)  */
  uint64_t sphere_tests, quad_tests, triangle_tests; /* primitive hit evaluations incl. light pdf tests */
  uint64_t shades;        /* material scatter evaluations                                               */
  uint64_t texel_fetches;
  uint64_t max_stack;     /* deepest traversal stack use                                                */
  /* SIMD lane utilisation of the kernel's phases: [0]/[1] traverse, [2]/[3] shade, [4]/[5] generate, each pair =
   * (active lanes summed over executions, 64 x executions). Instrumentation only. */
  uint64_t phase[6];
} SolStats;

typedef struct SolScene SolScene; /* opaque handle: owns device memory, stream; one host thread at a time */

/* Number of visible HIP devices (0 if none / no driver). */
int sol_device_count(void);

/* Validates and deep-copies `desc` onto HIP device `device` (caller may free its arrays on return).
 * Replaces the buffer/camera/pool set-up of Renderer::render (src/renderer/mod.rs:223-234) and carries
 * Renderer::new's "Scene should have at least one light" check (src/renderer/mod.rs:143-147, SOL_ENOLIGHT). */
int sol_scene_create(const SolSceneDesc* desc, int device, SolScene** out);
void sol_scene_destroy(SolScene* scene);

/* Creation options (no reference analogue: the reference has one BVH builder and no scheduler to tune). All-zero = the
 * defaults of sol_scene_create. The SOL_* environment variables of DESIGN.md 9 remain developer overrides, read once here. */
#define SOL_TREE_AUTO 0    /* the default: the GPU build (SOL_TREE_DEVICE); when that fails - a primitive with a non-finite box,
                              more than 2^23 primitives, no scratch memory - the host candidates + probe (SolSceneInfo says so) */
#define SOL_TREE_REF 1     /* the reference's topology (src/hittable/bvh.rs:84-162), collapsed 8-wide                 */
#define SOL_TREE_SAH8 2    /* host binned-SAH rebuild, 8 / 16 / 64 bins                                              */
#define SOL_TREE_SAH16 3
#define SOL_TREE_SAH64 4
#define SOL_TREE_DEVICE 5  /* built on the GPU (Morton sort, PLOC clustering, 7-wide collapse: sol_build.hip; 8f rank 3) */
#define SOL_TREE_HOST_PROBE 6 /* the four host candidates (1..4) + counted probe renders on the device pick one          */
typedef struct SolCreateOptions {
  uint32_t size;            /* sizeof(SolCreateOptions): lets the struct grow                                         */
  int32_t world_tree;       /* SOL_TREE_*                                                                              */
  int32_t no_work_order_probe; /* 1: skip the 4-spp cost probe of the frame (heavy-first work order); for previews     */
  int32_t split_percent;    /* device build: triangle pre-splitting may add this many references, in percent of the primitive
                               count (a split triangle gets one record per part of it). 0: the default - a budget of 30, used
                               only when it shrinks the summed box area of the primitives below 85 % (meshes of uniform small
                               triangles stay unsplit); > 0: that budget (at most 1000), always used; < 0: no pre-splitting   */
  int32_t reinsertion_rounds; /* device build: rounds of parallel reinsertion after the clustering (every node looks for the place
                               where its sub-tree adds the least surface area; results do not change). 0: the default (8), < 0: none,
                               at most 1024 (the rounds end when nothing moves)                                                  */
  int32_t no_background_blocks; /* 1: do not look for background blocks (below, SolSceneInfo::background_blocks)            */
  int32_t reserved[2];
} SolCreateOptions;
int sol_scene_create_ex(const SolSceneDesc* desc, int device, const SolCreateOptions* options, SolScene** out);
/* Seconds sol_scene_create spent in: [0] host tree candidates, [1] uploads, [2] device tree build, [3] probe renders. */
int sol_scene_build_times(const SolScene* scene, double out[4]);

/* What sol_scene_create decided (diagnostic; no reference analogue). stack_bound: the host's bound on the traversal stack use of
 * any search of this scene, in dwords (2 per level of the 7-wide tree + the deepest medium boundary + 2): SolStats.max_stack of
 * a counted render never exceeds it, and the kernel built without a spill path is launched only when it is <= lds_stack. */
typedef struct SolSceneInfo {
  uint32_t size;            /* in: sizeof(SolSceneInfo)                                                               */
  uint32_t stack_bound, lds_stack, spill_stack;
  uint32_t tree_fallback;   /* 1: SOL_TREE_AUTO's device build failed and the host candidates were used (tree_note)    */
  char tree_name[32];       /* "device", "ref", "sah8", "sah16", "sah64"                                               */
  char tree_note[192];
  uint32_t split_references; /* device build: references that triangle pre-splitting added (0: none, or not kept)            */
  uint32_t split_triangles;  /* triangles with more than one reference                                                    */
  float split_area_ratio;    /* summed box area of the primitives' references after / before pre-splitting (the splits are kept
                                by default when this is below 0.85)                                                         */
  uint32_t reinsertion_moves; /* device build: sub-trees the reinsertion rounds moved                                       */
  float reinsertion_area_ratio; /* summed surface area of the binary tree's inner nodes after / before those rounds          */
  uint32_t partition_table;  /* 1: the balanced partition table is in force (SOL_OPT_BALANCED_PARTITION was set AND the creation
                                probe's block costs exist AND world > 1); 0: block b belongs to rank b % world                  */
  uint32_t partition_crc;    /* checksum of the block -> (rank, local block) mapping in force: equal on every rank of a job, or the
                                ranks render different partitions (each derives the table from its own probe)                     */
  uint32_t strict_triangles; /* 1: the scene has needle triangles (sol_scene_has_needles): fatter box pad, triangle consistency rule  */
  uint32_t background_blocks; /* 8x8 pixel blocks of which sol_scene_create PROVED that no camera ray of any of their pixels, whatever the
                                jitter, comes near a primitive's box (pinhole camera, constant background, the world a tree: the pyramid of
                                the block's rays against the world tree's boxes, conservatively). Every sample of such a pixel is the
                                background colour; sol_render adds those sums up in the reference's order without generating the
                                samples (SOL_OPT_BACKGROUND_BLOCKS). Images never depend on it. 0: none found, or not looked for     */
  uint32_t background_pixels; /* pixels of the image inside those blocks                                                        */
} SolSceneInfo;
int sol_scene_info(const SolScene* scene, SolSceneInfo* out);

/* Scheduler options of a live handle (take effect at the next sol_render; results never depend on them). */
#define SOL_OPT_SWITCH_BELOW 1        /* 0..64: a wave leaves the search loop when fewer 64ths of its lanes search      */
#define SOL_OPT_MAX_BLOCKS_PER_CU 2   /* 0 = as many as fit; n >= 1 caps resident workgroups per CU (occupancy studies)  */
#define SOL_OPT_KERNEL 3              /* 0 auto, 1 one-path-per-lane (product), 2 / 3 wavefront variants (A/B only)      */
#define SOL_OPT_WORK_ORDER 4          /* 0: plain chunk-major order, 1: heavy-first order from the creation probe        */
#define SOL_OPT_FINE_TAIL 5           /* quarters of a 16-sample item per resident lane that the END of a launch hands out one
                                         sample at a time (shorter tail; images unchanged); 0 off, -1 (default) decided by the
                                         creation probe                                                                      */
#define SOL_OPT_BALANCED_PARTITION 6  /* 1: sol_scene_set_partition / sol_comm_init deal the blocks out by their cost in the creation probe
                                         (a table behind the partition, the same on every rank) instead of b % world; for runs that use
                                         sol_gather / sol_read / sol_unpermute of THIS library - a caller with its own collective and
                                         un-permute keeps the default 0, whose layout it can compute. Set it on every rank, before
                                         sol_comm_init. Images never depend on it.                                              */
#define SOL_OPT_BACKGROUND_BLOCKS 7   /* 1 (default): the samples of background blocks (SolSceneInfo::background_blocks) are summed without
                                         being traced; 0: every sample of every pixel is generated and traced. Images never depend on it;
                                         counted renders (sol_render_counted) trace everything - their counters describe the whole
                                         algorithm - unless the value is 2: then they count exactly what a plain render does.      */
int sol_scene_set_option(SolScene* scene, int option, int64_t value);

/* Image-tile sharding for one-process-per-GPU runs (no reference analogue). The image is cut
 * into 8x8-pixel blocks, block b (row-major) belongs to rank b % world (or, with SOL_OPT_BALANCED_PARTITION, to the rank a
 * cost-sorted deal gives it). Each rank accumulates only its own blocks in a compact buffer of sol_accum_floats() floats:
 * [local_block][py][px][rgb]. Default rank 0/1. */
int sol_scene_set_partition(SolScene* scene, int rank, int world);
/* A caller-bound accumulator (sol_scene_bind_accum) whose size would change makes this fail with SOL_EINVAL: unbind
 * (sol_scene_bind_accum(scene, NULL, 0)) first, re-bind a buffer of the new sol_accum_floats() afterwards. */

/* Multi-GPU behind the ABI ("the handle owns the communicators, the read gathers across GPUs"). One
 * process per GPU, each with its own SolScene of the same description:
 *   rank 0:  sol_comm_unique_id(id)  -> ship the 128 bytes to the other ranks by any means (file, socket, MPI, ..)
 *   all:     sol_comm_init(scene, rank, world, id)   collective: ncclCommInitRank (RCCL); also sets the tile partition
 *   all:     sol_render(...)         every rank renders the 8x8 tiles it owns
 *   all:     sol_gather(scene, image_dev)            collective, asynchronous on the scene's stream: the compact fp32
 *            accumulators travel to rank 0 (grouped ncclSend / ncclRecv: every shard rides its own xGMI link into the
 *            root, no ring) and are un-permuted there into `image_dev` (device, W*H*3 floats, row 0 = top; NULL = the
 *            scene's own image buffer, see sol_resolve_image); ranks != 0 ignore image_dev.
 *   rank 0:  sol_read_image(scene, host_rgb_sum)     blocks; copies the gathered image out (or use the device pointer)
 *   all:     sol_comm_destroy(scene)  (also done by sol_scene_destroy)
 * RCCL (librccl.so) is loaded on the first sol_comm_* call; a single-GPU process never needs it. */
#define SOL_UNIQUE_ID_BYTES 128
int sol_comm_unique_id(uint8_t id[SOL_UNIQUE_ID_BYTES]);
int sol_comm_init(SolScene* scene, int rank, int world, const uint8_t id[SOL_UNIQUE_ID_BYTES]);
int sol_comm_destroy(SolScene* scene);
int sol_gather(SolScene* scene, void* image_dev);
/* Diagnostic for single-GPU boxes: sends the accumulator to this very rank through the communicator (the grouped
 * ncclSend / ncclRecv pair sol_gather uses) and compares the bytes. */
int sol_comm_self_check(SolScene* scene);
/* The same gather for ONE process that drives n GPUs (the way a caller of the reference's ray_trace(), src/lib.rs:93-99, is written): scenes[i] is
 * rank i of n (sol_scene_set_partition(scenes[i], i, n)), each created on its own device. Waits for every rank's renders, copies the compact
 * accumulators device to device into rank 0's gather buffer (peer copies; no communicator, RCCL is not loaded) and un-permutes them into rank 0's
 * image buffer; *image_dev = that buffer (W*H*3 floats, row 0 = top, on scenes[0]'s device, valid until the next gather / read / resolve on
 * scenes[0]); asynchronous on scenes[0]'s stream - the post-processors (sol_tonemap_rgb8, sol_bloom*) of scenes[0] take it as they take
 * sol_resolve_image's, sol_read_image copies it to the host. */
int sol_gather_local(SolScene* const* scenes, int n, void** image_dev);
int sol_read_image(SolScene* scene, float* rgb_sum);
/* Largest n_samples one sol_render call accepts for the current partition (the work counter is 32 bits). */
uint32_t sol_max_samples_per_call(const SolScene* scene);

/* Compact accumulator (device memory, fp32 sums over samples). By default the handle owns it; a caller that
 * wants to hand it to a collective (torch.distributed / RCCL) may bind its own device buffer instead. */
size_t sol_accum_floats(const SolScene* scene);
void* sol_accum_ptr(SolScene* scene);
int sol_scene_bind_accum(SolScene* scene, void* device_ptr, size_t n_floats);
/* Use the caller's hipStream_t for all work of this handle (NULL = the handle's own stream). */
int sol_scene_set_stream(SolScene* scene, void* hip_stream);

/* Zero the accumulator. */
int sol_clear(SolScene* scene);

/* Enqueue samples [first_sample, first_sample + n_samples) of every pixel this rank owns; their colours are
 * ADDED to the accumulator (sums, not means: src/renderer/mod.rs:361-365, src/util/rgb_color.rs:21-35).
 * Asynchronous. Replaces the row tasks of one or more passes (src/renderer/mod.rs:241-291). `seed` keys the
 * counter-based RNG that replaces src/random.rs; the result is a pure function of (scene, seed, pixel,
 * sample index) and independent of the partition. */
int sol_render(SolScene* scene, uint32_t first_sample, uint32_t n_samples, uint64_t seed);
/* Same, with instrumentation counters enabled (slower; fills sol_stats). */
int sol_render_counted(SolScene* scene, uint32_t first_sample, uint32_t n_samples, uint64_t seed);
int sol_sync(SolScene* scene);

/* Blocks; writes the full-image fp32 sums, W*H*3 floats, row 0 = image TOP (row index (H-1-y),
 * src/renderer/mod.rs:261). Pixels owned by other ranks are written as 0 when world > 1. */
int sol_read(SolScene* scene, float* rgb_sum);

/* Auxiliary buffers of the first hit (src/renderer/mod.rs:175-204; consumed by denoising post-processors): samples
 * [first, first + n) of the albedo colour (AlbedoShader on the primary hit, background colour on a miss) and of the shading
 * normal (NormalShader, zero on a miss) are ADDED to two accumulators laid out like the colour accumulator (same partition).
 * Independent of sol_render; sol_clear_aux zeroes them, sol_read_aux blocks and writes W*H*3 floats each, row 0 = top
 * (either pointer may be NULL). */
int sol_render_aux(SolScene* scene, uint32_t first_sample, uint32_t n_samples, uint64_t seed);
int sol_clear_aux(SolScene* scene);
int sol_read_aux(SolScene* scene, float* albedo_sum, float* normal_sum);

/* Rank-0 side of the multi-GPU gather: `gathered` is device memory holding world compact buffers back to
 * back (rank r at offset r * sol_accum_floats()), as produced by an RCCL gather; writes the row-major image
 * (W*H*3 floats, row 0 = top) to device memory `image`. */
int sol_unpermute(SolScene* scene, const void* gathered_dev, int world, void* image_dev);

/* Device-side Nop post-processor: sums -> (/spp, sqrt, clamp, *256) -> RGB8, the arithmetic of
 * src/util/rgb_color.rs:14-35 via src/post/nop.rs:19-34. `image_dev` is W*H*3 floats row-major (device),
 * `rgb8_host` receives W*H*3 bytes. */
int sol_tonemap_rgb8(SolScene* scene, const void* image_dev, uint32_t num_samples, uint8_t* rgb8_host);

/* Un-permutes this scene's own accumulators into its internal row-major image buffer (W*H*3 floats, row 0 = top, device
 * memory owned by the scene, valid until the next sol_read / sol_resolve_image) and returns its device pointer: the input
 * of the device-side post-processors when no gather is involved (single GPU). Asynchronous on the scene's stream. */
int sol_resolve_image(SolScene* scene, void** image_dev);

/* Device-side BloomPostProcessor (src/post/bloom.rs:76-150), f64 arithmetic in the reference's summation order on the
 * fp32 sums of `image_dev` (W*H*3 floats, row-major, device). `threshold` and `max_intensity` are per-sample values
 * (BloomPostProcessor::new's defaults: |(1,1,1)| and f64::MAX); kernel_size_fraction outside [0, 0.5] is SOL_EINVAL with
 * the reference's message.
 *   sol_bloom       = intermediate_post_process: image_dev <- pixel + blurred bright pixels, rounded to fp32, in place;
 *   sol_bloom_rgb8  = post_process: the same sums carried in f64 through to_rgb_color into rgb8_host (W*H*3 bytes).
 *  */
int sol_bloom(SolScene* scene, void* image_dev, uint32_t num_samples, double kernel_size_fraction, double threshold,
              double max_intensity);
int sol_bloom_rgb8(SolScene* scene, const void* image_dev, uint32_t num_samples, double kernel_size_fraction, double threshold,
                   double max_intensity, uint8_t* rgb8_host);
/* create_gaussian_blur_weights (src/util/gaussian.rs:11-25), the weights sol_bloom uses; host-only, for known-answer tests. */
int sol_gaussian_blur_weights(uint32_t kernel_size, double std_dev, double* out);

int sol_stats(const SolScene* scene, SolStats* out);
/* What the paths of the last instrumented render (sol_render_counted, one-path-per-lane kernel) looked like - how hard a workload
 * is: the share of camera rays that hit something and the samples by the number of rays of their path (a path of n rays was
 * scattered n - 1 times: src/renderer/shader.rs:62-125). No reference analogue. */
typedef struct SolPathStats {
  uint32_t size, pad;      /* in: sizeof(SolPathStats)                                                              */
  uint64_t samples;
  uint64_t primary_hits;   /* samples whose camera ray hit a primitive                                              */
  uint64_t path_len[6];    /* samples with 1, 2, 3-4, 5-8, 9-16, 17 or more rays                                    */
} SolPathStats;
int sol_path_stats(const SolScene* scene, SolPathStats* out);

/* Diagnostic, host only (no device needed): builds the 7-wide quantised tree of the world exactly as sol_scene_create does
 * (use_sah = 0: collapsed from the reference's topology, 1: from the 16-bin SAH rebuild, n > 1: from the n-bin rebuild; -1: the tree
 * the GPU builder of SOL_TREE_DEVICE makes - this one needs a device) and verifies its structure with the
 * device's decode arithmetic. Returns SOL_OK with the findings in `out`; the tree is sound iff box_violations ==
 * leaf_mismatches == bad_empty_slots == 0. */
typedef struct SolTreeCheck {
  uint32_t n_wide;           /* wide nodes                                                                       */
  uint32_t n_leaf_refs;      /* primitive references reachable from the root                                      */
  uint32_t n_primitives;     /* primitive references of the reference-shaped tree (the multiset the tree must hold)*/
  uint32_t depth;            /* levels of wide nodes                                                              */
  uint32_t max_children;     /* most children in one node (<= 7)                                                  */
  uint32_t box_violations;   /* children whose decoded box does not contain every padded primitive box below it   */
  uint32_t leaf_mismatches;  /* primitive references missing from / surplus in the tree                           */
  uint32_t bad_empty_slots;  /* empty slots that are not (NONE reference, inverted box)                           */
  double inner_area, leaf_area; /* summed box areas of inner / primitive children (the surface-area cost estimate)  */
  uint32_t n_extra_references; /* device build: references that pre-splitting added (a split triangle has several)     */
  uint32_t n_split_triangles;  /* triangles with more than one reference                                              */
  uint32_t split_uncovered;    /* sample points of split triangles that no reference box of their triangle holds (must be 0) */
  uint32_t reserved;
} SolTreeCheck;
/* SolTreeCheck carries no size field and grew after its first release (the three split counters and `reserved`): sol_world_tree_check
 * writes the FIRST layout only - the fields through leaf_area, SOL_TREE_CHECK_V1_BYTES - so a binding compiled against that header is never
 * overrun; sol_world_tree_check_ex takes the caller's sizeof(SolTreeCheck) and fills every field that fits (the rest of `out` is zeroed). */
#define SOL_TREE_CHECK_V1_BYTES 48
int sol_world_tree_check(const SolSceneDesc* desc, int use_sah, SolTreeCheck* out);
int sol_world_tree_check_ex(const SolSceneDesc* desc, int use_sah, void* out, size_t out_size);
/* Diagnostic, host only: the background blocks (SolSceneInfo::background_blocks) found with the host-built tree `use_sah` (>= 0) names:
 * flags[b] = 1 for block b (row-major, (width + 7) / 8 blocks per row), *n_found their number. flags may be NULL. */
int sol_background_blocks(const SolSceneDesc* desc, int use_sah, uint8_t* flags, size_t n_flags, uint32_t* n_found);

/* Diagnostic: traces the single path (pixel x, y counted from the image top; sample index) and writes 12 floats per ray
 * (origin xyz, direction xyz, hit t, hit reference bits, dfs index bits, depth, 0, 0), closed by a row holding the sample's
 * colour in its first three floats and -1 in the fourth. For comparing a path bounce by bounce with the oracle. */
int sol_debug_path(SolScene* scene, uint32_t x, uint32_t y, uint32_t sample, uint64_t seed, float* rows, uint32_t max_rows);

/* Measurement: with timing enabled every sol_render brackets its render kernel with HIP events on the stream it is
 * launched on; sol_last_kernel_ms blocks on the last one and returns its duration (and the grid it was launched with).
 * No reference analogue (the reference reports only passes/s, src/renderer/mod.rs:367-373). */
int sol_kernel_timing(SolScene* scene, int enable);
int sol_last_kernel_ms(SolScene* scene, float* ms, uint32_t* grid_blocks);

/* Function-level evaluation of the device code on n rows of host floats (in_stride / out_stride floats per row), for
 * pinning the fp32 arithmetic contract bit for bit (tests/test_gpu_functions.py). fn: 0 arithmetic, 1 elementary
 * functions, 2 RNG, 3 vector ops + Onb::new, 4 Sphere::hit, 5 Quad::hit, 6 Triangle::hit, 7 Aabb::hit, 8 sampling
 * (row layouts: solstrale-rust_amd/csrc/sol_aux.hip, sol_eval_kernel; floats read / written per row, fn 0..8: 3/7, 3/5, 5/2, 7/18, 13/2,
 * 24/4, 17/4, 12/2, 4/7 - narrower strides and unknown functions are SOL_EINVAL). No reference analogue. */
int sol_eval(int device, uint32_t fn, const float* in, uint32_t n, uint32_t in_stride, float* out, uint32_t out_stride);

/* Sizes of the device records, for the algorithmic-bytes formula: node, sphere, quad, triangle
 * intersect records, shading record, material record (bytes). */
int sol_record_sizes(uint32_t out[6]);

const char* sol_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SOLSTRALE_HIP_H */
