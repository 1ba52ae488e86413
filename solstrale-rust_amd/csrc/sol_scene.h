// sol_scene.h -- the handle behind the C ABI (include/solstrale_hip.h) and what the translation units of libsolstrale_hip.so
// share: error reporting, the developer overrides (environment variables, parsed in ONE place), the device memory of a scene.
//   sol_api.cpp     handle life cycle, options, partition, accumulators, read-back, statistics
//   sol_create.cpp  validation of the flattened scene, conversion to the fp32 device layout (sol_types.h), world tree, upload, probes
//   sol_launch.cpp  sol_render* / auxiliary planes / debug hooks: launches of the kernels in sol_render.hip
//   sol_post.cpp    un-permute, Nop tone-map, bloom (kernels in sol_aux.hip)
//   sol_comm.cpp    RCCL communicator and the gather to rank 0
// There is NO CPU fallback: without a HIP device every compute entry point fails with SOL_EDEVICE.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/solstrale_hip.h"
#include "sol_launch.h"
#include "sol_types.h"

// Sets the thread's error string (sol_last_error) and returns `code`.
int sol_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return sol_fail(SOL_EDEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

template <typename T>
int sol_upload(const std::vector<T>& host, T** dev) {
  *dev = nullptr;
  size_t bytes = std::max<size_t>(host.size() * sizeof(T), 64);  // never a null device pointer
  HIP_TRY(hipMalloc((void**)dev, bytes));
  HIP_TRY(hipMemset(*dev, 0, bytes));
  if (!host.empty()) HIP_TRY(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  return SOL_OK;
}

#define SOL_MAX_ITEMS 0xFF000000ull

// Developer overrides: environment variables for experiments and A/B runs, parsed by sol_dev_overrides() - the
// only getenv site of the library - once per sol_scene_create / sol_world_tree_check; nothing on the launch path reads the environment.
struct SolDevOverrides {
  int kernel_version = 0;        // SOL_KERNEL=v1|v2|v3 (v2 / v3 exist in -DSOL_AB_KERNELS builds only)
  std::string bvh;               // SOL_BVH=device|host|ref|sah|sah8|sah16|sah64 ("" = SolCreateOptions.world_tree)
  bool greedy_collapse = false;  // SOL_COLLAPSE=greedy
  bool octant_slots = false;     // SOL_SLOTS=octant
  double node_cost = 2.5;        // SOL_NODE_COST
  std::vector<int> sah_bins;     // SOL_SAH_LIST=4,12,.. (empty: 8, 16, 64)
  int ploc_radius = 0;           // SOL_PLOC_R (0: the builder's default)
  int split_percent = -1;        // SOL_SPLIT: pre-split budget of the device build in percent of the primitive count (0 off; -1: not set)
  int split_slack = -1;          // SOL_SPLIT_SLACK: levels below the one-primitive cells a plane must lie to be worth a split (-1: not set)
  int background_blocks = -1;    // SOL_BACKGROUND_BLOCKS: 0 = do not look for background blocks (-1: not set)
  int split_keep = -1;           // SOL_SPLIT_KEEP: keep the splits when the summed box area falls below this percentage (-1: not set)
  int reinsert_rounds = -1;      // SOL_REINSERT: reinsertion rounds of the device build (0 off; -1: not set)
  int reinsert_stride = 0;       // SOL_REINSERT_STRIDE: every n-th node searches per round (0: not set)
  int order_mode = 2;            // SOL_ORDER: 0 no work-order probe, 1 heavy blocks first only, 2 + cost classes
  int switch_below = -1;         // SOL_SWITCH (-1: default)
  int max_bpc = -1;              // SOL_MAX_BPC
  int fine_tail = -2;            // SOL_FINE_TAIL (-2: not set)
  int probe_radii = -1;          // SOL_PROBE_RADII (AUTO device build: 1 = emit and probe both clustering radii, 0 = never, -1 = by the collapse costs)
  int pool_swap_min = 0;         // SOL_POOL_SWAP (pool kernel: RenderParams::swap_min; 0: the default)
  int pool_slots = 0, wf_slots = 0, wf_min_items = -1;  // SOL_POOL_SLOTS / SOL_WF_SLOTS / SOL_WF_MIN_ITEMS (v2 / v3)
  std::string rccl_lib;          // SOL_RCCL_LIB: the communication library to dlopen instead of librccl.so.1 (tests)
  bool verbose = false;          // SOL_VERBOSE
};
SolDevOverrides sol_dev_overrides();

// Device memory that depends on the choice of the world tree (sol_scene_create probes several candidates): the 7-wide tree, the
// primitive arrays in that tree's leaf order and every table holding references into them.
struct DevTree {
  DWide* wides = nullptr; uint32_t* leaf_refs = nullptr; DTri* tris = nullptr; DTriShade* tri_shade = nullptr; DQuad* quads = nullptr;
  DSphere* spheres = nullptr; DNode* nodes = nullptr; DMedium* mediums = nullptr; uint32_t* lights = nullptr;
  uint32_t emin = 1, depth = 0, root = 0, light0 = 0;
  uint32_t n_wide = 0, packed_depth = 0;  // wide nodes; stack bound with one-dword node groups (the pool kernel, sol_pool.hip)
  std::vector<uint32_t> old_tri, old_sphere, old_quad;  // device index -> index in the caller's arrays
  void release() {
    void* p[] = {wides, leaf_refs, tris, tri_shade, quads, spheres, nodes, mediums, lights};
    for (void* q : p) if (q) hipFree(q);
    *this = DevTree{};
  }
};

struct SolScene {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  DevScene S{};
  DevScene* dscene = nullptr; DevScene S_uploaded{}; bool dscene_valid = false;  // device copy of S (the v1 kernel reads it through a pointer)
  // owned device buffers
  std::vector<uint32_t> old_index[3];  // triangles / spheres / quads: device index -> index in the caller's SolSceneDesc arrays
  std::string tree_name;               // which world tree the handle walks ("ref", "sah8", .., "device")
  std::string tree_note;               // why it is not the one asked for (AUTO: the device build failed), else empty
  uint32_t split_references = 0, split_triangles = 0;  // device build: what triangle pre-splitting added
  float split_area_ratio = 1.f;
  uint32_t reinsertion_moves = 0; float reinsertion_area_ratio = 1.f;  // device build: sub-trees moved; summed inner-node area after / before
  uint32_t* leaf_refs = nullptr;
  DNode* nodes = nullptr; DWide* wides = nullptr; DTri* tris = nullptr; DTriShade* tri_shade = nullptr; DQuad* quads = nullptr;
  DSphere* spheres = nullptr; DMedium* mediums = nullptr; DMat* mats = nullptr; DTex* texs = nullptr;
  uint8_t* texels = nullptr; uint32_t* lights = nullptr; float* env = nullptr; DTri* light_tri = nullptr;
  float* acc_own = nullptr; float* acc = nullptr; size_t acc_floats = 0;
  float* aux[2] = {nullptr, nullptr}; size_t aux_floats = 0;
  std::vector<uint32_t> block_cost;  // per 8x8 block (global index): rays of its longest item in the cost probe; empty: no ordering
  std::vector<uint32_t> block_work;  // per 8x8 block (global index): its rays in the cost probe (balanced partition)
  bool balanced = false;             // SOL_OPT_BALANCED_PARTITION
  uint32_t partition_table = 0, partition_crc = 0;  // the partition in force: 1 = the balanced table (0: b % world), checksum of block -> slot
  std::vector<uint32_t> local_blocks;  // balanced partition: image block of every local block of this rank (empty: b = lb * world + rank)
  uint32_t* block_of_local_dev = nullptr; size_t block_of_local_cap = 0;
  uint32_t* slot_of_block = nullptr;   // balanced partition (device, all blocks): owner * blocks-per-buffer + local block; null: modulo
  uint32_t* order_dev = nullptr; size_t order_cap = 0;  // DevScene::block_order of the current partition  // albedo / normal accumulators (sol_render_aux), same layout as acc
  float* partial = nullptr; size_t partial_floats = 0;
  int fine_tail = -1;                // SOL_OPT_FINE_TAIL / SOL_FINE_TAIL: quarters of a whole item per resident lane that the end of a launch hands
                                     // out sample by sample; 0: none; -1: by the creation probe's node visits per sample (fine_tail_auto)
  int fine_tail_auto = 0;
  float* image = nullptr;  // W*H*3 scratch for sol_read / sol_resolve_image
  uint8_t* rgb8 = nullptr;
  double* bloom_a = nullptr; double* bloom_b = nullptr; double* bloom_w = nullptr; size_t bloom_w_cap = 0;  // sol_bloom scratch
  uint32_t* work = nullptr; uint32_t* spill = nullptr; size_t spill_words = 0;
  DevCounters* counters = nullptr;
  SolStats stats{};
  SolPathStats path_stats{};
  bool has_medium = false;
  bool strict_triangles = false;  // the scene has needle triangles: the STRICT kernel variants (sol_render.hip)
  uint32_t tree_depth = 0;
  uint32_t n_wide = 0, packed_depth = 0;  // (DevTree) wide nodes of the world tree; stack bound of a search with one-dword node groups
  uint32_t pool_swap_min = 0;             // SOL_POOL_SWAP (pool kernel, RenderParams::swap_min; 0: the default)
  int rank = 0, world = 1;
  uint32_t blocks_x = 0, blocks_y = 0, n_local_blocks = 0;
  int n_cu = 0;
  int kernel_version = 0;          // 0 auto; SOL_KERNEL=v1|v2|v3 forces one (A/B comparisons)
  void* pool = nullptr; size_t pool_bytes = 0;  // path-slot pool of the wavefront kernels
  uint32_t pool_slots_override = 0;  // SOL_POOL_SLOTS (v2: slots per wave)
  uint32_t switch_below = 0;         // SOL_SWITCH (v1, RenderParams::switch_below)
  uint32_t* queue = nullptr; size_t queue_slots = 0;  // v3 ray queue
  void* wf_ctr = nullptr; uint32_t* wf_ctr_host = nullptr;
  uint32_t wf_slots = 4u << 20;       // SOL_WF_SLOTS: pool size of the two-kernel wavefront
  uint32_t wf_min_items = 2u << 20;   // SOL_WF_MIN_ITEMS: jobs below this use the single-launch kernel
  uint32_t last_rounds = 0; int last_version = 0;
  double build_times[4] = {0., 0., 0., 0.};  // sol_scene_build_times
  bool order_enabled = true;         // SOL_OPT_WORK_ORDER
  // background blocks (solstrale_hip.h SolSceneInfo::background_blocks): per 8x8 block (global index) 1 = proved to see only the
  // background; n_background_local: how many of them this rank owns - the LAST so many entries of the work order
  std::vector<uint8_t> background_block;
  uint32_t n_background = 0, n_background_local = 0, background_pixels = 0;
  bool background_enabled = true;    // SOL_OPT_BACKGROUND_BLOCKS
  bool background_in_counted = false;  // (value 2) counted renders skip them too: the counters of exactly what a plain render does
  int order_mode = 2;                // (SOL_ORDER) 1: heavy blocks first only; 2: + cost classes within a chunk
  int max_bpc = 0;                   // SOL_OPT_MAX_BLOCKS_PER_CU (0 = what the occupancy query allows)
  // multi-GPU (sol_comm_init): RCCL communicator of the tile partition and rank 0's receive buffer
  void* comm = nullptr; float* gathered = nullptr; size_t gathered_floats = 0;
  bool timing = false;  // sol_kernel_timing: HIP events around the render kernel on its own stream
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  uint32_t timed_launches = 0, last_grid = 0;
};

int sol_rebuild_order(SolScene* s);
int sol_set_partition(SolScene* s, int rank, int world);
int sol_render_impl(SolScene* s, uint32_t first, uint32_t n, uint64_t seed, bool count);
inline int sol_render_probe(SolScene* s) { return sol_render_impl(s, 0, SOL_CHUNK, 0x50B3ull, true); }
