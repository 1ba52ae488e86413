// sol_build.h -- interface of the GPU builder of the world tree (sol_build.hip), internal to libsolstrale_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "sol_types.h"

// One primitive of the world: its reference (SOL_MAKE_REF) and its padded fp32 box (sol_tree.h, cast_box).
struct SolBuildPrim {
  float box[6];
  uint32_t ref;
  uint32_t pad;
};

// Triangle pre-splitting ahead of the Morton sort (sol_build.hip, step 0): how many extra references the build may make, as a
// fraction of the primitive count (0: none), and how far below the level of one-primitive cells a spatial-median plane must lie
// to be worth a split (`level_slack` = 3: planes that separate groups of about 2^3 primitives and more).
struct SolSplitOptions {
  float budget = 0.3f;
  int level_slack = 3;
  // The splits are kept only when they shrink the summed surface area of the primitives' boxes - what the number of primitive tests
  // of a random ray is proportional to - to below this fraction: a mesh of uniform small triangles gains nothing from the few
  // triangles that straddle a plane (C3's regular grids: 0.98, C5: 0.99 - node visits per ray go UP with the extra references),
  // walls, rails and rods beside small ornament do (the heterogeneous atrium: 0.45). 1: always keep them.
  float max_area_ratio = 0.85f;
  bool want_boxes = false;  // hand the references' boxes back (sol_world_tree_check)
  // Reinsertion rounds after the clustering (sol_build.hip, step 2b): every `stride`-th node looks for a better place in each round
  // (stride 1: all of them).
  float node_cost = 2.5f;  // collapse: what a wide-node visit costs against a primitive test
  int reinsertion_rounds = 8;
  int reinsertion_stride = 1;
  bool verbose = false;
};

// What the device build hands back (the fields WideLayout of sol_tree.h has): the 64-byte nodes, the listed references of
// mixed nodes, and per primitive array (triangles / spheres / quads) the map caller's index -> device index. With pre-splitting
// the triangle array is EXPANDED: reference e < counts[0] is (the first part of) triangle e, reference e >= counts[0] is a further
// part of triangle extra_of[e - counts[0]]; new_of_old[0] then has counts[0] + extra_of.size() entries.
struct SolDeviceTree {
  std::vector<DWide> nodes;
  std::vector<uint32_t> leaf_refs;
  std::vector<uint32_t> new_of_old[3];
  std::vector<uint32_t> extra_of;    // triangle of every extra reference
  std::vector<float> ref_box;        // (want_boxes) 6 floats per triangle reference (expanded index): its padded box; empty box = not in the world
  uint32_t depth = 0;   // levels of wide nodes
  uint32_t rounds = 0;  // clustering rounds
  uint32_t split_triangles = 0;  // triangles that were split
  float split_area_ratio = 1.f;  // summed box area of the references after / before pre-splitting (also when the splits were not kept)
  uint32_t reinsertion_moves = 0;            // sub-trees moved by the reinsertion rounds
  double area_before = 0., area_after = 0.;  // summed surface area of the binary tree's inner nodes before / after them
  float collapse_cost = 0.f;                 // the collapse's surface-area cost of the whole tree (C(root, 1): node cost x node areas + primitive areas)
};

// Builds the 7-wide tree over `prims` (host memory, n >= 1, every reference at most once) on the current HIP device.
// root_box: box of all primitives; pad: the scene's fp32 box pad; emin: smallest biased exponent of a node scale
// (WideBuilder::exponent_min); counts: sizes of the triangle / sphere / quad arrays; tris: the triangle records (counts[0] of them; the
// vertices pre-splitting clips, may be null: no splitting); ploc_radius: neighbours searched to each side in a clustering round
// (0: the default, 16). False + message on failure.
// The same build in two halves (a caller with several candidates compares SolDeviceTree::collapse_cost before it pays for an emission):
// prepare fills the split / reinsertion / cost fields of `out` and hands back a handle; emit fills the rest; release frees the handle.
struct SolDeviceBuild;
bool sol_build_world_tree_prepare(const SolBuildPrim* prims, uint32_t n, const float root_box[6], float pad, uint32_t emin, const uint32_t counts[3],
                                  const DTri* tris, const SolSplitOptions& split, int ploc_radius, hipStream_t stream, SolDeviceTree& out, SolDeviceBuild** handle,
                                  std::string& err);
bool sol_build_world_tree_emit(SolDeviceBuild* handle, SolDeviceTree& out, std::string& err);
void sol_build_world_tree_release(SolDeviceBuild* handle);
bool sol_build_world_tree_device(const SolBuildPrim* prims, uint32_t n, const float root_box[6], float pad, uint32_t emin, const uint32_t counts[3],
                                 const DTri* tris, const SolSplitOptions& split, int ploc_radius, hipStream_t stream, SolDeviceTree& out, std::string& err);
