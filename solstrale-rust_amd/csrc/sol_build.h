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

// What the device build hands back (the fields WideLayout of sol_tree.h has): the 64-byte nodes, the listed references of
// mixed nodes, and per primitive array (triangles / spheres / quads) the map caller's index -> device index.
struct SolDeviceTree {
  std::vector<DWide> nodes;
  std::vector<uint32_t> leaf_refs;
  std::vector<uint32_t> new_of_old[3];
  uint32_t depth = 0;   // levels of wide nodes
  uint32_t rounds = 0;  // clustering rounds
};

// Builds the 7-wide tree over `prims` (host memory, n >= 1, every reference at most once) on the current HIP device.
// root_box: box of all primitives; pad: the scene's fp32 box pad; emin: smallest biased exponent of a node scale
// (WideBuilder::exponent_min); counts: sizes of the triangle / sphere / quad arrays; ploc_radius: neighbours searched to each side in a
// clustering round (0: the default, 16). False + message on failure.
bool sol_build_world_tree_device(const SolBuildPrim* prims, uint32_t n, const float root_box[6], float pad, uint32_t emin, const uint32_t counts[3],
                                 int ploc_radius, hipStream_t stream, SolDeviceTree& out, std::string& err);
