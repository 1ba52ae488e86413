// sol_launch.h -- host-callable launch wrappers of sol_render.hip / sol_aux.hip (internal to libsolstrale_hip.so).
#pragma once
#include <hip/hip_runtime.h>

#include "sol_types.h"

// version 1: one path per lane (the product kernel); versions 2 / 3 (-DSOL_AB_KERNELS builds): the wavefront variants of sol_wavefront.hip
hipError_t sol_launch_render(int version, const DevScene& S, const DevScene* dS, const RenderParams& P, float* acc, float* partial, uint32_t* work,
                             uint32_t* spill, void* pool, DevCounters* cnt, uint32_t grid, bool count, bool medium, bool may_spill,
                             hipStream_t stream);
int sol_render_blocks_per_cu(int version, bool count, bool medium, bool strict);
hipError_t sol_launch_stage_resolve(const DevScene* dS, const RenderParams& P, float* partial, hipStream_t stream);
hipError_t sol_launch_fill_background(const DevScene* dS, const RenderParams& P, float* partial, hipStream_t stream);
hipError_t sol_launch_debug_path(const DevScene& S, const RenderParams& P, uint32_t px, uint32_t py, uint32_t s, uint32_t* spill,
                                 float* out, uint32_t max_rows, bool medium, hipStream_t stream);
// ---- sol_pool.hip: version 4, the pool kernel (a second path context per lane in LDS, handed out wave-wide; sample-granular work items) ----
hipError_t sol_launch_pool4(const DevScene& S, const DevScene* dS, const RenderParams& P, float* partial, uint32_t* work, uint32_t* spill, uint32_t grid,
                            bool medium, bool may_spill, DevCounters* cnt, hipStream_t stream);
int sol_pool4_blocks_per_cu(bool medium, bool strict);
int sol_pool4_lds_stack_depth();
// ---- sol_wavefront.hip (-DSOL_AB_KERNELS builds only) ----
// version 2: wave-private wavefront over a pool of path slots
hipError_t sol_launch_pool(const DevScene& S, const RenderParams& P, float* acc, float* partial, uint32_t* work, uint32_t* spill, void* pool,
                           DevCounters* cnt, uint32_t grid, bool count, bool medium, hipStream_t stream);
int sol_pool_blocks_per_cu(bool count, bool medium);
size_t sol_pool_bytes_per_wave(uint32_t slots);
// version 3: two-kernel wavefront (one shade + one trace launch per round)
hipError_t sol_launch_wf_shade(const DevScene& S, const RenderParams& P, float* acc, float* partial, void* ctr, void* rec,
                               void* reservoir, DevCounters* cnt, bool count, hipStream_t stream);
hipError_t sol_launch_wf_trace(const DevScene& S, const RenderParams& P, void* ctr, void* rec, uint32_t* spill, DevCounters* cnt,
                               uint32_t grid, bool count, bool medium, hipStream_t stream);
int sol_wf_trace_blocks_per_cu(bool count, bool medium);
size_t sol_wf_pool_bytes(uint32_t slots);
int sol_wf_lds_stack_depth();
// ---- sol_aux.hip ----
hipError_t sol_launch_resolve(float* acc, const float* partial, uint32_t n_floats, uint32_t n_chunks, hipStream_t stream);
hipError_t sol_launch_unpermute(const float* gathered, float* image, uint32_t width, uint32_t height, uint32_t blocks_x,
                                uint32_t world, uint32_t only_rank, size_t stride, const uint32_t* slot_of_block, hipStream_t stream);
hipError_t sol_launch_tonemap(const float* image, uint8_t* rgb, uint32_t n, uint32_t spp, hipStream_t stream);
// BloomPostProcessor on a W*H*3 fp32 image; a, b: W*H*3 doubles of scratch; rgb == nullptr: result back into `image`
hipError_t sol_launch_bloom(float* image, double* a, double* b, const double* weights, uint32_t k, uint32_t width, uint32_t height,
                            double threshold, double max_intensity, uint8_t* rgb, uint32_t spp, hipStream_t stream);
hipError_t sol_launch_eval(uint32_t fn, const float* in, uint32_t n, uint32_t in_stride, float* out, uint32_t out_stride,
                           hipStream_t stream);
