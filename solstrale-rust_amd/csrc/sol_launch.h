// sol_launch.h -- host-callable launch wrappers of sol_kernels.hip (internal to libsolstrale_hip.so).
#pragma once
#include <hip/hip_runtime.h>

#include "sol_types.h"

hipError_t sol_launch_render(const DevScene& S, const RenderParams& P, float* acc, float* partial, uint32_t* work,
                             uint32_t* spill, DevCounters* cnt, uint32_t grid, bool count, bool medium, hipStream_t stream);
int sol_render_blocks_per_cu(bool count, bool medium);
hipError_t sol_launch_resolve(float* acc, const float* partial, uint32_t n_floats, uint32_t n_chunks, hipStream_t stream);
hipError_t sol_launch_unpermute(const float* gathered, float* image, uint32_t width, uint32_t height, uint32_t blocks_x,
                                uint32_t world, uint32_t only_rank, size_t stride, hipStream_t stream);
hipError_t sol_launch_tonemap(const float* image, uint8_t* rgb, uint32_t n, uint32_t spp, hipStream_t stream);
hipError_t sol_launch_eval(uint32_t fn, const float* in, uint32_t n, uint32_t in_stride, float* out, uint32_t out_stride,
                           hipStream_t stream);
