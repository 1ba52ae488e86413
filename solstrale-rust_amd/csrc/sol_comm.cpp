// sol_comm.cpp -- multi-GPU behind the ABI: RCCL communicator + gather of the tile accumulators to rank 0.
#include <dlfcn.h>

#include <mutex>
#include <string>
#include <vector>

#include "sol_scene.h"

// RCCL is loaded on first use (dlopen), so that a single-GPU process has no dependency on it; when the host process has
// already loaded an RCCL (e.g. torch's), the loader hands back that one (same soname). The handful of types the eight entry
// points need are declared here (the library builds on a box without RCCL headers); where the header exists, it is included and
// the declarations must agree with it.
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define SOL_HAVE_RCCL_HEADER 1
#endif
#endif
#ifndef SOL_HAVE_RCCL_HEADER
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;  // (any other value: an error, named by ncclGetErrorString)
typedef enum { ncclFloat = 7 } ncclDataType_t;
#endif
static_assert(sizeof(ncclUniqueId) == SOL_UNIQUE_ID_BYTES && (int)ncclSuccess == 0 && (int)ncclFloat == 7, "RCCL declarations");
namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const std::string forced = sol_dev_overrides().rccl_lib;  // SOL_RCCL_LIB: this library and no other (tests)
    std::string why;
    auto open = [&](const char* name) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!r.lib) { const char* e = dlerror(); why = e ? e : "?"; }  // (dlerror() clears the message: read it once, right after the failure)
      return r.lib != nullptr;
    };
    if (!forced.empty()) open(forced.c_str());
    else
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
        if (open(name)) break;
    if (!r.lib) { r.error = "cannot load librccl.so: " + why; return; }
    auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p && r.error.empty()) r.error = std::string("librccl.so lacks ") + n; return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  });
  return r;
}
#define RCCL_TRY(expr)                                                                                   \
  do {                                                                                                   \
    ncclResult_t r_ = (expr);                                                                            \
    if (r_ != ncclSuccess) return sol_fail(SOL_EDEVICE, "%s: %s", #expr, rccl().GetErrorString(r_));         \
  } while (0)
}  // namespace

extern "C" {

int sol_comm_unique_id(uint8_t id[SOL_UNIQUE_ID_BYTES]) {
  if (!id) return sol_fail(SOL_EINVAL, "null argument");
  Rccl& R = rccl();
  if (!R.error.empty()) return sol_fail(SOL_EDEVICE, "%s", R.error.c_str());
  ncclUniqueId u;
  RCCL_TRY(R.GetUniqueId(&u));
  std::memcpy(id, &u, sizeof u);
  return SOL_OK;
}

int sol_comm_destroy(SolScene* s) {
  if (!s) return sol_fail(SOL_EINVAL, "null scene");
  if (s->comm) {
    hipSetDevice(s->device);
    hipStreamSynchronize(s->stream);
    rccl().CommDestroy((ncclComm_t)s->comm);
    s->comm = nullptr;
  }
  if (s->gathered) { hipFree(s->gathered); s->gathered = nullptr; s->gathered_floats = 0; }
  return SOL_OK;
}

int sol_comm_init(SolScene* s, int rank, int world, const uint8_t id[SOL_UNIQUE_ID_BYTES]) {
  if (!s || !id) return sol_fail(SOL_EINVAL, "null argument");
  if (world < 1 || rank < 0 || rank >= world) return sol_fail(SOL_EINVAL, "bad rank %d of %d", rank, world);
  Rccl& R = rccl();
  if (!R.error.empty()) return sol_fail(SOL_EDEVICE, "%s", R.error.c_str());
  int rc = sol_comm_destroy(s);
  if (rc || (rc = sol_scene_set_partition(s, rank, world))) return rc;
  HIP_TRY(hipSetDevice(s->device));
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  ncclComm_t comm = nullptr;
  RCCL_TRY(R.CommInitRank(&comm, world, u, rank));
  s->comm = comm;
  return SOL_OK;
}

// One collective per emitted image: every rank's compact accumulator (equal sizes, sol_accum_floats) goes to
// rank 0 in ONE group of point-to-point transfers - each shard rides its own xGMI link into the root - and rank 0 un-permutes
// the `world` compact buffers into the row-major image. Without a communicator (world 1) it is sol_resolve_image into image_dev.
int sol_gather(SolScene* s, void* image_dev) {
  if (!s) return sol_fail(SOL_EINVAL, "null scene");
  HIP_TRY(hipSetDevice(s->device));
  if (s->world > 1 && !s->comm) return sol_fail(SOL_EINVAL, "the scene is partitioned %d-way but has no communicator: call sol_comm_init", s->world);
  float* image = image_dev ? (float*)image_dev : s->image;
  if (s->world == 1) {
    HIP_TRY(sol_launch_unpermute(s->acc, image, s->S.width, s->S.height, s->blocks_x, 1u, 0u, s->acc_floats, s->slot_of_block, s->stream));
    return SOL_OK;
  }
  Rccl& R = rccl();
  ncclComm_t comm = (ncclComm_t)s->comm;
  const size_t n = s->acc_floats;
  if (s->rank == 0) {
    if (s->gathered_floats != n * (size_t)s->world) {
      HIP_TRY(hipStreamSynchronize(s->stream));
      if (s->gathered) hipFree(s->gathered);
      s->gathered = nullptr; s->gathered_floats = 0;
      HIP_TRY(hipMalloc((void**)&s->gathered, n * (size_t)s->world * sizeof(float)));
      s->gathered_floats = n * (size_t)s->world;
    }
    HIP_TRY(hipMemcpyAsync(s->gathered, s->acc, n * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    RCCL_TRY(R.GroupStart());
    for (int r = 1; r < s->world; ++r) {
      ncclResult_t e = R.Recv(s->gathered + (size_t)r * n, n, ncclFloat, r, comm, s->stream);
      if (e != ncclSuccess) { R.GroupEnd(); return sol_fail(SOL_EDEVICE, "ncclRecv from rank %d: %s", r, R.GetErrorString(e)); }
    }
    RCCL_TRY(R.GroupEnd());
    HIP_TRY(sol_launch_unpermute(s->gathered, image, s->S.width, s->S.height, s->blocks_x, (uint32_t)s->world, 0xFFFFFFFFu, n, s->slot_of_block, s->stream));
  } else {
    RCCL_TRY(R.Send(s->acc, n, ncclFloat, 0, comm, s->stream));
  }
  return SOL_OK;
}

// The gather of ONE process that drives several GPUs (the reference's ray_trace() is called from one process: src/lib.rs:93-99): scenes[i] is
// rank i of n, each on its own device with its own stream; their compact accumulators are copied device to device into rank 0's gather buffer
// (hipMemcpyPeerAsync: one xGMI hop each, no communicator, no RCCL) and un-permuted there, exactly as sol_gather does with what it received.
int sol_gather_local(SolScene* const* scenes, int n, void** image_dev) {
  if (!scenes || n < 1 || !image_dev) return sol_fail(SOL_EINVAL, "bad argument");
  for (int i = 0; i < n; ++i)
    if (!scenes[i]) return sol_fail(SOL_EINVAL, "scene %d is null", i);
  SolScene* root = scenes[0];
  const size_t floats = root->acc_floats;
  for (int i = 0; i < n; ++i) {
    const SolScene* s = scenes[i];
    if (s->world != n || s->rank != i) return sol_fail(SOL_EINVAL, "scene %d is rank %d of %d, expected rank %d of %d (sol_scene_set_partition)", i, s->rank, s->world, i, n);
    if (s->S.width != root->S.width || s->S.height != root->S.height || s->acc_floats != floats || s->partition_crc != root->partition_crc)
      return sol_fail(SOL_EINVAL, "scene %d does not share rank 0's frame and partition", i);
    for (int j = 0; j < i; ++j)
      if (scenes[j] == s) return sol_fail(SOL_EINVAL, "scene %d is scene %d again", i, j);
  }
  for (int i = 0; i < n; ++i) {  // (every rank's sums are complete before they are copied)
    HIP_TRY(hipSetDevice(scenes[i]->device));
    HIP_TRY(hipStreamSynchronize(scenes[i]->stream));
  }
  HIP_TRY(hipSetDevice(root->device));
  if (n == 1) {
    HIP_TRY(sol_launch_unpermute(root->acc, root->image, root->S.width, root->S.height, root->blocks_x, 1u, 0u, floats, root->slot_of_block, root->stream));
    *image_dev = root->image;
    return SOL_OK;
  }
  if (root->gathered_floats != floats * (size_t)n) {
    HIP_TRY(hipStreamSynchronize(root->stream));
    if (root->gathered) hipFree(root->gathered);
    root->gathered = nullptr; root->gathered_floats = 0;
    HIP_TRY(hipMalloc((void**)&root->gathered, floats * (size_t)n * sizeof(float)));
    root->gathered_floats = floats * (size_t)n;
  }
  for (int i = 0; i < n; ++i)
    HIP_TRY(hipMemcpyPeerAsync(root->gathered + (size_t)i * floats, root->device, scenes[i]->acc, scenes[i]->device, floats * sizeof(float), root->stream));
  HIP_TRY(sol_launch_unpermute(root->gathered, root->image, root->S.width, root->S.height, root->blocks_x, (uint32_t)n, 0xFFFFFFFFu, floats, root->slot_of_block, root->stream));
  *image_dev = root->image;
  return SOL_OK;
}

// Diagnostic for boxes with ONE GPU (where no second rank can exist): moves this rank's accumulator to itself through the
// communicator - grouped ncclSend + ncclRecv with peer = own rank, the same calls sol_gather issues - and compares the bytes.
int sol_comm_self_check(SolScene* s) {
  if (!s) return sol_fail(SOL_EINVAL, "null scene");
  if (!s->comm) return sol_fail(SOL_EINVAL, "no communicator: call sol_comm_init");
  HIP_TRY(hipSetDevice(s->device));
  Rccl& R = rccl();
  const size_t n = s->acc_floats;
  float* tmp = nullptr;
  HIP_TRY(hipMalloc((void**)&tmp, n * sizeof(float)));
  hipError_t e = hipMemsetAsync(tmp, 0xFF, n * sizeof(float), s->stream);
  ncclResult_t r = ncclSuccess;
  if (e == hipSuccess) {
    r = R.GroupStart();
    if (r == ncclSuccess) r = R.Send(s->acc, n, ncclFloat, s->rank, (ncclComm_t)s->comm, s->stream);
    if (r == ncclSuccess) r = R.Recv(tmp, n, ncclFloat, s->rank, (ncclComm_t)s->comm, s->stream);
    ncclResult_t r2 = R.GroupEnd();
    if (r == ncclSuccess) r = r2;
  }
  std::vector<float> a(n), b(n);
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(a.data(), s->acc, n * sizeof(float), hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(b.data(), tmp, n * sizeof(float), hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  hipFree(tmp);
  if (r != ncclSuccess) return sol_fail(SOL_EDEVICE, "RCCL self transfer: %s", R.GetErrorString(r));
  if (e != hipSuccess) return sol_fail(SOL_EDEVICE, "self check: %s", hipGetErrorString(e));
  if (std::memcmp(a.data(), b.data(), n * sizeof(float)) != 0) return sol_fail(SOL_EDEVICE, "RCCL self transfer returned different bytes");
  return SOL_OK;
}

}  // extern "C"
