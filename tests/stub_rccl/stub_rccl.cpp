// stub_rccl.cpp -- TEST INFRASTRUCTURE, never part of the product: a stand-in for librccl.so.1 that lets several ranks share ONE
// GPU. RCCL refuses two ranks on one device, and a box of the test pool has one GPU, so without this the world > 1 branch of
// sol_gather (solstrale-rust_amd/csrc/sol_comm.cpp: the grouped ncclRecv loop of rank 0, the ncclSend of the other ranks, the
// receive offsets) would first run on the day an 8-GPU node is available. The product dlopen()s "librccl.so.1" on its first
// sol_comm_* call; a test puts this directory's _build/ in front of LD_LIBRARY_PATH of the rank processes it spawns (which must
// not have loaded another RCCL, e.g. through torch) - nothing in the product knows about the stub.
//
// It implements exactly the eight entry points the product binds, with RCCL's signatures and stream semantics as far as the
// product relies on them: transfers are ordered after the work already queued on `stream` and complete before later work on it
// (here: by synchronising the stream). Transport: a star of unix stream sockets around rank 0 (whose path the unique id
// carries) - enough for "everyone sends to rank 0"; a transfer between two non-root ranks is refused with an error.
#include <hip/hip_runtime_api.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct ncclComm {
  int rank = 0, nranks = 1;
  std::vector<int> fd;  // rank 0: fd[r] = socket to rank r; rank r != 0: fd[0] = socket to rank 0
  std::string path;     // rank 0 owns the socket file
};
typedef struct ncclComm* ncclComm_t;

int sol_stub_rccl_marker = 1;  // lets a test see that THIS library is the one that was loaded

}  // extern "C"

namespace {

thread_local std::string g_err = "no error";
ncclResult_t fail(ncclResult_t code, const std::string& msg) { g_err = "stub_rccl: " + msg; return code; }

size_t dtype_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}
bool write_all(int fd, const void* p, size_t n) {
  const char* c = (const char*)p;
  while (n) {
    ssize_t w = ::write(fd, c, n);
    if (w < 0) { if (errno == EINTR) continue; return false; }
    c += w; n -= (size_t)w;
  }
  return true;
}
bool read_all(int fd, void* p, size_t n) {
  char* c = (char*)p;
  while (n) {
    ssize_t r = ::read(fd, c, n);
    if (r < 0) { if (errno == EINTR) continue; return false; }
    if (r == 0) return false;  // peer closed
    c += r; n -= (size_t)r;
  }
  return true;
}

struct Op { bool send; void* buf; size_t bytes; int peer; ncclComm_t comm; hipStream_t stream; };
thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_ops;

ncclResult_t peer_fd(ncclComm_t c, int peer, int& fd) {
  if (peer < 0 || peer >= c->nranks) return fail(ncclInvalidArgument, "peer out of range");
  if (c->rank != 0 && peer != 0) return fail(ncclInvalidArgument, "the stub's transport is a star around rank 0: no transfer between ranks " + std::to_string(c->rank) + " and " + std::to_string(peer));
  fd = c->fd[(size_t)peer];
  return ncclSuccess;
}

// Executes a batch the way a group would: transfers to self are matched pairwise (device copy), all sends go out, then the
// receives are served in the order they were posted (each peer has its own socket: no ordering between peers is assumed).
ncclResult_t run_ops(std::vector<Op>& ops) {
  for (const Op& o : ops)
    if (hipStreamSynchronize(o.stream) != hipSuccess) return fail(ncclUnhandledCudaError, "hipStreamSynchronize failed");
  std::vector<char> host;
  std::vector<const Op*> self_send, self_recv;
  for (const Op& o : ops)
    if (o.peer == o.comm->rank) (o.send ? self_send : self_recv).push_back(&o);
  if (self_send.size() != self_recv.size()) return fail(ncclInvalidArgument, "unmatched transfer to self");
  for (size_t k = 0; k < self_send.size(); ++k) {
    if (self_send[k]->bytes != self_recv[k]->bytes) return fail(ncclInvalidArgument, "self transfer: sizes differ");
    if (hipMemcpy(self_recv[k]->buf, self_send[k]->buf, self_send[k]->bytes, hipMemcpyDeviceToDevice) != hipSuccess) return fail(ncclUnhandledCudaError, "device copy failed");
  }
  for (int pass = 0; pass < 2; ++pass)  // sends first: a rank that both sends and receives never waits for its own data
    for (const Op& o : ops) {
      if (o.peer == o.comm->rank || o.send != (pass == 0)) continue;
      int fd = -1;
      ncclResult_t r = peer_fd(o.comm, o.peer, fd);
      if (r != ncclSuccess) return r;
      host.resize(o.bytes);
      uint64_t n = o.bytes;
      if (o.send) {
        if (hipMemcpy(host.data(), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(ncclUnhandledCudaError, "copy to host failed");
        if (!write_all(fd, &n, sizeof n) || !write_all(fd, host.data(), o.bytes)) return fail(ncclSystemError, std::string("socket write: ") + std::strerror(errno));
      } else {
        if (!read_all(fd, &n, sizeof n)) return fail(ncclSystemError, "socket read: peer " + std::to_string(o.peer) + " closed the connection");
        if (n != o.bytes) return fail(ncclInvalidArgument, "receive of " + std::to_string(o.bytes) + " bytes met a send of " + std::to_string(n));
        if (!read_all(fd, host.data(), o.bytes)) return fail(ncclSystemError, "socket read: short message");
        if (hipMemcpy(o.buf, host.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(ncclUnhandledCudaError, "copy to device failed");
      }
    }
  return ncclSuccess;
}

ncclResult_t post(Op o) {
  if (!o.comm) return fail(ncclInvalidArgument, "null communicator");
  if (g_group_depth > 0) { g_ops.push_back(o); return ncclSuccess; }
  std::vector<Op> one{o};
  return run_ops(one);
}

}  // namespace

extern "C" {

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : g_err.c_str(); }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return fail(ncclInvalidArgument, "null id");
  std::memset(id, 0, sizeof *id);
  const char* dir = std::getenv("TMPDIR");
  const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
  std::snprintf(id->internal, sizeof id->internal, "%s/solstub_%d_%llx.sock", dir && *dir ? dir : "/tmp", (int)getpid(), (unsigned long long)now);
  if (std::strlen(id->internal) >= sizeof(((sockaddr_un*)nullptr)->sun_path)) return fail(ncclSystemError, "socket path too long");
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks) return fail(ncclInvalidArgument, "bad rank / nranks");
  id.internal[sizeof id.internal - 1] = 0;
  ncclComm* c = new ncclComm();
  c->rank = rank; c->nranks = nranks;
  c->fd.assign((size_t)nranks, -1);
  sockaddr_un addr{};
  addr.sun_family = AF_UNIX;
  const size_t path_len = std::strlen(id.internal);
  if (path_len == 0 || path_len >= sizeof addr.sun_path) { delete c; return fail(ncclInvalidArgument, "the unique id does not hold a socket path"); }
  std::memcpy(addr.sun_path, id.internal, path_len + 1);
  auto bail = [&](const std::string& m) { for (int f : c->fd) if (f >= 0) ::close(f); delete c; return fail(ncclSystemError, m); };
  if (nranks > 1 && rank == 0) {
    int ls = ::socket(AF_UNIX, SOCK_STREAM, 0);
    if (ls < 0) return bail("socket()");
    ::unlink(addr.sun_path);
    if (::bind(ls, (sockaddr*)&addr, sizeof addr) != 0 || ::listen(ls, nranks) != 0) { ::close(ls); return bail(std::string("bind/listen ") + addr.sun_path + ": " + std::strerror(errno)); }
    c->path = addr.sun_path;
    for (int k = 1; k < nranks; ++k) {
      int fd = ::accept(ls, nullptr, nullptr);
      int32_t r = -1;
      if (fd < 0 || !read_all(fd, &r, sizeof r) || r < 1 || r >= nranks || c->fd[(size_t)r] >= 0) { if (fd >= 0) ::close(fd); ::close(ls); return bail("accept: bad peer"); }
      c->fd[(size_t)r] = fd;
    }
    ::close(ls);
    for (int k = 1; k < nranks; ++k) { int32_t go = 1; if (!write_all(c->fd[(size_t)k], &go, sizeof go)) return bail("handshake"); }
  } else if (nranks > 1) {
    int fd = -1;
    for (int attempt = 0; attempt < 1200; ++attempt) {  // rank 0 may not be listening yet: up to two minutes
      fd = ::socket(AF_UNIX, SOCK_STREAM, 0);
      if (fd >= 0 && ::connect(fd, (sockaddr*)&addr, sizeof addr) == 0) break;
      if (fd >= 0) ::close(fd);
      fd = -1;
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    if (fd < 0) return bail(std::string("cannot connect to rank 0 at ") + addr.sun_path);
    c->fd[0] = fd;
    int32_t r = rank, go = 0;
    if (!write_all(fd, &r, sizeof r) || !read_all(fd, &go, sizeof go) || go != 1) return bail("handshake");
  }
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  for (int f : c->fd) if (f >= 0) ::close(f);
  if (!c->path.empty()) ::unlink(c->path.c_str());
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { g_group_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  if (g_group_depth <= 0) return fail(ncclInvalidArgument, "ncclGroupEnd without ncclGroupStart");
  if (--g_group_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run_ops(ops);
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
  return post(Op{true, const_cast<void*>(buf), count * dtype_size(t), peer, comm, stream});
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
  return post(Op{false, buf, count * dtype_size(t), peer, comm, stream});
}

}  // extern "C"
