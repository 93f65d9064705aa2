// host_comm.cpp -- the path's only exchange between GPUs, on the library's own runtime.
//
// Replicas never interact (the reference runs them in separate processes, tnco/parallel.py:330-341, and only
// sorts the results, tnco/app/infinite_memory/sa.py:243-257), so N ranks -- one process per GPU -- need:
// an all-reduce(min) of the best cost, an all-gather of the few best results, a barrier.  RCCL over xGMI,
// bound HERE with dlopen("librccl.so") of the ROCm installation, i.e. the same HIP / HSA runtime this
// library links: no pointer crosses between PyTorch's bundled runtime and the system's (round 2 reduced into
// a torch tensor: two HIP runtimes writing each other's memory in one process).  The rendezvous (the 128-byte
// ncclUniqueId from rank 0 to the others) is the caller's: tnco_amd/parallel.py sends it over TCP (the side
// channel the ranks share, or a socket on MASTER_ADDR : MASTER_PORT + 17).
#include "../../include/tnco_hip.h"
#include "dev_cache.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

namespace {
// One message per process (ncclCommInitRank runs on a helper thread of the caller's: the thread that asks for the
// text is not the one that failed).  tnco_hip_comm_last_error hands out a per-thread copy.
std::mutex g_comm_err_mu;
std::string g_comm_err;
void set_comm_err(const std::string& m) {
  std::lock_guard<std::mutex> lk(g_comm_err_mu);
  g_comm_err = m;
}

// the directory a shared object was loaded from, by one of its symbols
std::string dir_of(const void* sym) {
  Dl_info info;
  if (!dladdr(sym, &info) || !info.dli_fname) return std::string();
  std::string p(info.dli_fname);
  const size_t k = p.rfind('/');
  return k == std::string::npos ? std::string() : p.substr(0, k);
}

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool load() {
    if (lib) return true;
    // RCCL must sit on the HIP runtime THIS library links -- a process may hold a second one (PyTorch's bundled
    // copy, with an RCCL of its own): only absolute paths, first the directory our libamdhip64 came from; a bare
    // soname could resolve to whatever copy the process has already mapped.
    const std::string hipdir = dir_of(reinterpret_cast<const void*>(&hipGetDeviceCount));
    std::string tried;
    for (const std::string& dir : {hipdir, std::string("/opt/rocm/lib")}) {
      if (dir.empty()) continue;
      for (const char* nm : {"/librccl.so.1", "/librccl.so"}) {
        const std::string path = dir + nm;
        lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
        tried += path + " ";
      }
      if (lib) break;
    }
    if (!lib) {
      set_comm_err("librccl.so not found next to the HIP runtime (tried: " + tried + ")");
      return false;
    }
#define TNCO_SYM(field, name)                                   \
  field = reinterpret_cast<decltype(field)>(dlsym(lib, name)); \
  if (!field) { set_comm_err(std::string("librccl.so lacks ") + name); return false; }
    TNCO_SYM(GetUniqueId, "ncclGetUniqueId")
    TNCO_SYM(CommInitRank, "ncclCommInitRank")
    TNCO_SYM(CommDestroy, "ncclCommDestroy")
    TNCO_SYM(AllReduce, "ncclAllReduce")
    TNCO_SYM(AllGather, "ncclAllGather")
    TNCO_SYM(GetErrorString, "ncclGetErrorString")
#undef TNCO_SYM
    return true;
  }
};
Rccl g_rccl;
}  // namespace

struct tnco_hip_comm_s {
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  uint8_t* dbuf = nullptr;  // device staging: [world + 1] slots of slot_bytes
  size_t slot_bytes = 0;
};

namespace {
int cfail(int code, const std::string& msg) {
  set_comm_err(msg);
  return code;
}
#define CH(expr)                                                                                  \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return cfail(TNCO_HIP_ERUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define CN(expr)                                                                                  \
  do {                                                                                            \
    ncclResult_t r_ = (expr);                                                                     \
    if (r_ != ncclSuccess) return cfail(TNCO_HIP_ERUNTIME, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
  } while (0)

int ensure_slots(tnco_hip_comm_s* c, size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes <= c->slot_bytes) return TNCO_HIP_OK;
  if (c->dbuf) (void)hipFree(c->dbuf);
  c->dbuf = nullptr;
  c->slot_bytes = 0;
  CH(tnco::dev_malloc((void**)&c->dbuf, bytes * (size_t)(c->world + 1)));
  c->slot_bytes = bytes;
  return TNCO_HIP_OK;
}
}  // namespace

extern "C" {

const char* tnco_hip_comm_last_error(void) {
  thread_local std::string copy;
  std::lock_guard<std::mutex> lk(g_comm_err_mu);
  copy = g_comm_err;
  return copy.c_str();
}

int tnco_hip_comm_unique_id(uint8_t* id128) {
  if (!id128) return cfail(TNCO_HIP_EINVAL, "null argument.");
  if (!g_rccl.load()) return TNCO_HIP_ERUNTIME;
  ncclUniqueId id;
  CN(g_rccl.GetUniqueId(&id));
  static_assert(sizeof(id) == 128, "ncclUniqueId");
  std::memcpy(id128, &id, 128);
  return TNCO_HIP_OK;
}

int tnco_hip_comm_init(int rank, int world, const uint8_t* id128, int device, tnco_hip_comm* out) {
  if (!out || !id128) return cfail(TNCO_HIP_EINVAL, "null argument.");
  *out = nullptr;
  if (world <= 0 || rank < 0 || rank >= world) return cfail(TNCO_HIP_EINVAL, "'rank' / 'world' are not valid.");
  if (!g_rccl.load()) return TNCO_HIP_ERUNTIME;
  CH(hipSetDevice(device));
  tnco_hip_comm_s* c = new tnco_hip_comm_s();
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId id;
  std::memcpy(&id, id128, 128);
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return cfail(TNCO_HIP_ERUNTIME, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
  }
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return cfail(TNCO_HIP_ERUNTIME, std::string("hipStreamCreate: ") + hipGetErrorString(e));
  }
  if (int rc = ensure_slots(c, 256)) {
    tnco_hip_comm_destroy(c);
    return rc;
  }
  *out = c;
  return TNCO_HIP_OK;
}

void tnco_hip_comm_destroy(tnco_hip_comm c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  if (c->dbuf) (void)hipFree(c->dbuf);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int tnco_hip_comm_allreduce_min(tnco_hip_comm c, tnco_hip_handle h, double local, double* out_min) {
  if (!c || !out_min) return cfail(TNCO_HIP_EINVAL, "null argument.");
  CH(hipSetDevice(c->device));
  double* d = reinterpret_cast<double*>(c->dbuf);
  if (h) {  // the handle's replicas, reduced on the device straight into the collective's operand
    if (int rc = tnco_hip_min_cost_device(h, d)) return cfail(rc, tnco_hip_last_error());
  } else {
    CH(hipMemcpy(d, &local, 8, hipMemcpyHostToDevice));
  }
  CN(g_rccl.AllReduce(d, d, 1, ncclDouble, ncclMin, c->comm, c->stream));
  CH(hipMemcpyAsync(out_min, d, 8, hipMemcpyDeviceToHost, c->stream));
  CH(hipStreamSynchronize(c->stream));
  return TNCO_HIP_OK;
}

int tnco_hip_comm_allgather(tnco_hip_comm c, const void* send, void* recv, uint64_t bytes) {
  if (!c || (bytes && (!send || !recv))) return cfail(TNCO_HIP_EINVAL, "null argument.");
  if (bytes == 0) return TNCO_HIP_OK;
  CH(hipSetDevice(c->device));
  if (int rc = ensure_slots(c, (size_t)bytes)) return rc;
  uint8_t* mine = c->dbuf;                  // slot 0: this rank's piece
  uint8_t* all = c->dbuf + c->slot_bytes;   // slots 1..world: everybody's, packed
  CH(hipMemcpy(mine, send, (size_t)bytes, hipMemcpyHostToDevice));
  CN(g_rccl.AllGather(mine, all, (size_t)bytes, ncclChar, c->comm, c->stream));
  CH(hipMemcpyAsync(recv, all, (size_t)bytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
  CH(hipStreamSynchronize(c->stream));
  return TNCO_HIP_OK;
}

int tnco_hip_device_name(int device, char* buf, int cap) {
  if (!buf || cap < 8) return cfail(TNCO_HIP_EINVAL, "null argument.");
  hipDeviceProp_t prop;
  CH(hipGetDeviceProperties(&prop, device));
  char uuid[33];
  for (int i = 0; i < 16; ++i) std::snprintf(uuid + 2 * i, 3, "%02x", (unsigned)(unsigned char)prop.uuid.bytes[i]);
  std::snprintf(buf, (size_t)cap, "%s|pci %04x:%02x:%02x|uuid %s|%d CUs", prop.name, prop.pciDomainID, prop.pciBusID,
                prop.pciDeviceID, uuid, prop.multiProcessorCount);
  return TNCO_HIP_OK;
}

int tnco_hip_comm_barrier(tnco_hip_comm c) {
  double x = 0;
  return tnco_hip_comm_allreduce_min(c, nullptr, 0.0, &x);
}

}  // extern "C"
