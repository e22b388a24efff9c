// hmpc_group.hip -- multi-device handle group behind the C ABI (include/hector_mpc.h, "device groups").
//
// SURVEY.md section 8(e), literally: every MPC instance is an independent QP, so a batch is cut into contiguous slices,
// one per GPU of the node, solved with no data-path collective; the single exchange step is the gather of the step-0
// wrenches (the 12 values ConvexMPCLocomotion.cpp:419-440 reads through get_solution(0..11)) and status words.
// Single process, one communicator (ncclCommInitAll), grouped ncclAllGather on per-device streams, no host staging in
// the collective; the host copy of the gathered block is one D2H from the first member.
//
// Built only from the public per-device C ABI (hmpc_create / hmpc_upload_records_async / hmpc_solve / ...): a group is
// a composition of handles, it has no access to their internals.
//
// Transport of the gather:
//   HMPC_GROUP_RCCL  librccl is dlopen'ed on first use (the library itself does not link it: a single-GPU user never
//                    loads it, and the process may already hold another copy, e.g. the one PyTorch bundles).
//   HMPC_GROUP_P2P   hipMemcpyPeerAsync of every member's packed slice into every member's gathered buffer
//                    (G*(G-1) small copies over xGMI).  Required when a device is listed twice (RCCL refuses two ranks
//                    on one GPU) -- which is how the slicing, packing, stream ordering and layout are exercised on a
//                    one-GPU box (tests/test_gpu_group.py).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "../../include/hector_mpc.h"

namespace {

// ---- the handful of RCCL entry points, resolved at run time (signatures: rccl.h of ROCm 7.2) ----
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;  // ncclSuccess == 0
enum { NCCL_INT32 = 2 };   // ncclDataType_t: ncclInt8 0, ncclUint8 1, ncclInt32 2
struct Rccl {
  void *dl = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int *) = nullptr;
  int version = 0;
  std::string err;
  bool load() {
    if (dl) return true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      dl = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (dl) break;
    }
    if (!dl) {
      err = std::string("dlopen(librccl): ") + (dlerror() ? dlerror() : "not found");
      return false;
    }
#define SYM(field, name)                                    \
  *(void **)(&field) = dlsym(dl, name);                     \
  if (!field) {                                             \
    err = std::string("librccl lacks ") + name;             \
    return false;                                           \
  }
    SYM(CommInitAll, "ncclCommInitAll")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
    SYM(GetVersion, "ncclGetVersion")
#undef SYM
    // The entry points above are declared by hand against rccl.h of ROCm 7.2 (RCCL 2.2x: NCCL_VERSION_CODE =
    // major * 10000 + minor * 100 + patch since 2.9).  Their signatures and ncclInt32 == 2 have been stable across the
    // whole 2.x line; any other major is refused rather than called through a guessed ABI.
    if (GetVersion(&version) != 0) version = -1;
    if (version < 20000 || version >= 30000) {
      err = "librccl reports version code " + std::to_string(version) + ": only RCCL/NCCL 2.x is supported by this binding";
      dlclose(dl);
      dl = nullptr;
      return false;
    }
    return true;
  }
};
Rccl g_rccl;
thread_local std::string g_group_err;

#define GHIP(expr)                                                          \
  do {                                                                      \
    hipError_t _e = (expr);                                                 \
    if (_e != hipSuccess) {                                                 \
      g_group_err = std::string(#expr) + ": " + hipGetErrorString(_e);      \
      return HMPC_E_HIP;                                                    \
    }                                                                       \
  } while (0)
#define GNCCL(expr)                                                                      \
  do {                                                                                   \
    ncclResult_t _r = (expr);                                                            \
    if (_r != 0) {                                                                       \
      g_group_err = std::string(#expr) + ": " + g_rccl.GetErrorString(_r);               \
      return HMPC_E_HIP;                                                                 \
    }                                                                                    \
  } while (0)

// per instance the exchange carries pw = 6 nc + 1 words: the step-0 wrench the controller reads (ConvexMPCLocomotion.cpp:
// 419-440: F of each contact, then M of each contact -- 12 values for two feet, 18 with the hand contact) + the status word

// forces [n][6 nc h] float, status [n] -> packed [n][pw] 32-bit words (one coalesced row per instance)
__global__ void pack_step0_kernel(const float *__restrict__ forces, const uint32_t *__restrict__ status, int n, int width,
                                  int pw, uint32_t *__restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * pw) return;
  const int i = t / pw, c = t % pw;
  out[t] = (c < pw - 1) ? __float_as_uint(forces[(size_t)i * width + c]) : status[i];
}

struct Member {
  int device = 0;
  hmpc_handle *h = nullptr;
  hipStream_t solve_stream = nullptr, comm_stream = nullptr;
  hipEvent_t packed = nullptr, gathered_ev = nullptr;
  uint32_t *d_pack = nullptr;      // [cap][pw]
  uint32_t *d_gathered = nullptr;  // [G][cap][pw]
  int lo = 0, n = 0;               // slice of the current batch: instances lo, lo + step, ... (n of them)
  int step = 1;                    // 1 = contiguous slice; G = striped deal (hmpc_group_set_deal)
  int posted_lo = 0, posted_n = 0, posted_step = 1; // slice of the batch whose exchange was posted last (may differ from the current one)
};

}  // namespace

struct hmpc_group {
  problem_setup setup;
  int G = 0, cap = 0 /* instances per member buffer = largest possible slice */, max_batch = 0, batch = 0;
  int nc = 2;  // contacts per horizon step of every member handle (hmpc_group_create_ex)
  int pw = 13; // 32-bit words per instance in the exchange: 6 nc + 1
  int transport = HMPC_GROUP_RCCL;
  std::vector<Member> m;
  std::vector<ncclComm_t> comms;
  uint32_t *h_stage = nullptr;  // pinned [G][cap][pw]
  // an exchange has been posted and not yet collected (by hmpc_group_wait_gather or hmpc_group_gather_wrench): the
  // pipelined pattern post(k) / solve(k+1) / gather_wrench() collects solve k's exchange, while a gather_wrench after a
  // collected exchange posts a fresh one for the solves enqueued since
  bool gather_posted = false;
  bool exchange_repair = true; // members run the device-side safe pass inside every solve: the exchange carries repaired rows
  int deal = HMPC_DEAL_CONTIGUOUS;  // how a batch is dealt to the members (hmpc_group_set_deal)
};

namespace {
// member i's share of a batch: first instance, count, index step
inline void member_share(const hmpc_group *g, int batch, int i, int *lo, int *n, int *step) {
  if (g->deal == HMPC_DEAL_STRIPED) {
    *lo = i, *step = g->G, *n = (batch > i) ? (batch - i + g->G - 1) / g->G : 0;
  } else {
    int hi;
    hmpc_shard_bounds(batch, g->G, i, lo, &hi);
    *n = hi - *lo, *step = 1;
  }
}
}  // namespace

#define GENTER() g_group_err.clear() /* a stale group-level message must not shadow a later member-level one */

extern "C" {

const char *hmpc_group_last_error(void) { return g_group_err.empty() ? hmpc_last_hip_error() : g_group_err.c_str(); }

int hmpc_shard_bounds(int global_batch, int n_shards, int index, int *lo, int *hi) {
  if (global_batch < 0 || n_shards < 1 || index < 0 || index >= n_shards || !lo || !hi) return HMPC_E_ARG;
  const int base = global_batch / n_shards, extra = global_batch % n_shards;
  *lo = index * base + (index < extra ? index : extra);
  *hi = *lo + base + (index < extra ? 1 : 0);
  return HMPC_OK;
}

int hmpc_group_destroy(hmpc_group *g) {
  if (!g) return HMPC_E_ARG;
  for (Member &mb : g->m) {
    hipSetDevice(mb.device);
    if (mb.solve_stream) hipStreamSynchronize(mb.solve_stream);
    if (mb.comm_stream) hipStreamSynchronize(mb.comm_stream);
  }
  for (ncclComm_t c : g->comms)
    if (c) g_rccl.CommDestroy(c);
  for (Member &mb : g->m) {
    hipSetDevice(mb.device);
    if (mb.h) hmpc_destroy(mb.h);
    if (mb.d_pack) hipFree(mb.d_pack);
    if (mb.d_gathered) hipFree(mb.d_gathered);
    if (mb.packed) hipEventDestroy(mb.packed);
    if (mb.gathered_ev) hipEventDestroy(mb.gathered_ev);
    if (mb.solve_stream) hipStreamDestroy(mb.solve_stream);
    if (mb.comm_stream) hipStreamDestroy(mb.comm_stream);
  }
  if (g->h_stage) hipHostFree(g->h_stage);
  delete g;
  return HMPC_OK;
}

int hmpc_group_create(hmpc_group **out, const struct problem_setup *setup, const int *devices, int n_devices,
                      int max_batch, int transport) {
  return hmpc_group_create_ex(out, setup, devices, n_devices, max_batch, transport, 2);
}

int hmpc_group_contacts(const hmpc_group *g) { return g ? g->nc : HMPC_E_ARG; }

int hmpc_group_create_ex(hmpc_group **out, const struct problem_setup *setup, const int *devices, int n_devices,
                         int max_batch, int transport, int n_contacts) {
  GENTER();
  if (!out || !setup || n_devices < 1 || max_batch < 1 || (n_contacts != 2 && n_contacts != 3)) return HMPC_E_ARG;
  if (transport != HMPC_GROUP_AUTO && transport != HMPC_GROUP_RCCL && transport != HMPC_GROUP_P2P) return HMPC_E_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    g_group_err = "no HIP device visible (libhector_mpc_hip has no CPU fallback)";
    return HMPC_E_NO_DEVICE;
  }
  int caller_device = 0;
  (void)hipGetDevice(&caller_device);
  struct RestoreDevice {  // the caller's current device is not a side effect of creating a group
    int d;
    ~RestoreDevice() { (void)hipSetDevice(d); }
  } restore{caller_device};
  hmpc_group *g = new (std::nothrow) hmpc_group();
  if (!g) return HMPC_E_ARG;
  g->setup = *setup;
  g->G = n_devices;
  g->max_batch = max_batch;
  g->cap = (max_batch + n_devices - 1) / n_devices;
  g->nc = n_contacts;
  g->pw = 6 * n_contacts + 1;
  g->m.resize(n_devices);
  bool repeated = false;
  for (int i = 0; i < n_devices; ++i) {
    g->m[i].device = devices ? devices[i] : i;
    if (g->m[i].device < 0 || g->m[i].device >= ndev) {
      g_group_err = "device index outside hipGetDeviceCount()";
      hmpc_group_destroy(g);
      return HMPC_E_NO_DEVICE;
    }
    for (int j = 0; j < i; ++j) repeated |= (g->m[j].device == g->m[i].device);
  }
  if (transport == HMPC_GROUP_AUTO) transport = repeated ? HMPC_GROUP_P2P : HMPC_GROUP_RCCL;
  if (transport == HMPC_GROUP_RCCL && repeated) {
    g_group_err = "RCCL refuses two ranks on one device: list distinct devices or use HMPC_GROUP_P2P";
    hmpc_group_destroy(g);
    return HMPC_E_ARG;
  }
  g->transport = transport;
#define GTRY(expr)                 \
  do {                             \
    int _rc = (expr);              \
    if (_rc != HMPC_OK) {          \
      hmpc_group_destroy(g);       \
      return _rc;                  \
    }                              \
  } while (0)
#define GHIPD(expr)                                                      \
  do {                                                                   \
    hipError_t _e = (expr);                                              \
    if (_e != hipSuccess) {                                              \
      g_group_err = std::string(#expr) + ": " + hipGetErrorString(_e);   \
      hmpc_group_destroy(g);                                             \
      return HMPC_E_HIP;                                                 \
    }                                                                    \
  } while (0)
  const size_t slice_words = (size_t)g->cap * g->pw;
  for (Member &mb : g->m) {
    GHIPD(hipSetDevice(mb.device));
    GTRY(hmpc_create_ex(&mb.h, setup, g->cap, mb.device, n_contacts));
    GTRY(hmpc_set_device_repair(mb.h, 1));
    GHIPD(hipStreamCreateWithFlags(&mb.solve_stream, hipStreamNonBlocking));
    GHIPD(hipStreamCreateWithFlags(&mb.comm_stream, hipStreamNonBlocking));
    GHIPD(hipEventCreateWithFlags(&mb.packed, hipEventDisableTiming));
    GHIPD(hipEventCreateWithFlags(&mb.gathered_ev, hipEventDisableTiming));
    GHIPD(hipMalloc(&mb.d_pack, slice_words * sizeof(uint32_t)));
    GHIPD(hipMalloc(&mb.d_gathered, slice_words * sizeof(uint32_t) * g->G));
    GHIPD(hipMemset(mb.d_pack, 0, slice_words * sizeof(uint32_t)));
    GHIPD(hipMemset(mb.d_gathered, 0, slice_words * sizeof(uint32_t) * g->G));
  }
  GHIPD(hipHostMalloc(&g->h_stage, slice_words * sizeof(uint32_t) * g->G, hipHostMallocDefault));
  if (transport == HMPC_GROUP_P2P) {
    for (int i = 0; i < g->G; ++i)
      for (int j = 0; j < g->G; ++j) {
        if (g->m[i].device == g->m[j].device) continue;
        int can = 0;
        GHIPD(hipDeviceCanAccessPeer(&can, g->m[i].device, g->m[j].device));
        if (can) {
          GHIPD(hipSetDevice(g->m[i].device));
          hipError_t e = hipDeviceEnablePeerAccess(g->m[j].device, 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) GHIPD(e);
          (void)hipGetLastError();
        }
      }
  } else {
    if (!g_rccl.load()) {
      g_group_err = g_rccl.err;
      hmpc_group_destroy(g);
      return HMPC_E_HIP;
    }
    g->comms.assign(g->G, nullptr);
    std::vector<int> devs(g->G);
    for (int i = 0; i < g->G; ++i) devs[i] = g->m[i].device;
    ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), g->G, devs.data());
    if (r != 0) {
      g_group_err = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r);
      g->comms.clear();
      hmpc_group_destroy(g);
      return HMPC_E_HIP;
    }
  }
#undef GTRY
#undef GHIPD
  *out = g;
  return HMPC_OK;
}

int hmpc_group_size(const hmpc_group *g) { return g ? g->G : HMPC_E_ARG; }
int hmpc_group_transport(const hmpc_group *g) { return g ? g->transport : HMPC_E_ARG; }
int hmpc_group_batch(const hmpc_group *g) { return g ? g->batch : HMPC_E_ARG; }

int hmpc_group_member_step(const hmpc_group *g, int member) {
  if (!g || member < 0 || member >= g->G) return HMPC_E_ARG;
  return g->m[member].step;
}

int hmpc_group_member(hmpc_group *g, int member, hmpc_handle **handle, int *device, int *lo, int *n, void **solve_stream) {
  if (!g || member < 0 || member >= g->G) return HMPC_E_ARG;
  const Member &mb = g->m[member];
  if (handle) *handle = mb.h;
  if (device) *device = mb.device;
  if (lo) *lo = mb.lo;
  if (n) *n = mb.n;
  if (solve_stream) *solve_stream = (void *)mb.solve_stream;
  return HMPC_OK;
}

// contiguous slices of a host batch -> the members' record buffers (asynchronous on each member's solve stream)
int hmpc_group_upload_records(hmpc_group *g, const void *host_records, int batch) {
  GENTER();
  if (!g || (!host_records && batch > 0) || batch < 0) return HMPC_E_ARG;
  if (batch > g->max_batch) return HMPC_E_BATCH;
  const size_t stride = hmpc_record_stride_ex(g->setup.horizon, g->nc);
  static const unsigned char empty = 0;
  if (!host_records) host_records = &empty;  // batch == 0
  for (int i = 0; i < g->G; ++i) {
    Member &mb = g->m[i];
    member_share(g, batch, i, &mb.lo, &mb.n, &mb.step);
    // (striped deal: one strided copy per member -- rows lo, lo + G, ... of the host batch -- no host staging)
    const int rc = hmpc_upload_records_strided_async(mb.h, (const unsigned char *)host_records + (size_t)mb.lo * stride, mb.n,
                                                     (size_t)mb.step * stride, mb.solve_stream);
    if (rc != HMPC_OK) return rc;
  }
  g->batch = batch;
  return HMPC_OK;
}

// How a batch is dealt to the members.  HMPC_DEAL_CONTIGUOUS (default): member i holds the contiguous slice of
// hmpc_shard_bounds.  HMPC_DEAL_STRIPED: member i holds instances i, i + G, i + 2 G, ... -- a parameter sweep is usually ORDERED
// (by commanded velocity, by gait phase ...), so its hard instances sit next to each other, a contiguous slice hands them all to
// one GPU, and the gather waits for that one; dealt round-robin every member gets the same mix (measured: tests/test_gpu_group.py
// ::test_striped_deal_balances_a_skewed_batch).  Same member sizes either way.  Host-facing results (hmpc_group_gather_wrench,
// hmpc_group_download) are in instance order in both modes; the device-resident gathered block keeps its [member][row] layout,
// row r of slot s being instance s + r G in striped mode.  Takes effect with the next upload / set_device_records.
int hmpc_group_set_params(hmpc_group *g, const struct hmpc_params *p) {
  if (!g) return HMPC_E_ARG;
  GENTER();
  for (auto &mm : g->m) {
    const int rc = hmpc_set_params(mm.h, p);
    if (rc != HMPC_OK) return rc;
  }
  return HMPC_OK;
}

int hmpc_group_set_deal(hmpc_group *g, int deal) {
  GENTER();
  if (!g || (deal != HMPC_DEAL_CONTIGUOUS && deal != HMPC_DEAL_STRIPED)) return HMPC_E_ARG;
  g->deal = deal;
  return HMPC_OK;
}
int hmpc_group_deal(const hmpc_group *g) { return g ? g->deal : HMPC_E_ARG; }

// records already resident on each member's device: slice sizes are taken from the shard arithmetic, the caller supplies
// one device pointer per member (device_records[i] points at that member's first record, on that member's device)
int hmpc_group_set_device_records(hmpc_group *g, const void *const *device_records, int batch, int max_reduced_vars) {
  GENTER();
  if (!g || !device_records || batch < 0) return HMPC_E_ARG;
  if (batch > g->max_batch) return HMPC_E_BATCH;
  for (int i = 0; i < g->G; ++i) {
    Member &mb = g->m[i];
    member_share(g, batch, i, &mb.lo, &mb.n, &mb.step);  // (striped deal: the caller's buffers hold instances i, i + G, ...)
    if (mb.n > 0 && !device_records[i]) return HMPC_E_ARG;
    int rc = hmpc_set_device_records(mb.h, mb.n > 0 ? device_records[i] : (const void *)mb.d_pack, mb.n);
    if (rc == HMPC_OK) rc = hmpc_set_max_reduced_vars(mb.h, max_reduced_vars);
    if (rc != HMPC_OK) return rc;
  }
  g->batch = batch;
  return HMPC_OK;
}

// enqueue the solve of every member's slice on its own stream; returns without waiting
int hmpc_group_solve(hmpc_group *g) {
  GENTER();
  if (!g) return HMPC_E_ARG;
  for (Member &mb : g->m) {
    GHIP(hipSetDevice(mb.device));
    const int rc = hmpc_solve(mb.h, mb.solve_stream);
    if (rc != HMPC_OK) return rc;
  }
  return HMPC_OK;
}

int hmpc_group_solve_command_sweep(hmpc_group *g, int group_size) {
  GENTER();
  if (!g || group_size < 1) return HMPC_E_ARG;
  if (g->deal != HMPC_DEAL_CONTIGUOUS) return HMPC_E_ARG;  // a striped deal would tear the groups apart
  for (Member &mb : g->m)
    if (mb.n % group_size != 0) return HMPC_E_ARG;         // every slice: whole groups only
  for (Member &mb : g->m) {
    GHIP(hipSetDevice(mb.device));
    const int rc = hmpc_solve_command_sweep(mb.h, group_size, mb.solve_stream);
    if (rc != HMPC_OK) return rc;
  }
  return HMPC_OK;
}

// The exchange carries repaired rows (default on): every member's solve is followed, on the same stream and without a
// host round trip, by the safe variant over the instances the fast variant flagged (hmpc_set_device_repair), so what
// pack_step0_kernel reads is the repaired wrench and status.  Off = the fast pass's results as they are (a flagged
// instance then shows in its status word and hmpc_group_download repairs it).
int hmpc_group_set_exchange_repair(hmpc_group *g, int on) {
  GENTER();
  if (!g) return HMPC_E_ARG;
  for (size_t i = 0; i < g->m.size(); ++i) {
    const int rc = hmpc_set_device_repair(g->m[i].h, on);
    if (rc != HMPC_OK) {  // all members or none: the ones already switched go back to the previous setting
      for (size_t j = 0; j < i; ++j) (void)hmpc_set_device_repair(g->m[j].h, g->exchange_repair ? 1 : 0);
      return rc;
    }
  }
  g->exchange_repair = on != 0;
  return HMPC_OK;
}

// post the exchange step for the solves enqueued so far: pack on the solve stream (so the NEXT solve may overwrite the
// force buffer at once), all-gather on the comm stream (so it runs under that next solve).  Does not block.
int hmpc_group_post_gather(hmpc_group *g) {
  GENTER();
  if (!g) return HMPC_E_ARG;
  const int width = 6 * g->nc * g->setup.horizon, PACK_WORDS = g->pw;
  for (Member &mb : g->m) {
    GHIP(hipSetDevice(mb.device));
    // the previous gather still reads this member's d_pack (RCCL: on its own comm stream; P2P: on every destination's):
    // the pack below must not overtake it.  By now that gather has had a whole solve to run under.
    if (g->transport == HMPC_GROUP_P2P)
      for (Member &dst : g->m) GHIP(hipStreamWaitEvent(mb.solve_stream, dst.gathered_ev, 0));
    else
      GHIP(hipStreamWaitEvent(mb.solve_stream, mb.gathered_ev, 0));
    if (mb.n > 0) {
      float *d_forces = nullptr;
      uint32_t *d_status = nullptr;
      const int rc = hmpc_get_device_outputs(mb.h, &d_forces, &d_status);
      if (rc != HMPC_OK) return rc;
      const int total = mb.n * PACK_WORDS;
      hipLaunchKernelGGL(pack_step0_kernel, dim3((total + 255) / 256), dim3(256), 0, mb.solve_stream, d_forces, d_status,
                         mb.n, width, PACK_WORDS, mb.d_pack);
      GHIP(hipGetLastError());
    }
    GHIP(hipEventRecord(mb.packed, mb.solve_stream));
    GHIP(hipStreamWaitEvent(mb.comm_stream, mb.packed, 0));
    mb.posted_lo = mb.lo, mb.posted_n = mb.n, mb.posted_step = mb.step;  // hmpc_group_gather_wrench unpacks THIS batch's slices
  }
  const size_t slice_words = (size_t)g->cap * PACK_WORDS;
  if (g->transport == HMPC_GROUP_RCCL) {
    GNCCL(g_rccl.GroupStart());
    for (int i = 0; i < g->G; ++i) {
      Member &mb = g->m[i];
      ncclResult_t r = g_rccl.AllGather(mb.d_pack, mb.d_gathered, slice_words, NCCL_INT32, g->comms[i], mb.comm_stream);
      if (r != 0) {
        g_rccl.GroupEnd();
        g_group_err = std::string("ncclAllGather: ") + g_rccl.GetErrorString(r);
        return HMPC_E_HIP;
      }
    }
    GNCCL(g_rccl.GroupEnd());
  } else {
    // every member pushes its packed slice into slot i of every member's gathered buffer; a destination's comm stream
    // first waits for the source's pack event
    for (int j = 0; j < g->G; ++j) {
      Member &dst = g->m[j];
      GHIP(hipSetDevice(dst.device));
      for (int i = 0; i < g->G; ++i) {
        Member &src = g->m[i];
        if (i != j) GHIP(hipStreamWaitEvent(dst.comm_stream, src.packed, 0));
        uint32_t *slot = dst.d_gathered + (size_t)i * slice_words;
        if (src.device == dst.device)
          GHIP(hipMemcpyAsync(slot, src.d_pack, slice_words * sizeof(uint32_t), hipMemcpyDeviceToDevice, dst.comm_stream));
        else
          GHIP(hipMemcpyPeerAsync(slot, dst.device, src.d_pack, src.device, slice_words * sizeof(uint32_t), dst.comm_stream));
      }
    }
  }
  for (Member &mb : g->m) {
    GHIP(hipSetDevice(mb.device));
    GHIP(hipEventRecord(mb.gathered_ev, mb.comm_stream));
  }
  g->gather_posted = true;
  return HMPC_OK;
}

// gathered device copy held by `member`: wrench/status interleaved as [G][cap][6 nc + 1] words; slot s holds member s's
// slice (instances lo_s .. lo_s + n_s - 1 in rows 0 .. n_s - 1).  Valid after hmpc_group_wait_gather.
int hmpc_group_device_gathered(hmpc_group *g, int member, const uint32_t **gathered, int *slot_rows) {
  if (!g || member < 0 || member >= g->G || !gathered) return HMPC_E_ARG;
  *gathered = g->m[member].d_gathered;
  if (slot_rows) *slot_rows = g->cap;
  return HMPC_OK;
}

int hmpc_group_wait_gather(hmpc_group *g) {
  GENTER();
  if (!g) return HMPC_E_ARG;
  g->gather_posted = false;  // collected: the caller reads hmpc_group_device_gathered from here on
  for (Member &mb : g->m) {
    GHIP(hipSetDevice(mb.device));
    GHIP(hipStreamSynchronize(mb.comm_stream));
  }
  return HMPC_OK;
}

// the exchange step, blocking form: post (if not posted yet), wait, and copy the gathered block of the first member to
// the host in instance order: wrench [batch][12] float, status [batch] (either may be NULL)
int hmpc_group_gather_wrench(hmpc_group *g, float *host_wrench, uint32_t *host_status) {
  GENTER();
  if (!g) return HMPC_E_ARG;
  if (!g->gather_posted) {
    const int rc = hmpc_group_post_gather(g);
    if (rc != HMPC_OK) return rc;
  }
  g->gather_posted = false;
  Member &m0 = g->m[0];
  GHIP(hipSetDevice(m0.device));
  const int PACK_WORDS = g->pw, nw = g->pw - 1;
  const size_t slice_words = (size_t)g->cap * PACK_WORDS;
  GHIP(hipMemcpyAsync(g->h_stage, m0.d_gathered, slice_words * sizeof(uint32_t) * g->G, hipMemcpyDeviceToHost, m0.comm_stream));
  GHIP(hipStreamSynchronize(m0.comm_stream));
  for (int s = 0; s < g->G; ++s) {
    const Member &mb = g->m[s];
    const uint32_t *rows = g->h_stage + (size_t)s * slice_words;
    for (int i = 0; i < mb.posted_n; ++i) {
      const size_t inst = (size_t)mb.posted_lo + (size_t)i * mb.posted_step;
      if (host_wrench) memcpy(host_wrench + inst * nw, rows + (size_t)i * PACK_WORDS, nw * sizeof(float));
      if (host_status) host_status[inst] = rows[(size_t)i * PACK_WORDS + nw];
    }
  }
  return HMPC_OK;
}

// every member's full force block [n][12h] and status words to the host, in instance order (no collective: G D2H copies);
// flagged instances get the members' safe pass exactly as hmpc_download gives it
int hmpc_group_download(hmpc_group *g, float *forces, uint32_t *status) {
  GENTER();
  if (!g) return HMPC_E_ARG;
  const size_t width = (size_t)6 * g->nc * g->setup.horizon;
  std::vector<float> tf;
  std::vector<uint32_t> ts;
  for (Member &mb : g->m) {
    if (mb.step == 1) {
      const int rc = hmpc_download(mb.h, forces ? forces + (size_t)mb.lo * width : nullptr, status ? status + mb.lo : nullptr);
      if (rc != HMPC_OK) return rc;
      continue;
    }
    // striped deal: the member's block comes back contiguous and is dealt out to instances lo, lo + step, ...
    tf.resize(forces ? (size_t)mb.n * width : 0);
    ts.resize(status ? (size_t)mb.n : 0);
    const int rc = hmpc_download(mb.h, forces ? tf.data() : nullptr, status ? ts.data() : nullptr);
    if (rc != HMPC_OK) return rc;
    for (int i = 0; i < mb.n; ++i) {
      const size_t inst = (size_t)mb.lo + (size_t)i * mb.step;
      if (forces) memcpy(forces + inst * width, tf.data() + (size_t)i * width, width * sizeof(float));
      if (status) status[inst] = ts[i];
    }
  }
  return HMPC_OK;
}

int hmpc_group_synchronize(hmpc_group *g) {
  GENTER();
  if (!g) return HMPC_E_ARG;
  for (Member &mb : g->m) {
    GHIP(hipSetDevice(mb.device));
    GHIP(hipStreamSynchronize(mb.solve_stream));
    GHIP(hipStreamSynchronize(mb.comm_stream));
  }
  return HMPC_OK;
}

}  // extern "C"
