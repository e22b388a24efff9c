// hmpc_capi.hip -- host side of libhector_mpc_hip.so: the C ABI of include/hector_mpc.h over the gfx950 kernel.
//
// Reference interface this replaces (a maintainer drops the library in place of these translation units):
//   ConvexMPC/convexMPC_interface.cpp:42-118  setup_problem / update_problem_data / get_solution / update_solver_settings
//   ConvexMPC/SolverMPC.cpp:94-97, 371-738    get_q_soln / solve_mpc
// There is NO CPU fallback: without a gfx950 device every entry point fails (HMPC_E_NO_DEVICE) and the legacy
// entry points print the error and leave the previous solution in place.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "../../include/hector_mpc.h"
#include "hmpc_kernel_args.h"
#include "hmpc_variants.h"
#include "hmpc_builder.h"

namespace {

thread_local std::string g_hip_err;

#define HIP_TRY(expr)                                                                                    \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess) {                                                                              \
      g_hip_err = std::string(#expr) + ": " + hipGetErrorString(_e);                                     \
      return HMPC_E_HIP;                                                                                 \
    }                                                                                                    \
  } while (0)

// The kernel family (instantiated in hmpc_variants.hip, one translation unit per group so that the build runs in parallel).
// NMAX = reduced variables held on chip (6 per stance leg-step); 120 -> 256-thread workgroups (210 register blocks),
// 60 (single support over h <= 10) -> 128-thread workgroups (55 blocks).  QCAP = working-set capacity: the fast variants
// hold 64 rows (49-50 KB LDS, 168 VGPRs: three workgroups per CU, also at h = 20); the "safe"
// variants hold NMAX rows (can never overflow)
// and re-solve the few instances the fast pass flags (hmpc_resolve_failed).
// The extension with a third (hand) contact -- BASELINE config 5, 180 variables x 240 rows at h = 10 -- runs 256-thread
// workgroups with two register blocks per thread (465 blocks, two workgroups per CU; round 2: 512 threads, one per CU --
// still the shape of its safe pass).  BPT = blocks per thread, see hmpc_kernel.h.
const Variant *variants() {
  static const Variant v[] = {hmpc_variant_0(), hmpc_variant_1(), hmpc_variant_2(),  hmpc_variant_3(),
                              hmpc_variant_4(), hmpc_variant_5(), hmpc_variant_6(),  hmpc_variant_7(),
                              hmpc_variant_8(), hmpc_variant_9(), hmpc_variant_10(), hmpc_variant_11(),
                              hmpc_variant_12(), hmpc_variant_13(), hmpc_variant_14(), hmpc_variant_15()};
  return v;
}
constexpr int N_FAST = 4;       // two-contact fast variants [0, N_FAST), their safe variants N_FAST + (h > 10)
constexpr int V3_FAST = 6, V3_SAFE = 7, V3_FAST_512 = 8;  // (V3_FAST_512: the one-workgroup-per-CU variant of round 2, HMPC_3C_512=1)
// double support over more than ten steps (two contacts, 121 .. 240 reduced variables, h <= 20): 820 register blocks on 512
// threads, two each, one workgroup per CU; working set up to HMPC_QCAP_WIDE rows (what 160 KB of LDS leave room for).
constexpr int V2_WIDE = 9;
// QCAP = 0: working set as large as the variable count with the packed Schur inverse in GLOBAL memory (a scratch slice per
// workgroup; 231 KB for 240 variables -- more than a CU's LDS): the safe pass of the wide variant, and the second safe pass of
// the three-contact one (whose LDS-resident safe variant holds 140 of 180 possible rows)
constexpr int V2_WIDE_SAFE = 10, V3_SAFE_G = 11;
// the CONTINUATION variants of the 120-variable shapes (h <= 10, h <= 20): working set of HMPC_QCAP_CONT = 96 rows, 70 KB of LDS = two
// workgroups per CU; they take over -- state and all -- the solves whose working set outgrew the fast variants' 64 rows, run block
// rounds of up to 96 rows on them, and leave what outgrows them in turn (HMPC_S_WORKSET again) to the 120-row safe variants
constexpr int V2_CONT = 12;  // + (h > 10)
// command sweeps (MODE 1, hmpc_solve_command_sweep): a workgroup solves a chunk of instances that share state and gait on ONE
// inverse -- the 120-variable h <= 10 shape and the 60-variable (single support) one
constexpr int V2_SWEEP_120 = 14, V2_SWEEP_60 = 15;
constexpr int N_VARIANTS = 16;
constexpr int MAX_VARS_ANY = 240;
constexpr int DBG_FLOATS_MAX = hmpc::DbgLayout<240, 2>::TOTAL > hmpc::DbgLayout<180, 3>::TOTAL
                                   ? hmpc::DbgLayout<240, 2>::TOTAL
                                   : hmpc::DbgLayout<180, 3>::TOTAL;

int fixed_floats(int nc) { return nc == 3 ? 73 : 54; }
size_t record_stride(int h, int nc = 2) { return (size_t)(((fixed_floats(nc) + 12 * h) * 4 + nc * h + 15) / 16 * 16); }

}  // namespace

struct hmpc_handle {
  problem_setup setup;
  int nc;  // contacts per horizon step: 2 (reference) or 3 (hand-contact extension)
  int max_batch, device, batch;
  size_t stride;
  unsigned char *d_records_own;
  const unsigned char *d_records;
  float *d_forces_own, *d_forces;
  uint32_t *d_status_own, *d_status;
  double *d_x64, *d_obj64;
  float *d_dbg_f;
  int *d_dbg_i;
  long long *d_prof;
  int warm;        // block warm start of the working set (default on)
  signed char *d_wset;  // working sets carried from tick to tick (hmpc_set_tick_warm_start), [max_batch][8 nc h]
  int tick_warm, tick_shift;
  int auto_resolve;  // hmpc_download re-solves flagged instances with the safe variant (default on)
  int max_stance;  // max reduced variables of the current batch (known only for host-uploaded records; else -1)
  hipStream_t last_stream;
  bool attrs_set[N_VARIANTS];
  // persistent device scratch for the host-pointer convenience entry points (grown on demand, freed in hmpc_destroy):
  // no hipMalloc/hipFree per call and nothing to leak on an early error return
  void *d_scratch;
  size_t scratch_bytes;
  // number of instances the solve kernels have flagged (working set full / max-iter / infeasible / KKT) since the handle
  // was created; monotonically increasing device counter, hmpc_download compares it with the value it saw last and
  // skips the status scan of the safe pass when nothing new was flagged
  unsigned int *d_flagged;
  unsigned int flagged_seen;
  // device-side safe pass (hmpc_set_device_repair): list of the instances the last fast launch flagged + its counter
  int device_repair;
  int *d_flag_list;
  unsigned int *d_flag_count;
  // parity hook (hmpc_debug_solve_external_qp): device copies of caller-supplied QP data, only set during that call
  const float *d_ext_H, *d_ext_g, *d_ext_Fc;
  int ext_ld;
  int iter_cap;  // hmpc_set_max_iterations: cap on the active-set iterations of every solve (0 = the variant's own bound)
  // hmpc_set_dispatch_order: 1 = workgroups take the instances longest-previous-solve first (d_order, rebuilt at the head of
  // every solve from the status words the previous solve of a batch of the same size left; order_batch = that size, 0 = none)
  int dispatch_order, order_batch;
  bool order_valid;
  int *d_order;
  unsigned char *d_keys;  // predicted cost bucket per instance (cold-handle order), allocated on first use
  // size classes of a device-resident batch whose widest reduced QP the host was not told (hmpc_set_max_reduced_vars < 0):
  // stance leg-steps per instance, written on the device by the record builder (cls_valid) or, for records handed in by
  // pointer, by classify_records_kernel at the head of every solve
  unsigned char *d_cls;
  int cls_valid;
  // packed Schur inverses of the EGLOBAL safe variants: [e_slices][nmax (nmax + 1) / 2] doubles, grown on demand
  double *d_escratch;
  size_t e_bytes;
  // hand-over of full working sets (KernelArgs::spill): one slot per instance (slot = instance index), allocated on the first
  // launch of a variant that saves its state; spill_stride = bytes per slot of the allocation, spill_cap = slots
  unsigned char *d_spill;
  int *d_spill_slot;
  size_t spill_stride;
  int spill_cap;
  int handover;  // hmpc_set_handover (default on)
  hmpc_params params;  // robot / contact constants (hmpc_set_params; defaults = the reference's literals)
  const float *d_mu_inst;  // hmpc_set_instance_mu: per-instance friction parameter in HBM (caller-owned), nullptr = params.mu for all
  double *d_reg_rho;  // Hessians that are not positive definite: the pivot the safe variant found, then rho of hmpc_resolve_failed's regularisation steps, per instance (allocated with the first list launch)
  double *d_sweep_m;  // command sweeps: every group's M = H^-1, [groups][36][threads per workgroup] doubles (grown on demand)
  size_t sweep_m_bytes;
};
// longest-first dispatch (hmpc_set_dispatch_order, on by default): only where a launch has a tail to shorten -- more instances
// than the ~512-1536 workgroup slots of the chip -- and not beyond what the one-workgroup sort handles in a few microseconds
constexpr int DISPATCH_ORDER_MIN_BATCH = 512, DISPATCH_ORDER_MAX_BATCH = 32768;
constexpr int REPAIR_GRID_CAP = 65536;  // workgroups of the device-side safe launch = most instances it can repair per solve (until round 6: 2 048; workgroups beyond the flagged count leave at once)
constexpr double SAFE_PASS_RELAX = 1e-6;  // first safe pass (device- and host-driven): bounds moved outward by this, exact re-solve + exact KKT check at its end
constexpr int SPILL_SLOT_CAP = 32768;   // hand-over slots per handle at most (101 KB each for 120 variables: 3.3 GB of the 288 GB); instances beyond it are re-solved cold
constexpr int EGLOBAL_CHUNK = 512;         // host-driven safe pass of the global-E variants: instances per launch (118 MB of scratch at 240 variables)
constexpr int DEVICE_REG_MIN_HORIZON = 10;  // device-side chain: the regularisation launches are enqueued for longer horizons only (enqueue_reg_steps)
constexpr int REG_LIST_CAP = 256;  // device-side chain: instances per solve whose Hessian is not positive definite that get the regularisation launches (the rest: hmpc_resolve_failed)
constexpr int REPAIR_GRID_CAP_WIDE = 4096;  // ... of the wide variant's, whose safe pass keeps 231 KB per workgroup in global memory (0.95 GB of 288: the list mixes size classes, so a small cap could leave a wide instance behind 256 others unrepaired -- ADVICE round 5)

// returns a device buffer of at least `bytes` owned by the handle (contents undefined)
static int scratch(hmpc_handle *h, size_t bytes, void **out) {
  if (bytes > h->scratch_bytes) {
    if (h->d_scratch) {
      HIP_TRY(hipDeviceSynchronize());  // a previous user of the old buffer may still be in flight
      HIP_TRY(hipFree(h->d_scratch));
      h->d_scratch = nullptr;
      h->scratch_bytes = 0;
    }
    const size_t want = (bytes + 4095) & ~(size_t)4095;
    HIP_TRY(hipMalloc(&h->d_scratch, want));
    h->scratch_bytes = want;
  }
  *out = h->d_scratch;
  return HMPC_OK;
}

// RAII for the two timing events of hmpc_time_solve
struct EventPair {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ~EventPair() {
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
  }
};

static const Variant &pick_variant(const hmpc_handle *h, int *index) {
  const Variant *v = variants();
  const int hz = h->setup.horizon;
  if (h->nc == 3) {
    static const bool old512 = getenv("HMPC_3C_512") && getenv("HMPC_3C_512")[0] == '1';  // developer A/B switch
    const int vi3 = old512 ? V3_FAST_512 : V3_FAST;
    if (index) *index = vi3;
    return v[vi3];
  }
  if (h->max_stance > HMPC_MAX_VARS && hz > 10) {  // double support beyond ten steps: the wide variant
    if (index) *index = V2_WIDE;
    return v[V2_WIDE];
  }
  int best = -1;
  for (int i = 0; i < N_FAST; ++i) {
    if (v[i].hmax < hz) continue;
    if (h->max_stance >= 0 && v[i].nmax < h->max_stance) continue;
    if (h->max_stance < 0 && v[i].nmax < HMPC_MAX_VARS) continue;
    if (best < 0 || v[i].smem < v[best].smem) best = i;
  }
  if (best < 0) best = N_FAST - 1;  // oversize batches are reported per instance (HMPC_S_TOO_LARGE)
  if (index) *index = best;
  return v[best];
}

struct LaunchOpt {
  bool assemble_only = false;
  int dbg_index = 0;
  const int *d_index_list = nullptr;  // workgroup b solves instance d_index_list[b] with the SAFE variant (re-solve of flagged ones)
  int n_list = 0;
  double relax = 0.0;
  int warm = -1;           // -1 = the handle's setting, 0/1 = override for this launch (the safe pass chooses per pass without touching the handle)
  bool carry_wset = true;  // false keeps a repeated launch of the same batch from consuming/advancing the tick-to-tick working sets
  const unsigned int *d_list_count = nullptr;
  bool record_flagged = false;
  bool longest_first = false;  // workgroup b takes instance h->d_order[b] (enqueue_solve, hmpc_set_dispatch_order)
  int variant = -1;        // -1 = pick_variant; else this entry of variants() (the size-class launches)
  bool ultimate = false;   // safe pass, second level (three contacts): the variant whose working set cannot overflow
  int cls_lo = 0, cls_hi = -1;  // cls_hi >= 0: only instances whose size class lies in [cls_lo, cls_hi] (h->d_cls)
  int safe_variant = -1;   // safe pass over an index list: this entry of variants() instead of the one derived from pick_variant
  bool resume = false;     // safe pass over an index list: instances whose fast solve left its state in a hand-over slot continue from it
  bool continuation = false;  // list launch of the CONTINUATION variant (V2_CONT): only instances with a hand-over slot, everything else on the list is left alone
  int skip_ok = 0;         // list launch: instances an earlier pass over the same list solved are left alone (1: ok / ok-relaxed, 2: ok only)
  int sweep_k = 0, sweep_phase = 0;  // command sweep: group size; phase 0 = one workgroup per group forms M, 1 = one per instance solves with it (variant = a MODE 1 entry)
  bool list_indefinite = false;  // device-side chain, safe launch: instances ended as HMPC_S_INDEFINITE are appended to the handle's short list
  int reg_step = 0;        // safe pass over an index list: regularisation step 1 / 2 for instances whose Hessian is not positive definite (KernelArgs::reg_step)
};

// the flagged list of the device-side chain: flag_list_cap entries, then REG_LIST_CAP more for the instances the safe launch ends as
// HMPC_S_INDEFINITE; the two counters sit next to each other (one memset clears both)
static int flag_list_cap(const hmpc_handle *h) { return h->max_batch < REPAIR_GRID_CAP ? h->max_batch : REPAIR_GRID_CAP; }

static int launch(hmpc_handle *h, hipStream_t stream, const LaunchOpt &o) {
  int vi = 0;
  const Variant *pv = &pick_variant(h, &vi);
  if (o.variant >= 0) vi = o.variant, pv = &variants()[vi];
  if (o.d_index_list) {  // safe variant: working set as large as the variable count
    if (o.continuation) vi = V2_CONT + (h->setup.horizon > 10 ? 1 : 0);
    else if (o.safe_variant >= 0) vi = o.safe_variant;
    else if (vi == V2_WIDE) vi = V2_WIDE_SAFE;             // ... which for 240 variables only global memory holds
    else if (h->nc == 3) vi = o.ultimate ? V3_SAFE_G : V3_SAFE;
    else vi = (h->setup.horizon <= 10) ? N_FAST : N_FAST + 1;
    pv = &variants()[vi];
  }
  const Variant &v = *pv;
  const int grid_all = o.assemble_only ? 1 : (o.d_index_list ? o.n_list : ((o.sweep_k > 0 && o.sweep_phase == 0) ? h->batch / o.sweep_k : h->batch));
  if (grid_all < 1) return HMPC_OK;
  // EGLOBAL variants keep NMAX (NMAX + 1) / 2 doubles of global scratch per WORKGROUP (231 KB for 240 variables): a host-driven
  // safe pass over thousands of flagged instances goes through the list in chunks that reuse one bounded buffer (stream order
  // keeps the chunks apart).  The device-driven pass (d_list_count) is one launch, capped by its caller.
  const int chunk = (v.qcap == 0 && o.d_index_list && !o.d_list_count && grid_all > EGLOBAL_CHUNK) ? EGLOBAL_CHUNK : grid_all;
  if (v.qcap == 0) {  // EGLOBAL: one slice of packed triangle per workgroup of this launch
    const size_t need = (size_t)chunk * ((size_t)v.nmax * (v.nmax + 1) / 2) * sizeof(double);
    if (need > h->e_bytes) {
      if (h->d_escratch) {
        HIP_TRY(hipStreamSynchronize(stream));  // (an earlier launch on this stream may still be using the old buffer)
        HIP_TRY(hipFree(h->d_escratch));
        h->d_escratch = nullptr, h->e_bytes = 0;
      }
      HIP_TRY(hipMalloc(&h->d_escratch, need));
      h->e_bytes = need;
    }
  }
  // hand-over slots: allocated on the first launch of a variant that saves its state (one slot per instance of the handle).
  // The per-instance slot table is written by EVERY ordinary launch of such a variant (-1 where nothing was saved), also when
  // saving itself is off for the launch, so that a later safe pass never meets an entry of an earlier batch.
  if (o.assemble_only && !v.assemble) return HMPC_E_ARG;
  if ((v.mode == 1) != (o.sweep_k > 0)) return HMPC_E_ARG;
  const bool can_save = v.spill_stride > 0 && !o.assemble_only && !o.d_index_list;
  const bool saves = can_save && h->handover && !h->d_ext_H;
  if (can_save && !h->d_spill_slot) {
    HIP_TRY(hipMalloc(&h->d_spill_slot, (size_t)h->max_batch * sizeof(int)));
    HIP_TRY(hipMemset(h->d_spill_slot, 0xff, (size_t)h->max_batch * sizeof(int)));
  }
  if (saves && (!h->d_spill || h->spill_stride < v.spill_stride)) {
    if (h->d_spill) {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipFree(h->d_spill));
      h->d_spill = nullptr;
    }
    const int cap = h->max_batch < SPILL_SLOT_CAP ? h->max_batch : SPILL_SLOT_CAP;
    HIP_TRY(hipMalloc(&h->d_spill, (size_t)cap * v.spill_stride));
    h->spill_stride = v.spill_stride, h->spill_cap = cap;
  }
  kernel_fn fn = o.assemble_only ? v.assemble : v.solve;
  if (!h->attrs_set[vi]) {
    HIP_TRY(hipFuncSetAttribute((const void *)v.solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.smem));
    if (v.assemble) HIP_TRY(hipFuncSetAttribute((const void *)v.assemble, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.smem));
    h->attrs_set[vi] = true;
  }
  hmpc::KernelArgs a;
  a.records = h->d_records;
  a.stride = (int)h->stride;
  a.batch = h->batch;
  a.horizon = h->setup.horizon;
  a.dt = h->setup.dt;
  a.f_max = h->setup.f_max;
  a.forces = h->d_forces;
  a.status = h->d_status;
  a.x64 = h->d_x64;
  a.obj64 = h->d_obj64;
  a.dbg_index = o.dbg_index;
  a.dbg_f = h->d_dbg_f;
  a.dbg_i = h->d_dbg_i;
  a.prof = h->d_prof;
  a.warm = (o.warm < 0) ? h->warm : o.warm;
  a.index_list = o.d_index_list;
  if (!o.d_index_list && !o.assemble_only && o.longest_first) a.index_list = h->d_order;
  a.wset = (h->tick_warm && !o.assemble_only && o.carry_wset) ? h->d_wset : nullptr;
  a.flagged = h->d_flagged;
  a.wset_shift = h->tick_shift;
  a.relax = o.relax;
  a.flag_list = o.record_flagged ? h->d_flag_list : nullptr;
  a.flag_count = o.record_flagged ? h->d_flag_count : nullptr;
  a.flag_cap = o.record_flagged ? flag_list_cap(h) : 0;
  a.list_count = o.d_list_count;
  a.ext_H = h->d_ext_H, a.ext_g = h->d_ext_g, a.ext_Fc = h->d_ext_Fc, a.ext_ld = h->ext_ld;
  a.iter_cap = h->iter_cap;
  a.cls = (o.cls_hi >= 0) ? h->d_cls : nullptr;
  a.cls_lo = o.cls_lo, a.cls_hi = o.cls_hi;
  a.e_scratch = h->d_escratch;
  a.spill = nullptr, a.spill_stride = 0, a.spill_cap = 0, a.spill_slot = nullptr, a.resume = 0;
  if (can_save) {
    a.spill_slot = h->d_spill_slot;
    if (saves) a.spill = h->d_spill, a.spill_stride = h->spill_stride, a.spill_cap = h->spill_cap;
  } else if (o.d_index_list && o.resume && v.resumes && h->handover && h->d_spill && h->d_spill_slot && o.relax == 0.0 && !h->d_ext_H) {
    a.spill = h->d_spill, a.spill_stride = h->spill_stride, a.spill_cap = h->spill_cap, a.spill_slot = h->d_spill_slot;
    a.resume = o.continuation ? 2 : 1;
  } else if (o.continuation) {
    return HMPC_OK;  // nothing was handed over (hand-over off / no slots): the continuation pass has nothing to do
  }
  a.skip_ok = o.skip_ok;
  if (o.d_index_list && !h->d_reg_rho) HIP_TRY(hipMalloc(&h->d_reg_rho, (size_t)h->max_batch * sizeof(double)));  // (safe variants: where a pivot that is not positive is left)
  a.reg_step = o.reg_step, a.reg_rho = h->d_reg_rho;
  a.reg_list = nullptr, a.reg_count = nullptr, a.reg_cap = 0;
  if (o.list_indefinite && h->d_flag_list && h->d_flag_count)
    a.reg_list = h->d_flag_list + flag_list_cap(h), a.reg_count = h->d_flag_count + 1, a.reg_cap = REG_LIST_CAP;
  a.sweep_k = o.sweep_k > 0 ? o.sweep_k : 1, a.sweep_phase = o.sweep_phase, a.sweep_m = h->d_sweep_m;
  a.inv_mass = 1.0f / h->params.mass;  // (binary32 division, correctly rounded: the value the reference's 1.f / 9.f folds to for the default)
  a.Ib[0] = h->params.inertia[0], a.Ib[1] = h->params.inertia[1], a.Ib[2] = h->params.inertia[2];
  a.mu = h->params.mu, a.lt = h->params.lt, a.lh = h->params.lh, a.gravity = h->params.gravity;
  a.mu_inst = h->d_mu_inst;
  for (int off = 0; off < grid_all; off += chunk) {
    const int grid = (grid_all - off < chunk) ? grid_all - off : chunk;
    if (off > 0) a.index_list = o.d_index_list + off;  // (only list launches are ever chunked)
    hipLaunchKernelGGL(fn, dim3(grid), dim3(v.nt), v.smem, stream, a);
    HIP_TRY(hipGetLastError());
  }
  return HMPC_OK;
}

// The safe pass over a list of flagged instances.  Where the host knows the batch's widest reduced QP (or the family has one safe
// variant) that is one launch.  A two-contact batch at h > 10 whose sizes only the DEVICE knows (records built on the device or
// handed in by pointer: max_stance < 0) may hold both <= 120-variable instances and double-support ones with up to 240: the list
// is then run twice, once per safe variant, each workgroup leaving at once unless its instance's size class belongs to the
// variant -- a wide instance must never reach the 120-variable kernel (it would end as HMPC_S_TOO_LARGE with zero forces, which
// nothing re-solves).
static int launch_safe(hmpc_handle *h, hipStream_t stream, LaunchOpt s) {
  if (h->nc == 2 && h->max_stance < 0 && h->setup.horizon > 10 && h->d_cls) {
    s.safe_variant = N_FAST + 1, s.cls_lo = 0, s.cls_hi = 20;
    int rc = launch(h, stream, s);
    if (rc != HMPC_OK) return rc;
    s.safe_variant = V2_WIDE_SAFE, s.cls_lo = 21, s.cls_hi = 255;
    if (s.d_list_count && s.n_list > REPAIR_GRID_CAP_WIDE) s.n_list = REPAIR_GRID_CAP_WIDE;  // (one launch: bounded scratch)
    return launch(h, stream, s);
  }
  return launch(h, stream, s);
}

// Device-side chain, behind the safe launch: instances whose Hessian is not positive definite (that launch found a sweep pivot <= 0,
// ended them as HMPC_S_INDEFINITE and listed them) get the reference's two regularised QPs (KernelArgs::reg_step; hmpc_resolve_failed
// runs the same two launches from the status words).  The list is a short one of its own (REG_LIST_CAP entries, its counter next to
// the flagged counter), so the two launches are a few hundred workgroups that leave at once when it is empty.
static int enqueue_reg_steps(hmpc_handle *h, hipStream_t stream, LaunchOpt s) {
  // Two launches = ~4 us of dispatch latency per solve even when their list is empty (scripts/dev/chain_overhead.py: the whole chain
  // 11 -> 15 us at b8192, 6 -> 10 us at b1024), so only where such Hessians occur: horizons beyond 10 steps (binary32 round-off in H
  // grows with the horizon; 107 of 4 096 double-support h = 20 instances at 10x the input ranges, none in any h <= 10 stress row up to
  // 10x -- 20 000 instances).  A shorter-horizon handle would leave such an instance HMPC_S_INDEFINITE for hmpc_resolve_failed.
  if (h->setup.horizon <= DEVICE_REG_MIN_HORIZON) return HMPC_OK;
  s.d_index_list = h->d_flag_list + flag_list_cap(h);
  s.d_list_count = h->d_flag_count + 1;
  s.n_list = h->batch < REG_LIST_CAP ? h->batch : REG_LIST_CAP;
  s.relax = 0.0, s.warm = 1, s.skip_ok = 0, s.list_indefinite = false;
  int rc = HMPC_OK;
  for (int step = 1; step <= 2 && rc == HMPC_OK; ++step) {
    s.reg_step = step;
    rc = launch_safe(h, stream, s);
  }
  return rc;
}

// One solve of the current batch, enqueued on `stream` -- what hmpc_solve does and what hmpc_time_solve times:
//  * widest reduced QP known (host-uploaded records, or hmpc_set_max_reduced_vars >= 0): one launch of the variant
//    that holds it;
//  * unknown (records built on the device or handed in by device pointer; two contacts): the instances' size classes are
//    on the device (from the record builder, else counted here from the gait bytes) and EVERY variant of the family is
//    launched over the whole batch -- a workgroup whose instance belongs to another variant leaves at once -- so that a
//    walking sweep built on the device runs on the 60-variable kernel without the host ever seeing a gait table;
//  * device repair: the fast launches list what they flag, the safe variant follows over that list (trimmed on the
//    device by the counter: workgroups beyond it leave at once).  One stream per handle at a time: the list and its
//    counter belong to the handle, two solves of one handle in flight on two streams would race on them.
static int enqueue_solve(hmpc_handle *h, hipStream_t stream, bool carry_wset) {
  const bool repair = h->device_repair != 0;
  if (repair) HIP_TRY(hipMemsetAsync(h->d_flag_count, 0, 2 * sizeof(unsigned int), stream));
  // longest-first dispatch: only where the tail of a launch matters (small and medium batches).  Keyed by the iteration counts
  // of the previous solve when that was of a batch of this size (the caller's contract: instance i of this tick is instance i
  // of the last one); otherwise -- a cold handle, another batch size, mode 2 -- by the cost predicted from the records themselves
  // (predicted_cost_bucket: no previous solve needed)
  h->order_valid = false;
  // (... and not for batches known to hold single-support QPs only: those solve in one or two iterations, there is nothing to
  //  sort and the extra launch costs a walking batch 1-4 %)
  const bool small_qps_only = h->nc == 2 && h->max_stance >= 0 && h->max_stance <= 60;
  if (h->dispatch_order != 0 && h->d_order && h->batch > DISPATCH_ORDER_MIN_BATCH && h->batch <= DISPATCH_ORDER_MAX_BATCH &&
      !small_qps_only && !h->d_ext_H) {
    const bool from_previous = h->dispatch_order == 1 && h->order_batch == h->batch;
    if (!from_previous) {
      if (!h->d_keys) HIP_TRY(hipMalloc(&h->d_keys, (size_t)h->max_batch));
      hipLaunchKernelGGL(hmpc::predicted_cost_kernel, dim3((h->batch + 255) / 256), dim3(256), 0, stream, h->d_records, (int)h->stride,
                         h->batch, h->setup.horizon, h->nc, h->d_keys);
      HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(hmpc::dispatch_order_kernel, dim3(1), dim3(1024), 0, stream, h->d_status, h->batch, h->d_order,
                       from_previous ? (const unsigned char *)nullptr : h->d_keys);
    HIP_TRY(hipGetLastError());
    h->order_valid = true;
  }
  h->order_batch = h->batch;
  LaunchOpt o;
  o.carry_wset = carry_wset;
  o.record_flagged = repair;
  o.longest_first = h->order_valid;
  int rc = HMPC_OK;
  if (h->nc == 2 && h->max_stance < 0 && h->d_cls) {
    if (!h->cls_valid) {
      hipLaunchKernelGGL(hmpc::classify_records_kernel, dim3((h->batch + 255) / 256), dim3(256), 0, stream, h->d_records,
                         (int)h->stride, h->batch, h->setup.horizon, h->setup.f_max, h->d_cls);
      HIP_TRY(hipGetLastError());
    }
    const int hz = h->setup.horizon;
    // variants(): [0] <60,10,128>, [1] <120,10,256>, [2] <60,20,128>, [3] <120,20,256>, V2_WIDE <240,20,512>
    o.variant = (hz <= 10) ? 0 : 2, o.cls_lo = 0, o.cls_hi = 10;
    rc = launch(h, stream, o);
    if (rc != HMPC_OK) return rc;
    o.variant = (hz <= 10) ? 1 : 3, o.cls_lo = 11, o.cls_hi = (hz <= 10) ? 255 : 20;  // (> 20 at h <= 10 cannot occur)
    rc = launch(h, stream, o);
    if (rc != HMPC_OK) return rc;
    if (hz > 10) {
      o.variant = V2_WIDE, o.cls_lo = 21, o.cls_hi = 255;
      rc = launch(h, stream, o);
    }
  } else {
    rc = launch(h, stream, o);
  }
  if (rc != HMPC_OK || !repair) return rc;
  LaunchOpt s;
  s.d_index_list = h->d_flag_list;
  s.n_list = h->batch < REPAIR_GRID_CAP ? h->batch : REPAIR_GRID_CAP;
  {
    int vsel = 0;
    (void)pick_variant(h, &vsel);
    if (vsel == V2_WIDE && s.n_list > REPAIR_GRID_CAP_WIDE) s.n_list = REPAIR_GRID_CAP_WIDE;  // (231 KB of global scratch per workgroup)
  }
  s.warm = 1;  // (the block start: what differs from the fast variant is capacity, periodic rebuild of E, in-kernel perturbation)
  s.carry_wset = carry_wset;
  s.d_list_count = h->d_flag_count;
  // (1) continuation: instances whose working set outgrew the fast variant go on, from the state it handed over, on the variant
  //     with 96 rows and block rounds of its own (two per CU); (2) the safe variant, cold, for everything still flagged -- what
  //     the fast variant flagged for other reasons, and what outgrew the continuation variant as well
  if (h->nc == 2 && h->handover && h->d_spill) {
    LaunchOpt c = s;
    c.continuation = true, c.resume = true;
    rc = launch(h, stream, c);
    if (rc != HMPC_OK) return rc;
    s.skip_ok = 1;
    static const bool cont_only = getenv("HMPC_DEBUG_CONT_ONLY") && getenv("HMPC_DEBUG_CONT_ONLY")[0] == '1';  // developer switch: what the continuation pass alone leaves
    if (cont_only) return HMPC_OK;
  }
  if (h->device_repair == 2) return HMPC_OK;  // continuation only: the safe pass is left to hmpc_resolve_failed / hmpc_download
  // The safe pass over what is still flagged: cold, with every bound moved outward by SAFE_PASS_RELAX (1 + frac(0.618 row)) from the start.
  // What reaches it are the instances that cycle at degenerate vertices (the continuation's budget, a KKT check): perturbed, they
  // take ~150 iterations instead of up to 480, and the kernel's epilogue re-solves on the final working set with the EXACT bounds and
  // repeats the exact KKT check -- measured at 6x the input ranges: 8.8 -> 6.9 ms for the whole chain AND 3 -> 0 of 8 192 left flagged
  // (10x: 19.5 -> 14.6 ms, 10 -> 1); every one of them HMPC_S_OK, exact (profiles/r06/range_scale.txt).
  s.relax = SAFE_PASS_RELAX, s.warm = 0;
  {
    static const char *dbg_relax = getenv("HMPC_DEBUG_SAFE_RELAX");  // developer A/B: another perturbation (0 = the exact, warm pass of before)
    if (dbg_relax && *dbg_relax) s.relax = atof(dbg_relax), s.warm = (s.relax == 0.0) ? 1 : 0;
  }
  s.list_indefinite = true;
  rc = launch_safe(h, stream, s);
  if (rc != HMPC_OK) return rc;
  return enqueue_reg_steps(h, stream, s);
}

extern "C" {

const char *hmpc_last_hip_error(void) { return g_hip_err.c_str(); }
const char *hmpc_version(void) { return "hector_mpc_hip 0.1 (gfx950)"; }

size_t hmpc_record_stride(int horizon) { return record_stride(horizon); }
size_t hmpc_record_stride_ex(int horizon, int n_contacts) { return record_stride(horizon, n_contacts == 3 ? 3 : 2); }

int hmpc_pack_record_ex(void *record, int horizon, int n_contacts, const double *p, const double *v, const double *q,
                        const double *w, const double *r, const double *joint_angles, double yaw, const double *weights,
                        const double *state_trajectory, const double *Alpha_K, const int *gait, const double *Rhand,
                        double f_max_hand) {
  if (n_contacts == 2)
    return hmpc_pack_record(record, horizon, p, v, q, w, r, joint_angles, yaw, weights, state_trajectory, Alpha_K, gait);
  if (n_contacts != 3) return HMPC_E_ARG;
  if (!record || !p || !v || !q || !w || !r || !joint_angles || !weights || !state_trajectory || !Alpha_K || !gait || !Rhand)
    return HMPC_E_ARG;
  if (horizon < 1 || horizon > 10) return HMPC_E_HORIZON;
  memset(record, 0, record_stride(horizon, 3));
  float *f = (float *)record;
  for (int i = 0; i < 3; ++i) f[0 + i] = (float)p[i], f[3 + i] = (float)v[i], f[10 + i] = (float)w[i];
  for (int i = 0; i < 4; ++i) f[6 + i] = (float)q[i];
  for (int i = 0; i < 9; ++i) f[13 + i] = (float)r[i], f[63 + i] = (float)Rhand[i];
  for (int i = 0; i < 10; ++i) f[22 + i] = (float)joint_angles[i];
  f[32] = (float)yaw;
  for (int i = 0; i < 12; ++i) f[33 + i] = (float)weights[i];
  for (int i = 0; i < 18; ++i) f[45 + i] = (float)Alpha_K[i];
  f[72] = (float)f_max_hand;
  for (int i = 0; i < 12 * horizon; ++i) f[73 + i] = (float)state_trajectory[i];
  unsigned char *g = (unsigned char *)record + 4 * (73 + 12 * horizon);
  for (int i = 0; i < 3 * horizon; ++i) g[i] = (unsigned char)gait[i];
  return HMPC_OK;
}

int hmpc_pack_record(void *record, int horizon, const double *p, const double *v, const double *q, const double *w,
                     const double *r, const double *joint_angles, double yaw, const double *weights,
                     const double *state_trajectory, const double *Alpha_K, const int *gait) {
  if (!record || !p || !v || !q || !w || !r || !joint_angles || !weights || !state_trajectory || !Alpha_K || !gait)
    return HMPC_E_ARG;
  if (horizon < 1 || horizon > HMPC_MAX_HORIZON) return HMPC_E_HORIZON;
  memset(record, 0, record_stride(horizon));
  float *f = (float *)record;
  for (int i = 0; i < 3; ++i) f[0 + i] = (float)p[i], f[3 + i] = (float)v[i], f[10 + i] = (float)w[i];
  for (int i = 0; i < 4; ++i) f[6 + i] = (float)q[i];
  for (int i = 0; i < 6; ++i) f[13 + i] = (float)r[i];
  for (int i = 0; i < 10; ++i) f[19 + i] = (float)joint_angles[i];
  f[29] = (float)yaw;
  for (int i = 0; i < 12; ++i) f[30 + i] = (float)weights[i], f[42 + i] = (float)Alpha_K[i];
  for (int i = 0; i < 12 * horizon; ++i) f[54 + i] = (float)state_trajectory[i];
  unsigned char *g = (unsigned char *)record + 4 * (54 + 12 * horizon);
  for (int i = 0; i < 2 * horizon; ++i) g[i] = (unsigned char)gait[i];
  return HMPC_OK;
}

int hmpc_create(hmpc_handle **out, const struct problem_setup *setup, int max_batch, int device) {
  return hmpc_create_ex(out, setup, max_batch, device, 2);
}

int hmpc_contacts(const hmpc_handle *h) { return h ? h->nc : HMPC_E_ARG; }

int hmpc_create_ex(hmpc_handle **out, const struct problem_setup *setup, int max_batch, int device, int n_contacts) {
  if (!out || !setup || max_batch < 1 || (n_contacts != 2 && n_contacts != 3)) return HMPC_E_ARG;
  if (setup->horizon < 1 || setup->horizon > (n_contacts == 3 ? 10 : HMPC_MAX_HORIZON)) return HMPC_E_HORIZON;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device >= ndev) {
    g_hip_err = "no HIP device visible (libhector_mpc_hip has no CPU fallback)";
    return HMPC_E_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(device));
  hmpc_handle *h = new (std::nothrow) hmpc_handle();
  if (!h) return HMPC_E_ARG;
  memset(h, 0, sizeof(*h));
  h->setup = *setup;
  h->max_batch = max_batch;
  h->device = device;
  h->nc = n_contacts;
  h->stride = record_stride(setup->horizon, n_contacts);
  h->max_stance = -1;
  h->warm = 1;
  h->auto_resolve = 1;
  h->handover = 1;
  hmpc_default_params(&h->params);
  const size_t nf = (size_t)max_batch * 6 * n_contacts * setup->horizon;
  if (hipMalloc(&h->d_records_own, (size_t)max_batch * h->stride) != hipSuccess ||
      hipMalloc(&h->d_forces_own, nf * sizeof(float)) != hipSuccess ||
      hipMalloc(&h->d_status_own, (size_t)max_batch * sizeof(uint32_t)) != hipSuccess) {
    g_hip_err = "hipMalloc failed in hmpc_create";
    hmpc_destroy(h);
    return HMPC_E_HIP;
  }
  if (hipMalloc(&h->d_flagged, sizeof(unsigned int)) != hipSuccess ||
      hipMemset(h->d_flagged, 0, sizeof(unsigned int)) != hipSuccess) {
    g_hip_err = "hipMalloc failed in hmpc_create";
    hmpc_destroy(h);
    return HMPC_E_HIP;
  }
  if (n_contacts == 2 && (hipMalloc(&h->d_cls, (size_t)max_batch) != hipSuccess || hipMemset(h->d_cls, 0, (size_t)max_batch) != hipSuccess)) {
    g_hip_err = "hipMalloc failed in hmpc_create";
    hmpc_destroy(h);
    return HMPC_E_HIP;
  }
  h->dispatch_order = 1;
  if (max_batch > DISPATCH_ORDER_MIN_BATCH && (hipMalloc(&h->d_order, (size_t)max_batch * sizeof(int)) != hipSuccess ||
                                               hipMalloc(&h->d_keys, (size_t)max_batch) != hipSuccess)) {
    g_hip_err = "hipMalloc failed in hmpc_create";
    hmpc_destroy(h);
    return HMPC_E_HIP;
  }
  h->d_records = h->d_records_own;
  h->d_forces = h->d_forces_own;
  h->d_status = h->d_status_own;
  *out = h;
  return HMPC_OK;
}

int hmpc_destroy(hmpc_handle *h) {
  if (!h) return HMPC_E_ARG;
  hipSetDevice(h->device);
  if (h->d_records_own) hipFree(h->d_records_own);
  if (h->d_forces_own) hipFree(h->d_forces_own);
  if (h->d_status_own) hipFree(h->d_status_own);
  if (h->d_x64) hipFree(h->d_x64);
  if (h->d_obj64) hipFree(h->d_obj64);
  if (h->d_dbg_f) hipFree(h->d_dbg_f);
  if (h->d_dbg_i) hipFree(h->d_dbg_i);
  if (h->d_prof) hipFree(h->d_prof);
  if (h->d_wset) hipFree(h->d_wset);
  if (h->d_keys) hipFree(h->d_keys);
  if (h->d_scratch) hipFree(h->d_scratch);
  if (h->d_flagged) hipFree(h->d_flagged);
  if (h->d_flag_list) hipFree(h->d_flag_list);
  if (h->d_flag_count) hipFree(h->d_flag_count);
  if (h->d_cls) hipFree(h->d_cls);
  if (h->d_order) hipFree(h->d_order);
  if (h->d_escratch) hipFree(h->d_escratch);
  if (h->d_sweep_m) hipFree(h->d_sweep_m);
  if (h->d_reg_rho) hipFree(h->d_reg_rho);
  if (h->d_spill) hipFree(h->d_spill);
  if (h->d_spill_slot) hipFree(h->d_spill_slot);
  delete h;
  return HMPC_OK;
}

// pitch = bytes between consecutive records of the batch in host memory (h->stride: a packed array)
static int upload_common(hmpc_handle *h, const void *host_records, int batch, bool async, hipStream_t stream, size_t pitch = 0) {
  if (!h || !host_records || batch < 0) return HMPC_E_ARG;
  if (batch > h->max_batch) return HMPC_E_BATCH;
  if (pitch == 0) pitch = h->stride;
  if (pitch < h->stride || pitch % 4 != 0) return HMPC_E_ARG;
  HIP_TRY(hipSetDevice(h->device));
  if (pitch != h->stride) {
    if (batch > 0)
      HIP_TRY(hipMemcpy2DAsync(h->d_records_own, h->stride, host_records, pitch, h->stride, (size_t)batch, hipMemcpyHostToDevice, stream));
  } else if (async)
    HIP_TRY(hipMemcpyAsync(h->d_records_own, host_records, (size_t)batch * h->stride, hipMemcpyHostToDevice, stream));
  else
    HIP_TRY(hipMemcpy(h->d_records_own, host_records, (size_t)batch * h->stride, hipMemcpyHostToDevice));
  h->d_records = h->d_records_own;
  h->batch = batch;
  h->cls_valid = 0;
  // host-side scan of the gait tables: the widest reduced QP in the batch picks the kernel variant (LDS footprint)
  const int hz = h->setup.horizon;
  int mx = 0;
  const unsigned char *rec = (const unsigned char *)host_records;
  const int nc = h->nc, nfix = fixed_floats(nc);
  for (int b = 0; b < batch; ++b) {
    const unsigned char *rb = rec + (size_t)b * pitch;
    const unsigned char *g = rb + 4 * (nfix + 12 * hz);
    float hand_cap = 0.f;
    if (nc == 3) memcpy(&hand_cap, rb + 4 * 72, 4);
    int cnt = 0;
    for (int i = 0; i < nc * hz; ++i) {
      float ub = ((i % nc) == 2 ? hand_cap : h->setup.f_max) * (float)g[i];
      if (!(ub < 0.0001 && ub > -.0001)) ++cnt;
    }
    if (cnt > mx) mx = cnt;
  }
  h->max_stance = 6 * mx;
  return HMPC_OK;
}

int hmpc_upload_records(hmpc_handle *h, const void *host_records, int batch) {
  return upload_common(h, host_records, batch, false, nullptr);
}

int hmpc_upload_records_async(hmpc_handle *h, const void *host_records, int batch, void *stream) {
  return upload_common(h, host_records, batch, true, (hipStream_t)stream);
}

int hmpc_upload_records_strided_async(hmpc_handle *h, const void *host_records, int batch, size_t pitch_bytes, void *stream) {
  return upload_common(h, host_records, batch, true, (hipStream_t)stream, pitch_bytes);
}

int hmpc_download_async(hmpc_handle *h, float *forces, uint32_t *status, void *stream) {
  if (!h) return HMPC_E_ARG;
  HIP_TRY(hipSetDevice(h->device));
  const size_t nf = (size_t)h->batch * 6 * h->nc * h->setup.horizon;
  if (forces && nf)
    HIP_TRY(hipMemcpyAsync(forces, h->d_forces, nf * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
  if (status && h->batch)
    HIP_TRY(hipMemcpyAsync(status, h->d_status, (size_t)h->batch * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  return HMPC_OK;
}

int hmpc_set_device_records(hmpc_handle *h, const void *device_records, int batch) {
  if (!h || !device_records || batch < 0) return HMPC_E_ARG;
  if (batch > h->max_batch) return HMPC_E_BATCH;
  h->d_records = (const unsigned char *)device_records;
  h->batch = batch;
  h->max_stance = -1;  // unknown: hmpc_solve counts the size classes on the device (or hmpc_set_max_reduced_vars tells)
  h->cls_valid = 0;
  return HMPC_OK;
}

int hmpc_set_max_reduced_vars(hmpc_handle *h, int n_reduced) {
  if (!h) return HMPC_E_ARG;
  h->max_stance = n_reduced;
  return HMPC_OK;
}

int hmpc_set_dispatch_order(hmpc_handle *h, int mode) {
  if (!h || (mode != 0 && mode != 1 && mode != 2)) return HMPC_E_ARG;
  if (mode != 0 && !h->d_order && h->max_batch > DISPATCH_ORDER_MIN_BATCH) {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMalloc(&h->d_order, (size_t)h->max_batch * sizeof(int)));
  }
  if (mode != 0 && !h->d_keys && h->max_batch > DISPATCH_ORDER_MIN_BATCH) {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMalloc(&h->d_keys, (size_t)h->max_batch));
  }
  h->dispatch_order = mode;
  h->order_batch = 0;  // the next solve is ordered by the predictor (mode 1, 2) and leaves the iteration counts the one after it sorts by (mode 1)
  h->order_valid = false;
  return HMPC_OK;
}

int hmpc_set_max_iterations(hmpc_handle *h, int max_iter) {
  if (!h || max_iter < 0) return HMPC_E_ARG;
  h->iter_cap = max_iter;
  return HMPC_OK;
}

void hmpc_default_params(struct hmpc_params *p) {
  if (!p) return;
  p->mass = 9.0f;                                                   // SolverMPC.cpp:423
  p->inertia[0] = 0.5413f, p->inertia[1] = 0.5200f, p->inertia[2] = 0.0691f;  // RobotState.cpp:45
  p->mu = 2.0f, p->lt = 0.09f, p->lh = 0.06f;                       // SolverMPC.cpp:488-490
  p->gravity = 9.81f;                                               // SolverMPC.cpp:420
}

static bool params_ok(const hmpc_params &p) {
  auto pos = [](float v) { return v > 0.0f && v < 1e30f; };
  return pos(p.mass) && pos(p.inertia[0]) && pos(p.inertia[1]) && pos(p.inertia[2]) && pos(p.mu) && p.lt == p.lt && p.lh == p.lh &&
         p.gravity == p.gravity && fabsf(p.lt) < 1e30f && fabsf(p.lh) < 1e30f && fabsf(p.gravity) < 1e30f;
}

int hmpc_set_params(hmpc_handle *h, const struct hmpc_params *p) {
  if (!h) return HMPC_E_ARG;
  if (!p) {
    hmpc_default_params(&h->params);
    return HMPC_OK;
  }
  if (!params_ok(*p)) return HMPC_E_ARG;
  h->params = *p;
  return HMPC_OK;
}

int hmpc_set_instance_mu(hmpc_handle *h, const float *device_mu) {
  if (!h) return HMPC_E_ARG;
  h->d_mu_inst = device_mu;
  return HMPC_OK;
}

int hmpc_get_params(const hmpc_handle *h, struct hmpc_params *p) {
  if (!h || !p) return HMPC_E_ARG;
  *p = h->params;
  return HMPC_OK;
}

int hmpc_set_handover(hmpc_handle *h, int on) {
  if (!h) return HMPC_E_ARG;
  h->handover = on ? 1 : 0;
  return HMPC_OK;
}

int hmpc_set_auto_resolve(hmpc_handle *h, int on) {
  if (!h) return HMPC_E_ARG;
  h->auto_resolve = on ? 1 : 0;
  return HMPC_OK;
}

int hmpc_set_warm_start(hmpc_handle *h, int on) {
  if (!h) return HMPC_E_ARG;
  h->warm = on ? 1 : 0;
  return HMPC_OK;
}

int hmpc_set_tick_warm_start(hmpc_handle *h, int on, int horizon_shift) {
  if (!h || horizon_shift < 0) return HMPC_E_ARG;
  HIP_TRY(hipSetDevice(h->device));
  if (on && !h->d_wset) {
    const size_t nb = (size_t)h->max_batch * 8 * h->nc * h->setup.horizon;
    HIP_TRY(hipMalloc(&h->d_wset, nb));
    HIP_TRY(hipMemset(h->d_wset, 0, nb));
  }
  h->tick_warm = on ? 1 : 0;
  h->tick_shift = horizon_shift;
  return HMPC_OK;
}

int hmpc_reset_tick_warm_start(hmpc_handle *h) {
  if (!h) return HMPC_E_ARG;
  if (!h->d_wset) return HMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->last_stream));
  HIP_TRY(hipMemset(h->d_wset, 0, (size_t)h->max_batch * 8 * h->nc * h->setup.horizon));
  return HMPC_OK;
}

int hmpc_set_device_outputs(hmpc_handle *h, float *device_forces, uint32_t *device_status) {
  if (!h) return HMPC_E_ARG;
  h->d_forces = device_forces ? device_forces : h->d_forces_own;
  h->d_status = device_status ? device_status : h->d_status_own;
  return HMPC_OK;
}

int hmpc_solve(hmpc_handle *h, void *stream) {
  if (!h) return HMPC_E_ARG;
  if (h->batch == 0) return HMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  h->last_stream = (hipStream_t)stream;
  return enqueue_solve(h, (hipStream_t)stream, /*carry_wset=*/true);
}

// Command sweeps (MODE 1 kernels): phase 0 forms every group's M = H^-1 once (one workgroup per group) and leaves it in HBM, phase 1
// solves every instance with its group's M (one workgroup per instance, stages H and S skipped).
int hmpc_solve_command_sweep(hmpc_handle *h, int group_size, void *stream) {
  if (!h || group_size < 1) return HMPC_E_ARG;
  if (h->nc != 2 || h->setup.horizon > 10) return HMPC_E_ARG;  // (the shapes the sweep kernels are built for)
  if (h->batch == 0) return HMPC_OK;
  if (h->batch % group_size != 0) return HMPC_E_ARG;
  if (group_size == 1) return hmpc_solve(h, stream);
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  h->last_stream = st;
  const bool small = h->max_stance >= 0 && h->max_stance <= 60;  // single-support sweeps: the 60-variable kernel (six workgroups per CU)
  const int vi = small ? V2_SWEEP_60 : V2_SWEEP_120;
  const int groups = h->batch / group_size;
  const size_t need = (size_t)groups * 36 * (size_t)variants()[vi].nt * sizeof(double);
  if (need > h->sweep_m_bytes) {
    if (h->d_sweep_m) {
      HIP_TRY(hipStreamSynchronize(st));
      HIP_TRY(hipFree(h->d_sweep_m));
      h->d_sweep_m = nullptr, h->sweep_m_bytes = 0;
    }
    HIP_TRY(hipMalloc(&h->d_sweep_m, need));
    h->sweep_m_bytes = need;
  }
  const bool repair = h->device_repair != 0;
  if (repair) HIP_TRY(hipMemsetAsync(h->d_flag_count, 0, 2 * sizeof(unsigned int), st));
  h->order_valid = false, h->order_batch = 0;  // (natural order inside a sweep; the next ordinary solve starts from the predictor)
  LaunchOpt o;
  o.variant = vi, o.sweep_k = group_size, o.sweep_phase = 0;
  int rc = launch(h, st, o);
  if (rc != HMPC_OK) return rc;
  o.sweep_phase = 1;
  o.record_flagged = repair;
  rc = launch(h, st, o);
  if (rc != HMPC_OK || !repair) return rc;
  // whatever a sweep flags is repaired as an independent instance (its record is complete): the cold safe pass
  LaunchOpt s;
  s.d_index_list = h->d_flag_list;
  s.n_list = h->batch < REPAIR_GRID_CAP ? h->batch : REPAIR_GRID_CAP;
  s.warm = 0;
  s.d_list_count = h->d_flag_count;
  s.list_indefinite = true;
  rc = launch_safe(h, st, s);
  if (rc != HMPC_OK) return rc;
  return enqueue_reg_steps(h, st, s);
}

int hmpc_set_device_repair(hmpc_handle *h, int on) {
  if (!h) return HMPC_E_ARG;
  HIP_TRY(hipSetDevice(h->device));
  if (on && (!h->d_flag_list || !h->d_flag_count)) {
    // both buffers or neither: they are committed to the handle only once both exist and are cleared
    const int cap = flag_list_cap(h);
    int *list = nullptr;
    unsigned int *count = nullptr;
    if (hipMalloc(&list, (size_t)(cap + REG_LIST_CAP) * sizeof(int)) != hipSuccess || hipMalloc(&count, 2 * sizeof(unsigned int)) != hipSuccess ||
        hipMemset(list, 0, (size_t)(cap + REG_LIST_CAP) * sizeof(int)) != hipSuccess || hipMemset(count, 0, 2 * sizeof(unsigned int)) != hipSuccess) {
      if (list) (void)hipFree(list);
      if (count) (void)hipFree(count);
      g_hip_err = "hipMalloc failed in hmpc_set_device_repair";
      return HMPC_E_HIP;
    }
    if (h->d_flag_list) (void)hipFree(h->d_flag_list);
    if (h->d_flag_count) (void)hipFree(h->d_flag_count);
    h->d_flag_list = list, h->d_flag_count = count;
  }
  h->device_repair = (on == 2) ? 2 : (on ? 1 : 0);
  return HMPC_OK;
}

// an HMPC_S_MAXITER status is the CALLER's answer (not re-solved) exactly when the kernel says so: the caller's cap was the
// bound in force, i.e. the iteration count reached it (hmpc_kernel.h: `capped`; the variants' own bounds are 4 m + 16 and up)
static bool capped_by_caller(const hmpc_handle *h, uint32_t status) {
  return h->iter_cap > 0 && (int)HMPC_STATUS_ITERS(status) >= h->iter_cap && (int)HMPC_STATUS_ITERS(status) <= h->iter_cap + 1;
}

int hmpc_resolve_failed(hmpc_handle *h, int *n_resolved) {
  if (!h) return HMPC_E_ARG;
  if (n_resolved) *n_resolved = 0;
  if (h->batch == 0) return HMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->last_stream));
  std::vector<uint32_t> st(h->batch);
  HIP_TRY(hipMemcpy(st.data(), h->d_status, (size_t)h->batch * sizeof(uint32_t), hipMemcpyDeviceToHost));
  std::vector<int> idx;
  for (int i = 0; i < h->batch; ++i) {
    const uint32_t c = HMPC_STATUS_CODE(st[i]);
    // (a solve that ran into the caller's own iteration cap, hmpc_set_max_iterations, is the caller's answer: not re-solved --
    //  the kernel's rule: the cap counts only when it is below the variant's own bound, i.e. when the iteration count of the
    //  status word reached it; a solve that hit the VARIANT's bound under a generous cap is re-solved like any other)
    if (c == HMPC_S_WORKSET || (c == HMPC_S_MAXITER && !capped_by_caller(h, st[i])) || c == HMPC_S_INFEASIBLE || c == HMPC_S_KKT ||
        c == HMPC_S_INDEFINITE)
      idx.push_back(i);
  }
  if (idx.empty()) return HMPC_OK;
  int *d_idx = nullptr;  // lives in the handle's scratch: nothing to free on the error paths below
  {
    void *sp = nullptr;
    const int rc = scratch(h, idx.size() * sizeof(int), &sp);
    if (rc != HMPC_OK) return rc;
    d_idx = (int *)sp;
  }
  HIP_TRY(hipMemcpy(d_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
  // (launch parameters; the handle's own warm-start setting is not touched)
  LaunchOpt so;
  so.d_index_list = d_idx, so.n_list = (int)idx.size(), so.warm = 1;  // (first pass: with the block start; the perturbed passes below start cold)
  int rc = HMPC_OK;
  if (h->nc == 2 && h->handover && h->d_spill) {
    // continuation first (see enqueue_solve): instances with a hand-over slot go on where the fast variant stopped
    LaunchOpt c = so;
    c.continuation = true, c.resume = true;
    rc = launch(h, h->last_stream, c);
    if (rc != HMPC_OK) return rc;
    so.skip_ok = 1;
  }
  // (the same launch as the device-side chain's: cold, bounds perturbed by SAFE_PASS_RELAX, exact re-solve at its end -- see enqueue_solve)
  so.relax = SAFE_PASS_RELAX, so.warm = 0;
  rc = launch_safe(h, h->last_stream, so);
  if (rc != HMPC_OK) return rc;
  // ... then, for what is still flagged, the exact pass with the block start (the first safe pass of rounds 4-6a)
  so.relax = 0.0, so.warm = 1, so.skip_ok = 2;  // (an answer that is only ok-relaxed gets the exact attempt as well)
  rc = launch_safe(h, h->last_stream, so);
  so.skip_ok = 0;
  if (rc != HMPC_OK) return rc;
  HIP_TRY(hipStreamSynchronize(h->last_stream));
  if (n_resolved) *n_resolved = (int)idx.size();
  {
    // A Hessian that is not positive definite (found by the safe variants' sweeps: HMPC_S_INDEFINITE; the fast variants diverge on it
    // and flag it through their KKT check): the reference's qpOASES run regularises -- H + rho I, then one more QP with the gradient
    // g - rho x_1 (QProblem.cpp:1753-1860, QProblemB.cpp:1999-2031; KernelArgs::reg_step) -- and so do two more launches here
    std::vector<int> indef;
    HIP_TRY(hipMemcpy(st.data(), h->d_status, (size_t)h->batch * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int i : idx)
      if (HMPC_STATUS_CODE(st[i]) == HMPC_S_INDEFINITE) indef.push_back(i);
    if (!indef.empty()) {
      HIP_TRY(hipMemcpy(d_idx, indef.data(), indef.size() * sizeof(int), hipMemcpyHostToDevice));
      LaunchOpt ro = so;
      ro.n_list = (int)indef.size(), ro.relax = 0.0, ro.warm = 1, ro.skip_ok = 0;
      for (int step = 1; step <= 2; ++step) {
        ro.reg_step = step;
        rc = launch_safe(h, h->last_stream, ro);
        if (rc != HMPC_OK) return rc;
      }
      HIP_TRY(hipStreamSynchronize(h->last_stream));
      HIP_TRY(hipMemcpy(d_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));  // (the passes below index into the full list again)
    }
  }
  // last resort for instances that cycle at a degenerate vertex even with the full-size working set: bounds moved outward
  // by 1e-7, then 1e-6 (a different amount per row), reported as HMPC_S_OK_RELAXED
  if (h->nc == 3) {
    // second level for three contacts: instances whose working set outgrew even the LDS-resident safe variant (140 of 180 rows)
    std::vector<int> full;
    HIP_TRY(hipMemcpy(st.data(), h->d_status, (size_t)h->batch * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int i : idx)
      if (HMPC_STATUS_CODE(st[i]) == HMPC_S_WORKSET) full.push_back(i);
    if (!full.empty()) {
      HIP_TRY(hipMemcpy(d_idx, full.data(), full.size() * sizeof(int), hipMemcpyHostToDevice));
      so.n_list = (int)full.size(), so.ultimate = true;
      rc = launch(h, h->last_stream, so);
      if (rc != HMPC_OK) return rc;
      HIP_TRY(hipStreamSynchronize(h->last_stream));
    }
  }
  // (with the exact re-solve that ends a relaxed pass -- see the kernel -- a larger perturbation costs nothing when its working
  //  set turns out to be optimal for the exact bounds: such an instance is reported HMPC_S_OK, exact)
  const double relax_levels[3] = {1e-7, 1e-6, 1e-5};
  for (int lvl = 0; lvl < 3; ++lvl) {
    std::vector<int> still;
    HIP_TRY(hipMemcpy(st.data(), h->d_status, (size_t)h->batch * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int i : idx) {
      const uint32_t c = HMPC_STATUS_CODE(st[i]);
      if ((c == HMPC_S_MAXITER && !capped_by_caller(h, st[i])) || c == HMPC_S_INFEASIBLE || c == HMPC_S_KKT || c == HMPC_S_WORKSET) still.push_back(i);
    }
    if (still.empty()) break;
    HIP_TRY(hipMemcpy(d_idx, still.data(), still.size() * sizeof(int), hipMemcpyHostToDevice));
    so.n_list = (int)still.size(), so.relax = relax_levels[lvl], so.warm = 0;
    rc = launch_safe(h, h->last_stream, so);
    if (rc != HMPC_OK) return rc;
    HIP_TRY(hipStreamSynchronize(h->last_stream));
  }
  return HMPC_OK;
}

// the safe pass of hmpc_download / hmpc_download_f64: one 4-byte read of the device's flagged counter decides whether
// the status words need to be scanned at all (they almost never do: nominal inputs flag nothing)
static int resolve_if_flagged(hmpc_handle *h) {
  unsigned int now = 0;
  HIP_TRY(hipMemcpy(&now, h->d_flagged, sizeof(now), hipMemcpyDeviceToHost));
  if (now == h->flagged_seen) return HMPC_OK;
  const int rc = hmpc_resolve_failed(h, nullptr);
  if (rc != HMPC_OK) return rc;
  HIP_TRY(hipMemcpy(&now, h->d_flagged, sizeof(now), hipMemcpyDeviceToHost));  // the safe pass may have flagged again
  h->flagged_seen = now;
  return HMPC_OK;
}

int hmpc_download(hmpc_handle *h, float *forces, uint32_t *status) {
  if (!h) return HMPC_E_ARG;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->last_stream));
  if (h->auto_resolve && h->batch) {
    int rc = resolve_if_flagged(h);
    if (rc != HMPC_OK) return rc;
  }
  const size_t nf = (size_t)h->batch * 6 * h->nc * h->setup.horizon;
  if (forces && nf) HIP_TRY(hipMemcpy(forces, h->d_forces, nf * sizeof(float), hipMemcpyDeviceToHost));
  if (status && h->batch)
    HIP_TRY(hipMemcpy(status, h->d_status, (size_t)h->batch * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return HMPC_OK;
}

int hmpc_get_device_outputs(hmpc_handle *h, float **device_forces, uint32_t **device_status) {
  if (!h) return HMPC_E_ARG;
  if (device_forces) *device_forces = h->d_forces;
  if (device_status) *device_status = h->d_status;
  return HMPC_OK;
}

int hmpc_batch(const hmpc_handle *h) { return h ? h->batch : HMPC_E_ARG; }
int hmpc_horizon(const hmpc_handle *h) { return h ? h->setup.horizon : HMPC_E_ARG; }

int hmpc_time_solve(hmpc_handle *h, void *stream, int reps, float *ms_per_launch) {
  if (!h || reps < 1 || !ms_per_launch) return HMPC_E_ARG;
  *ms_per_launch = 0.f;
  if (h->batch == 0) return HMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  EventPair ev;  // destroyed on every return path
  HIP_TRY(hipEventCreate(&ev.e0));
  HIP_TRY(hipEventCreate(&ev.e1));
  hipStream_t s = (hipStream_t)stream;
  h->last_stream = s;
  HIP_TRY(hipEventRecord(ev.e0, s));
  for (int i = 0; i < reps; ++i) {
    // a single timed launch is a genuine solve of the current batch (it consumes and leaves the tick-to-tick working sets);
    // repetitions of the same batch must not advance them again
    int rc = enqueue_solve(h, s, /*carry_wset=*/reps == 1);  // the same launches hmpc_solve enqueues (size classes, device repair)
    if (rc != HMPC_OK) return rc;
  }
  HIP_TRY(hipEventRecord(ev.e1, s));
  HIP_TRY(hipEventSynchronize(ev.e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  *ms_per_launch = ms / (float)reps;
  return HMPC_OK;
}

int hmpc_debug_assemble(hmpc_handle *h, int index, int *n, int *m, int *var_ind, float *H, float *g, float *Fc,
                        float *lb, float *ub, float *x0, float *Acd, float *Bcd) {
  if (!h || index < 0 || index >= h->batch) return HMPC_E_ARG;
  HIP_TRY(hipSetDevice(h->device));
  int vi = 0;
  const Variant &v = pick_variant(h, &vi);
  if (!h->d_dbg_f) {
    HIP_TRY(hipMalloc(&h->d_dbg_f, sizeof(float) * (size_t)DBG_FLOATS_MAX));
    HIP_TRY(hipMalloc(&h->d_dbg_i, sizeof(int) * (2 + MAX_VARS_ANY)));
  }
  HIP_TRY(hipMemset(h->d_dbg_i, 0, sizeof(int) * (2 + MAX_VARS_ANY)));
  LaunchOpt ao;
  ao.assemble_only = true, ao.dbg_index = index;
  int rc = launch(h, 0, ao);
  if (rc != HMPC_OK) return rc;
  HIP_TRY(hipDeviceSynchronize());
  std::vector<float> hf(v.dbg_floats);
  std::vector<int> hi(2 + MAX_VARS_ANY);
  HIP_TRY(hipMemcpy(hf.data(), h->d_dbg_f, sizeof(float) * hf.size(), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(hi.data(), h->d_dbg_i, sizeof(int) * hi.size(), hipMemcpyDeviceToHost));
  const int nn = hi[0], mm = hi[1], hz = h->setup.horizon;
  if (n) *n = nn;
  if (m) *m = mm;
  if (nn > v.nmax) return HMPC_OK;  // too large: only n, m are meaningful
  const int NM = v.nmax, nc = v.nc;
  const int oG = NM * NM, oFC = oG + NM, oLB = oFC + 48 * nc * nc, oUB = oLB + 160 * nc, oX0 = oUB + 160 * nc,
            oACD = oX0 + 16, oBCD = oACD + 176;
  if (var_ind) memcpy(var_ind, hi.data() + 2, sizeof(int) * nn);
  if (H) memcpy(H, hf.data(), sizeof(float) * (size_t)nn * nn);
  if (g) memcpy(g, hf.data() + oG, sizeof(float) * nn);
  if (Fc) memcpy(Fc, hf.data() + oFC, sizeof(float) * 48 * nc * nc);
  if (lb) memcpy(lb, hf.data() + oLB, sizeof(float) * 8 * nc * hz);
  if (ub) memcpy(ub, hf.data() + oUB, sizeof(float) * 8 * nc * hz);
  if (x0) memcpy(x0, hf.data() + oX0, sizeof(float) * 13);
  if (Acd) memcpy(Acd, hf.data() + oACD, sizeof(float) * 169);
  if (Bcd) memcpy(Bcd, hf.data() + oBCD, sizeof(float) * 78 * nc);
  return HMPC_OK;
}

// Parity hook: stages S, W, Q of the kernel (inverse by sweeps, block start, dual active set, scatter) on QP data handed
// in from outside -- e.g. the reference's own H_red / g_red / fmat as its source left them -- instead of the kernel's own
// assembly.  The current batch's records still provide the gait tables (which leg-steps exist) and f_max.
int hmpc_debug_solve_external_qp(hmpc_handle *h, const float *H, const float *g, const float *Fc, int ld) {
  if (!h || !H || !g || !Fc || ld < 1) return HMPC_E_ARG;
  if (h->batch == 0) return HMPC_OK;
  {
    // the kernel reads H[i*ld + j], g[i] for i, j < the instance's reduced-variable count: ld must cover the widest one
    // (known for host-uploaded records; otherwise whatever the variant that will be launched can hold)
    int vi = 0;
    const Variant &v = pick_variant(h, &vi);
    const int need = (h->max_stance >= 0) ? h->max_stance : v.nmax;
    if (ld < need) return HMPC_E_ARG;
  }
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->last_stream));
  const size_t nb = (size_t)h->batch, nfc = (size_t)8 * h->nc * 6 * h->nc;
  const size_t bytes = sizeof(float) * nb * ((size_t)ld * ld + ld + nfc);
  float *d = nullptr;
  HIP_TRY(hipMalloc(&d, bytes));
  struct Free {
    float *p;
    ~Free() { (void)hipFree(p); }
  } guard{d};
  float *dH = d, *dg = dH + nb * ld * ld, *dF = dg + nb * ld;
  HIP_TRY(hipMemcpy(dH, H, sizeof(float) * nb * ld * ld, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dg, g, sizeof(float) * nb * ld, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dF, Fc, sizeof(float) * nb * nfc, hipMemcpyHostToDevice));
  h->d_ext_H = dH, h->d_ext_g = dg, h->d_ext_Fc = dF, h->ext_ld = ld;
  LaunchOpt xo;
  xo.carry_wset = false;
  const int rc = launch(h, h->last_stream, xo);
  hipError_t e = hipStreamSynchronize(h->last_stream);
  h->d_ext_H = h->d_ext_g = h->d_ext_Fc = nullptr;
  h->ext_ld = 0;
  if (rc != HMPC_OK) return rc;
  HIP_TRY(e);
  return HMPC_OK;
}

int hmpc_debug_phase_cycles(hmpc_handle *h, long long *cycles /*[batch][NPROF = 32]*/) {
#ifndef HMPC_PROFILE
  (void)h;
  (void)cycles;
  g_hip_err = "library built without -DHMPC_PROFILE";
  return HMPC_E_ARG;
#else
  if (!h || !cycles) return HMPC_E_ARG;
  HIP_TRY(hipSetDevice(h->device));
  const size_t nb = (size_t)h->max_batch * hmpc::NPROF * sizeof(long long);
  if (!h->d_prof) HIP_TRY(hipMalloc(&h->d_prof, nb));
  HIP_TRY(hipMemset(h->d_prof, 0, nb));
  // (with the device-side repair on, the whole chain: an instance's slot then holds the numbers of the LAST variant that ran it)
  int rc = h->device_repair ? enqueue_solve(h, h->last_stream, false) : launch(h, h->last_stream, LaunchOpt());
  if (rc != HMPC_OK) return rc;
  HIP_TRY(hipStreamSynchronize(h->last_stream));
  HIP_TRY(hipMemcpy(cycles, h->d_prof, (size_t)h->batch * hmpc::NPROF * sizeof(long long), hipMemcpyDeviceToHost));
  HIP_TRY(hipFree(h->d_prof));
  h->d_prof = nullptr;
  return HMPC_OK;
#endif
}

// ---------------------------------------------------------------------------------------------------- rows f1-f3
int hmpc_build_records_device(hmpc_handle *h, const void *device_ticks, int batch, double dtMPC, double *device_wpd_out,
                              void *stream) {
  if (h && h->nc != 2) return HMPC_E_ARG;  // rows f1-f3 restate the reference's two-foot controller code
  if (!h || !device_ticks || batch < 0) return HMPC_E_ARG;
  if (batch > h->max_batch) return HMPC_E_BATCH;
  HIP_TRY(hipSetDevice(h->device));
  if (batch > 0) {
    const int nwords = (int)h->stride / 4;
    const int bs = ((nwords + 63) / 64) * 64 > 256 ? 256 : ((nwords + 63) / 64) * 64;
    hipLaunchKernelGGL(hmpc::build_records_kernel, dim3(batch), dim3(bs), 0, (hipStream_t)stream,
                       (const hmpc_tick_inputs *)device_ticks, batch, h->setup.horizon, dtMPC, h->d_records_own,
                       (int)h->stride, device_wpd_out, h->setup.f_max, h->d_cls);
    HIP_TRY(hipGetLastError());
  }
  h->d_records = h->d_records_own;
  h->batch = batch;
  h->max_stance = -1;  // the builder left every instance's size class on the device: hmpc_solve routes by it
  h->cls_valid = 1;
  h->last_stream = (hipStream_t)stream;
  return HMPC_OK;
}

int hmpc_build_records(hmpc_handle *h, const struct hmpc_tick_inputs *host_ticks, int batch, double dtMPC, double *wpd_out) {
  if (!h || !host_ticks || batch < 0) return HMPC_E_ARG;
  if (batch > h->max_batch) return HMPC_E_BATCH;
  HIP_TRY(hipSetDevice(h->device));
  const size_t nb = (size_t)(batch > 0 ? batch : 1);
  const size_t off_w = (sizeof(hmpc_tick_inputs) * nb + 255) & ~(size_t)255;
  void *sp = nullptr;
  int rc = scratch(h, off_w + sizeof(double) * 2 * nb, &sp);
  if (rc != HMPC_OK) return rc;
  hmpc_tick_inputs *d_t = (hmpc_tick_inputs *)sp;
  double *d_w = (double *)((char *)sp + off_w);
  HIP_TRY(hipMemcpy(d_t, host_ticks, sizeof(hmpc_tick_inputs) * (size_t)batch, hipMemcpyHostToDevice));
  rc = hmpc_build_records_device(h, d_t, batch, dtMPC, d_w, nullptr);
  if (rc != HMPC_OK) return rc;
  HIP_TRY(hipStreamSynchronize(nullptr));
  if (wpd_out && batch) HIP_TRY(hipMemcpy(wpd_out, d_w, sizeof(double) * 2 * (size_t)batch, hipMemcpyDeviceToHost));
  return HMPC_OK;
}

int hmpc_body_wrench_device(hmpc_handle *h, const double *device_rBody, double *device_f_ff, void *stream) {
  if (h && h->nc != 2) return HMPC_E_ARG;  // rows f1-f3 restate the reference's two-foot controller code
  if (!h || !device_rBody || !device_f_ff) return HMPC_E_ARG;
  HIP_TRY(hipSetDevice(h->device));
  if (h->batch > 0) {
    const int total = 12 * h->batch;
    hipLaunchKernelGGL(hmpc::body_wrench_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->d_forces,
                       h->batch, h->setup.horizon, device_rBody, device_f_ff);
    HIP_TRY(hipGetLastError());
  }
  return HMPC_OK;
}

int hmpc_body_wrench(hmpc_handle *h, const double *host_rBody, double *host_f_ff) {
  if (!h || !host_rBody || !host_f_ff) return HMPC_E_ARG;
  if (h->batch == 0) return HMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  const size_t b = (size_t)h->batch;
  void *sp = nullptr;
  int rc = scratch(h, sizeof(double) * (9 + 12) * b, &sp);
  if (rc != HMPC_OK) return rc;
  double *d_r = (double *)sp, *d_f = d_r + 9 * b;
  HIP_TRY(hipMemcpy(d_r, host_rBody, sizeof(double) * 9 * b, hipMemcpyHostToDevice));
  HIP_TRY(hipStreamSynchronize(h->last_stream));
  rc = hmpc_body_wrench_device(h, d_r, d_f, nullptr);
  if (rc != HMPC_OK) return rc;
  HIP_TRY(hipStreamSynchronize(nullptr));
  HIP_TRY(hipMemcpy(host_f_ff, d_f, sizeof(double) * 12 * b, hipMemcpyDeviceToHost));
  return HMPC_OK;
}

int hmpc_leg_torques_device(hmpc_handle *h, const double *device_rBody, const double *device_leg_q, double *device_f_ff,
                            double *device_tau, void *stream) {
  if (h && h->nc != 2) return HMPC_E_ARG;  // rows f1-f3 restate the reference's two-foot controller code
  if (!h || !device_rBody || !device_leg_q || !device_tau) return HMPC_E_ARG;
  HIP_TRY(hipSetDevice(h->device));
  if (h->batch > 0) {
    const int total = 2 * h->batch;
    hipLaunchKernelGGL(hmpc::leg_torque_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->d_forces,
                       h->batch, h->setup.horizon, device_rBody, device_leg_q, device_f_ff, device_tau,
                       (const hmpc_tick_inputs *)nullptr);
    HIP_TRY(hipGetLastError());
  }
  return HMPC_OK;
}

int hmpc_leg_torques(hmpc_handle *h, const double *host_rBody, const double *host_leg_q, double *host_f_ff, double *host_tau) {
  if (!h || !host_rBody || !host_leg_q || !host_tau) return HMPC_E_ARG;
  if (h->batch == 0) return HMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  const size_t nb = (size_t)h->batch;
  void *sp = nullptr;
  int rc = scratch(h, sizeof(double) * (9 + 10 + 12 + 10) * nb, &sp);
  if (rc != HMPC_OK) return rc;
  double *d_r = (double *)sp, *d_q = d_r + 9 * nb, *d_f = d_q + 10 * nb, *d_t = d_f + 12 * nb;
  HIP_TRY(hipMemcpy(d_r, host_rBody, sizeof(double) * 9 * nb, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_q, host_leg_q, sizeof(double) * 10 * nb, hipMemcpyHostToDevice));
  HIP_TRY(hipStreamSynchronize(h->last_stream));
  rc = hmpc_leg_torques_device(h, d_r, d_q, d_f, d_t, nullptr);
  if (rc != HMPC_OK) return rc;
  HIP_TRY(hipStreamSynchronize(nullptr));
  if (host_f_ff) HIP_TRY(hipMemcpy(host_f_ff, d_f, sizeof(double) * 12 * nb, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(host_tau, d_t, sizeof(double) * 10 * nb, hipMemcpyDeviceToHost));
  return HMPC_OK;
}

// One MPC tick of a whole batch without leaving the device (rows f1+f2 -> a1..a16 -> f3 of SURVEY.md section 8):
// updateMPCIfNeeded's input construction + gait table (ConvexMPCLocomotion.cpp:283-406, GaitGenerator.cpp:85-103), the solve
// routed by size class (a walking tick runs on the 60-variable variant), then f_ff = -rBody [GRF; GRM] and tau = J_fm' f_ff
// (ConvexMPCLocomotion.cpp:419-440, common/LegController.cpp:57-61, 108-167) -- three or four launches on ONE stream, no
// host synchronisation, nothing but the tick structs in and the torques out.
int hmpc_tick_solve_device(hmpc_handle *h, const void *device_ticks, int batch, double dtMPC, double *device_wpd_out,
                           double *device_f_ff, double *device_tau, void *stream) {
  if (!h || !device_ticks || !device_tau || batch < 0) return HMPC_E_ARG;
  int rc = hmpc_build_records_device(h, device_ticks, batch, dtMPC, device_wpd_out, stream);
  if (rc != HMPC_OK) return rc;
  if (batch == 0) return HMPC_OK;
  rc = enqueue_solve(h, (hipStream_t)stream, /*carry_wset=*/true);
  if (rc != HMPC_OK) return rc;
  const int total = 2 * batch;
  hipLaunchKernelGGL(hmpc::leg_torque_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->d_forces, batch,
                     h->setup.horizon, (const double *)nullptr, (const double *)nullptr, device_f_ff, device_tau,
                     (const hmpc_tick_inputs *)device_ticks);
  HIP_TRY(hipGetLastError());
  return HMPC_OK;
}

int hmpc_download_records(hmpc_handle *h, void *host_records) {
  if (!h || !host_records) return HMPC_E_ARG;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->last_stream));
  if (h->batch) HIP_TRY(hipMemcpy(host_records, h->d_records, (size_t)h->batch * h->stride, hipMemcpyDeviceToHost));
  return HMPC_OK;
}

int hmpc_enable_f64_output(hmpc_handle *h) {
  if (!h) return HMPC_E_ARG;
  if (h->d_x64) return HMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  const size_t nf = (size_t)h->max_batch * 6 * h->nc * h->setup.horizon;
  HIP_TRY(hipMalloc(&h->d_x64, nf * sizeof(double)));
  if (hipMalloc(&h->d_obj64, (size_t)h->max_batch * sizeof(double)) != hipSuccess) {
    hipFree(h->d_x64);
    h->d_x64 = nullptr;
    g_hip_err = "hipMalloc failed in hmpc_enable_f64_output";
    return HMPC_E_HIP;
  }
  return HMPC_OK;
}

int hmpc_download_f64(hmpc_handle *h, double *x, double *obj) {
  if (!h) return HMPC_E_ARG;
  if (h->batch == 0) return HMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  if (!h->d_x64) {
    // the copy-out was not enabled before the solve: enable it and run the current batch once more -- without consuming
    // the tick-to-tick working sets a second time -- then give flagged instances the same safe pass hmpc_download gives
    int rc = hmpc_enable_f64_output(h);
    if (rc != HMPC_OK) return rc;
    rc = enqueue_solve(h, h->last_stream, /*carry_wset=*/false);
    if (rc != HMPC_OK) return rc;
  }
  HIP_TRY(hipStreamSynchronize(h->last_stream));
  if (h->auto_resolve) {
    const int rc = resolve_if_flagged(h);
    if (rc != HMPC_OK) return rc;
  }
  const size_t nb = (size_t)h->batch * 6 * h->nc * h->setup.horizon;
  if (x && nb) HIP_TRY(hipMemcpy(x, h->d_x64, nb * sizeof(double), hipMemcpyDeviceToHost));
  if (obj && h->batch) HIP_TRY(hipMemcpy(obj, h->d_obj64, (size_t)h->batch * sizeof(double), hipMemcpyDeviceToHost));
  return HMPC_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// The reference's own interface (convexMPC_interface.cpp:42-118): process-global, single-threaded, blocking.
// ------------------------------------------------------------------------------------------------------------------
static problem_setup g_setup = {0.f, 0.f, 0.f, 0};
static update_data_t g_update;
static hmpc_handle *g_handle = nullptr;
static double *g_q_soln = nullptr;  // 12*horizon doubles, solver-owned (SolverMPC.cpp:52, :94-97)
static int g_q_len = 0;
static int g_has_solved = 0;
static uint32_t g_last_status = 0;
static int g_setup_error = 0;
static hmpc_params g_legacy_params = {9.0f, {0.5413f, 0.5200f, 0.0691f}, 2.0f, 0.09f, 0.06f, 9.81f};  // hmpc_legacy_set_params
static int g_legacy_iter_cap = 0;  // hmpc_legacy_set_max_iterations: explicit opt-in (update_solver_settings is inert, as in the reference)
// one tick = one pinned staging buffer [record | 12h forces | status word] and one contiguous device output block, so that
// a blocking tick costs one asynchronous H2D copy, one launch, one asynchronous D2H copy and a single synchronisation
static unsigned char *g_pin = nullptr;
static float *g_dev_out = nullptr;
static size_t g_pin_rec_bytes = 0;

static void free_tick_buffers(void) {
  if (g_pin) hipHostFree(g_pin);
  if (g_dev_out) hipFree(g_dev_out);
  g_pin = nullptr, g_dev_out = nullptr, g_pin_rec_bytes = 0;
}

void setup_problem(double dt, int horizon, double mu, double f_max) {
  g_setup.horizon = horizon;
  g_setup.f_max = (float)f_max;
  g_setup.mu = (float)mu;
  g_setup.dt = (float)dt;
  g_setup_error = 0;
  if (horizon < 1 || horizon > HMPC_MAX_HORIZON) {
    // the reference throws std::runtime_error("horizon is too long!") from c2qp for horizon > 19; we never throw across C
    fprintf(stderr, "[hector_mpc_hip] setup_problem: horizon %d outside [1,%d]\n", horizon, HMPC_MAX_HORIZON);
    g_setup_error = HMPC_E_HORIZON;
    return;
  }
  // the reference frees and re-mallocs every buffer on every call (resize_qp_mats); we only rebuild when the
  // problem shape or scalars change, the observable behaviour (q_soln valid until the next setup) is the same.
  if (g_handle && (g_handle->setup.horizon != horizon || g_handle->setup.dt != g_setup.dt ||
                   g_handle->setup.f_max != g_setup.f_max)) {
    hmpc_destroy(g_handle);
    g_handle = nullptr;
    free_tick_buffers();
  }
  if (!g_handle) {
    // the reference has no notion of a device: HMPC_DEVICE (default 0) picks the GPU of the process-global solver
    const char *env = getenv("HMPC_DEVICE");
    const int dev = (env && *env) ? atoi(env) : 0;
    int rc = hmpc_create(&g_handle, &g_setup, 1, dev);
    if (rc != HMPC_OK) {
      fprintf(stderr, "[hector_mpc_hip] setup_problem failed (%d): %s\n", rc, hmpc_last_hip_error());
      g_handle = nullptr;
      g_setup_error = rc;
      return;
    }
    g_pin_rec_bytes = (record_stride(horizon) + 63) & ~(size_t)63;
    const size_t out_bytes = sizeof(float) * 12 * horizon + sizeof(uint32_t);
    if (hipHostMalloc((void **)&g_pin, g_pin_rec_bytes + out_bytes, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&g_dev_out, out_bytes) != hipSuccess ||
        hmpc_set_device_outputs(g_handle, g_dev_out, (uint32_t *)(g_dev_out + 12 * horizon)) != HMPC_OK) {
      fprintf(stderr, "[hector_mpc_hip] setup_problem: could not allocate the tick buffers\n");
      free_tick_buffers();
      hmpc_destroy(g_handle);
      g_handle = nullptr;
      g_setup_error = HMPC_E_HIP;
      return;
    }
  }
  if (g_q_len != 12 * horizon) {
    free(g_q_soln);
    g_q_soln = (double *)calloc((size_t)12 * horizon, sizeof(double));
    g_q_len = 12 * horizon;
  }
}

static void solve_global(void) {
  if (!g_handle || g_setup_error) {
    fprintf(stderr, "[hector_mpc_hip] solve requested without a valid setup_problem (error %d)\n", g_setup_error);
    return;
  }
  const int hz = g_setup.horizon;
  unsigned char *rec = g_pin;
  memset(rec, 0, record_stride(hz));
  float *f = (float *)rec;
  memcpy(f + 0, g_update.p, 12), memcpy(f + 3, g_update.v, 12), memcpy(f + 6, g_update.q, 16);
  memcpy(f + 10, g_update.w, 12), memcpy(f + 13, g_update.r, 24), memcpy(f + 19, g_update.joint_angles, 40);
  f[29] = g_update.yaw;
  memcpy(f + 30, g_update.weights, 48), memcpy(f + 42, g_update.Alpha_K, 48);
  memcpy(f + 54, g_update.traj, sizeof(float) * 12 * hz);
  memcpy(rec + 4 * (54 + 12 * hz), g_update.gait, 2 * hz);
  const float *forces = (const float *)(g_pin + g_pin_rec_bytes);
  const uint32_t *pst = (const uint32_t *)(forces + 12 * hz);
  const size_t out_bytes = sizeof(float) * 12 * hz + sizeof(uint32_t);
  uint32_t st = 0;
  hmpc_set_max_iterations(g_handle, g_legacy_iter_cap);
  hmpc_set_params(g_handle, &g_legacy_params);
  int rc = hmpc_upload_records_async(g_handle, rec, 1, nullptr);  // pinned source: a true asynchronous copy
  if (rc == HMPC_OK) rc = hmpc_solve(g_handle, nullptr);
  if (rc == HMPC_OK && (hipMemcpyAsync(g_pin + g_pin_rec_bytes, g_dev_out, out_bytes, hipMemcpyDeviceToHost, nullptr) != hipSuccess ||
                        hipStreamSynchronize(nullptr) != hipSuccess))
    rc = HMPC_E_HIP;
  if (rc == HMPC_OK) {
    st = *pst;
    const uint32_t c0 = HMPC_STATUS_CODE(st);
    if (c0 == HMPC_S_WORKSET || (c0 == HMPC_S_MAXITER && !capped_by_caller(g_handle, st)) || c0 == HMPC_S_INFEASIBLE || c0 == HMPC_S_KKT) {
      // flagged by the fast variant: the safe pass (full-size working set, then relaxed bounds), as hmpc_download gives it
      rc = hmpc_resolve_failed(g_handle, nullptr);
      if (rc == HMPC_OK && hipMemcpy(g_pin + g_pin_rec_bytes, g_dev_out, out_bytes, hipMemcpyDeviceToHost) != hipSuccess)
        rc = HMPC_E_HIP;
      st = *pst;
    }
  }
  if (rc != HMPC_OK) {
    fprintf(stderr, "[hector_mpc_hip] solve failed (%d): %s\n", rc, hmpc_last_hip_error());
    return;
  }
  g_last_status = st;
  // SolverMPC.cpp:714-715: the reference prints this line and scatters whatever qpOASES left in q_red all the same; so do
  // we (the forces of a flagged instance are the last iterate; hmpc_last_status() tells the caller, which the reference
  // cannot).  HMPC_S_OK_RELAXED is a solved instance.
  const uint32_t code = HMPC_STATUS_CODE(st);
  if (code != HMPC_S_OK && code != HMPC_S_OK_RELAXED) printf("failed to solve!\n");
  for (int i = 0; i < 12 * hz; ++i) g_q_soln[i] = (double)forces[i];
  g_has_solved = 1;
}

void update_problem_data(double *p, double *v, double *q, double *w, double *r, double *joint_angles, double yaw,
                         double *weights, double *state_trajectory, double *Alpha_K, int *gait) {
  const int hz = g_setup.horizon;
  if (hz < 1 || hz > HMPC_MAX_HORIZON) return;
  for (int i = 0; i < 3; ++i) g_update.p[i] = (float)p[i], g_update.v[i] = (float)v[i], g_update.w[i] = (float)w[i];
  for (int i = 0; i < 4; ++i) g_update.q[i] = (float)q[i];
  for (int i = 0; i < 6; ++i) g_update.r[i] = (float)r[i];
  for (int i = 0; i < 10; ++i) g_update.joint_angles[i] = (float)joint_angles[i];
  g_update.yaw = (float)yaw;
  for (int i = 0; i < 12; ++i) g_update.weights[i] = (float)weights[i], g_update.Alpha_K[i] = (float)Alpha_K[i];
  for (int i = 0; i < 12 * hz; ++i) g_update.traj[i] = (float)state_trajectory[i];
  for (int i = 0; i < 2 * hz; ++i) g_update.gait[i] = (unsigned char)gait[i];
  solve_global();
}

double get_solution(int index) {
  if (!g_has_solved) return 0.0;  // convexMPC_interface.cpp:107
  if (index < 0 || index >= g_q_len) return 0.0;
  return g_q_soln[index];
}

void update_solver_settings(int max_iter, double rho, double sigma, double solver_alpha, double terminate,
                            double use_jcqp) {
  // Stored exactly as the reference stores them (convexMPC_interface.cpp:112-118) -- and, as in the reference, read by
  // NOTHING: its qpOASES path runs with a fixed nWSR (SolverMPC.cpp:706) whatever max_iter says, so a caller that passes a
  // small JCQP/ADMM-style max_iter (the knobs belong to a solver the reference does not ship) gets full solves there and
  // must get them here.  The opt-in with a meaning for this solver is hmpc_legacy_set_max_iterations / hmpc_set_max_iterations.
  g_update.max_iterations = max_iter;
  g_update.rho = rho;
  g_update.sigma = sigma;
  g_update.solver_alpha = solver_alpha;
  g_update.terminate = terminate;
  (void)use_jcqp;
}

int hmpc_legacy_set_params(const struct hmpc_params *p) {
  hmpc_params d;
  hmpc_default_params(&d);
  if (p && !params_ok(*p)) return HMPC_E_ARG;
  g_legacy_params = p ? *p : d;
  return HMPC_OK;
}

int hmpc_legacy_set_max_iterations(int max_iter) {
  if (max_iter < 0) return HMPC_E_ARG;
  g_legacy_iter_cap = max_iter;
  return HMPC_OK;
}

void hmpc_solve_mpc(struct update_data_t *update, struct problem_setup *setup) {
  if (!update || !setup) return;
  if (!g_handle || g_setup.horizon != setup->horizon || g_setup.dt != setup->dt || g_setup.f_max != setup->f_max)
    setup_problem((double)setup->dt, setup->horizon, (double)setup->mu, (double)setup->f_max);
  if (update != &g_update) g_update = *update;
  solve_global();
}
void solveDenseMPC(struct update_data_t *update, struct problem_setup *setup) { hmpc_solve_mpc(update, setup); }
double *hmpc_get_q_soln(void) { return g_q_soln; }
uint32_t hmpc_last_status(void) { return g_last_status; }

}  // extern "C"

// C++-linkage symbols with the reference's exact names (SolverMPC.h:56, :63), for callers that include its header
void solve_mpc(update_data_t *update, problem_setup *setup) { hmpc_solve_mpc(update, setup); }
double *get_q_soln() { return hmpc_get_q_soln(); }
