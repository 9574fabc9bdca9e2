// fccqp_capi.cpp -- implementation of the C ABI in include/fccqp.h on top of the gfx950 kernels.
//
// Replaces the body of the reference class fcc_qp::FCCQP (src/fcc_qp.cpp:24-207): the handle owns
// the device-resident workspace and warm-start state that the reference keeps in Eigen members
// (src/fcc_qp.hpp:135-171), one slice per instance of the batch.
// Compiled by hipcc (host side only; kernels live in the .hip files).
#include "../../include/fccqp.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <new>
#include <string>
#include <vector>

#include "fccqp_device.h"
#include "fccqp_jit.h"

struct fccqp_solver {
  int n = 0, m = 0, nc = 0, s = 0, batch = 0, device = 0;
  fccqp_options opt{1000, 1e-6, 1e-3, 1e-6};  // reference defaults, src/fcc_qp.hpp:30-35
  bool warm_start = false;                    // src/fcc_qp.hpp:133
  int kernel_request = FCCQP_KERNEL_AUTO;
  int kernel_used = FCCQP_KERNEL_GENERIC;
  bool solved = false;
  bool last_was_warm = false;
  bool last_was_f32 = false;   // the last solve read Q / A_eq as float
  double last_solve_wall = 0.0;
  size_t lds_limit = 64 * 1024;
  int compute_units = 256;
  hipStream_t stream = nullptr;      // the solver's own stream
  hipStream_t last_stream = nullptr; // stream of the last enqueue
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;   // kernel start / end, async result copy
  bool copy_pending = false;
  // staged inputs (host-pointer solve)
  double *wb = nullptr, *wbeq = nullptr, *wmu = nullptr, *wlb = nullptr, *wub = nullptr;  // widened vectors
  double *dQ = nullptr, *db = nullptr, *dA = nullptr, *dbeq = nullptr, *dmu = nullptr,
         *dlb = nullptr, *dub = nullptr;
  // The seven staged inputs are carved out of ONE device block, and (small batches) mirrored by one
  // page-locked host block of the same layout: a host-pointer solve of a few instances is one
  // memcpy-into-pinned + one H2D copy instead of seven staged copies from pageable memory (~10 us each:
  // more than the kernel takes for the reference's one-instance-at-a-time use).
  char *din_block = nullptr, *hin_block = nullptr;
  size_t in_off[7] = {0, 0, 0, 0, 0, 0, 0}, in_bytes = 0;
  // Likewise the per-instance results (z, metrics, ticks, n_iter, flags): one device block, one D2H copy.
  char *dres_block = nullptr, *hres_block = nullptr;
  // small synchronous host-pointer solves (the reference's one-QP-at-a-time use): the kernel reads its inputs straight
  // out of the page-locked block (hin_dev: the block's device-side address; no H2D copy), the result block is copied
  // back behind the kernel under the same synchronisation (hres_fresh: GetSolution then copies nothing), and the
  // fallback-list launch is issued only if an instance was flagged (pending_fb)
  char *hin_dev = nullptr;
  bool hres_fresh = false;
  // Transfers between CALLER memory and the device go through two page-locked bounce buffers of the handle (copy_in /
  // copy_out below): the GPU never addresses the caller's pages.  See the note at copy_in.
  char *bounce[2] = {nullptr, nullptr};
  hipEvent_t bounce_ev[2] = {nullptr, nullptr};
  bool bounce_busy[2] = {false, false};
  int bounce_next = 0;
  // the last solve was waited for on the host and recorded no events (a timing event costs ~5 us of stream time
  // each; a one-QP solve is 50-90 us): nothing is left to wait for, and fccqp_last_kernel_ms reports host_device_ms,
  // the host-measured time from the launch to the completion of the device work
  bool quiesced = false;
  float host_device_ms = 0.f;
  bool fb_deferred = false;
  fccqp::KernelArgs pending_fb{};
  bool pending_fb_f32 = false;
  size_t res_off[5] = {0, 0, 0, 0, 0}, res_bytes = 0;
  // state / outputs
  double *dx = nullptr, *dmux = nullptr, *dmulc = nullptr, *dmetrics = nullptr, *dwork = nullptr;
  int *dniter = nullptr, *dflags = nullptr;
  long long *dticks = nullptr;
  long long *dphase = nullptr;  // [8][batch], allocated on demand
  bool fallback = true;          // device COD fallback for a rank-deficient presolve
  bool kernel_timing = true;     // record the start event of every solve (fccqp_set_kernel_timing)
  bool self_check = false;       // equality-residual check behind every solve (fccqp_set_self_check; on for N > 256)
  fccqp::KernelArgs check_args{}; // ... its arguments for a deferred fallback pass (run_deferred_fallback)
  bool timed = false;            // ... and whether the last solve did
  int *dfb_list = nullptr, *dfb_count = nullptr;  // dfb_count[2]: used alternately, see enqueue()
  unsigned solve_seq = 0;
  double *dcod_ws = nullptr;
  long long cod_stride = 0;
  // persistent one-wave workgroups of the fallback launch (list mode).  Round 1 used 64 whatever the
  // batch: a batch whose instances all need the COD pass ran on 64 of the 1024 wave slots.  Now one
  // per instance up to a full grid, bounded by 256 MiB of COD workspace (2 N^2 doubles per block).
  int fb_blocks = 1;
  // The list is empty for all but degenerate inputs, and a full grid of the generic kernel that finds it empty still
  // costs 6.4 us of every solve (1024 workgroups with ~60 KB of LDS each to dispatch), 64 workgroups 3.7 us less (8 or 1
  // workgroups: no further gain, measured round 4).  Block 0 of
  // a fallback launch that HAS a list leaves its length in page-locked host memory (fb_seen, device alias fb_seen_dev;
  // the host clears the word when it has read it);
  // the next launches use the full grid while a recent one was non-empty (or nothing is known yet) and
  // kFbSmallGrid workgroups otherwise.  A performance heuristic only: the persistent workgroups walk the whole list
  // whatever their number.
  static constexpr int kFbSmallGrid = 64, kFbFullFor = 8;
  volatile int *fb_seen = nullptr;
  int *fb_seen_dev = nullptr;
  int fb_full_left = kFbFullFor;
  long long work_stride = 0;
  // block -> instance order of the one-wave kernel (fccqp_set_dispatch_order): longest-first from the previous solve's
  // iteration counts, built on the device in front of the launch
  int order_mode = FCCQP_ORDER_HASH;
  int *dorder = nullptr;
  // two-launch forms (FCCQP_KERNEL_FUSED_SPLIT, FCCQP_KERNEL_COOP_SPLIT): rows of rho G + x_c, allocated on first use
  // (a handle's shape has at most one of the two forms in use at a time; the stride is the larger need seen so far)
  double *dsplit = nullptr;
  long long split_stride = 0;
  // -DFCCQP_TICKET_DISPATCH experiment (block_instance, fccqp_device.h): a counter that only grows and the
  // ticket the next launch starts at
  unsigned *dticket = nullptr;
  unsigned ticket_base = 0;
  // run-time compiled specialised kernel (fccqp_jit.cpp) for a shape that is not compiled in: jit_kind is
  // FCCQP_KERNEL_FUSED / FCCQP_KERNEL_COOP once fccqp_create has one (0: none); the float-storage instantiation is
  // compiled on the first fp32-storage solve
  int jit_kind = 0;
  fccqp::jit::Kernel jit_f64{}, jit_f32{};
  bool jit_f32_tried = false, jit_f32_ok = false;
  std::string err;
};

namespace {

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

int fail(fccqp_solver *s, int code, const char *what, hipError_t e = hipSuccess) {
  if (s) {
    char buf[512];
    if (e != hipSuccess)
      std::snprintf(buf, sizeof buf, "%s: %s (%s)", what, hipGetErrorName(e), hipGetErrorString(e));
    else
      std::snprintf(buf, sizeof buf, "%s", what);
    s->err = buf;
  }
  return code;
}

#define HIP_TRY(s, call)                                            \
  do {                                                              \
    hipError_t e_ = (call);                                         \
    if (e_ != hipSuccess) return fail((s), FCCQP_ERR_HIP, #call, e_); \
  } while (0)

// FCCQP_DEBUG_POISON=1 in the environment fills every fresh device allocation with 0xFF bytes (NaN
// as double): a kernel that reads memory it has not written then shows up as NaN in z instead of
// depending on what the allocator handed back.
bool poison_allocations() {
  static const bool on = [] { const char *e = std::getenv("FCCQP_DEBUG_POISON"); return e && e[0] == '1'; }();
  return on;
}

template <class T>
hipError_t dmalloc(T **p, size_t count) {
  const size_t bytes = (count ? count : 1) * sizeof(T);
  hipError_t e = hipMalloc(reinterpret_cast<void **>(p), bytes);
  if (e == hipSuccess && poison_allocations()) {
    // (hipMemset on device memory returns before the fill has run, and the solver's own stream is a non-blocking one
    // that does not wait for the null stream: without the wait the fill can land ON TOP of the first copy into the
    // block -- the randomised API run with poisoning showed exactly that, NaN inputs in one solve in 5 000)
    e = hipMemset(*p, 0xFF, bytes);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  }
  return e;
}

size_t n_mu(const fccqp_solver *s) { return (size_t)(s->nc / 3); }

// host-pointer solves of at most this many bytes go through the page-locked mirror blocks
constexpr size_t kPinnedMirrorBytes = 1u << 20;
// relative tolerance of the equality-residual self-check (fccqp_check.hip)
constexpr double kSelfCheckTol = 1e-6;
// FCCQP_KERNEL_AUTO takes the two-launch form of the cooperative kernel from this many instances per compute unit
// (the one-launch kernel runs one QP per CU at a time: below two rounds the second launch's ramp is not repaid)
constexpr int kCoopSplitMinBatchPerCu = 2;
size_t align256(size_t v) { return (v + 255u) & ~size_t(255); }

// ---- transfers between CALLER memory and the device ---------------------------------------------------------------
// For a copy of 1 MiB or more out of (or into) pageable memory the HIP runtime page-locks the CALLER's pages on the fly
// and lets a copy kernel read them in place ("HSA Copy Using Pinned resource" in its own log, tools/pageable_copy_probe.py;
// smaller copies go through the runtime's staging buffers), and it keeps such locks cached by address.  Both process
// aborts this library has seen -- one of eleven runs of the GPU test suite in round 4, one of six in round 5: "Memory access
// fault by GPU node-2 on address 0x64bef7d16000", an address in the middle of the process's malloc heap -- were host-pointer
// solves whose Q array is >= 1 MiB, out of numpy arrays that had been allocated and freed many times over in that process:
// the only way a GPU ever sees a heap address here is that path.  What exactly goes stale inside the runtime (the cached
// lock of pages the allocator has since returned to the kernel is the candidate) was not reproduced in isolation
// (tools/pageable_churn_stress.py: 3000 solves clean), so the remedy does not depend on it: memory that is not page-locked
// already is copied by the CPU into / out of two page-locked bounce buffers of the handle, and only those are handed to the
// DMA engines.  The copy out of the caller's arrays is complete when a solve entry point returns, whatever the stream is
// still doing.  Page-locked caller memory (hipHostMalloc / hipHostRegister: the fccqp_solve_async contract) is used in place.
constexpr size_t kBounceBytes = size_t(8) << 20;

// the CPU side of a bounce: one core's memcpy runs at ~26 GB/s here, half of what the DMA engines take over PCIe, so a
// piece of 1 MiB and more is copied by up to four threads (std::async: ~20 us to start one, against the ~80 us a megabyte
// takes)
void bounce_memcpy(void *dst, const void *src, size_t bytes) {
  const int parts = (bytes >= (size_t(4) << 20)) ? 4 : ((bytes >= (size_t(1) << 20)) ? 2 : 1);
  if (parts == 1) {
    std::memcpy(dst, src, bytes);
    return;
  }
  const size_t step = (bytes / (size_t)parts) & ~size_t(4095);
  std::future<void> rest[3];
  for (int p = 1; p < parts; ++p) {
    const size_t off = step * (size_t)p, len = (p + 1 == parts) ? bytes - off : step;
    rest[p - 1] = std::async(std::launch::async, [=] { std::memcpy(static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, len); });
  }
  std::memcpy(dst, src, step);
  for (int p = 1; p < parts; ++p) rest[p - 1].wait();
}

bool page_locked(const void *p) {
  hipPointerAttribute_t a;
  const hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

int ensure_bounce(fccqp_solver *s) {
  if (s->bounce[1]) return FCCQP_OK;
  for (int k = 0; k < 2; ++k) {
    if (!s->bounce[k]) HIP_TRY(s, hipHostMalloc(reinterpret_cast<void **>(&s->bounce[k]), kBounceBytes, hipHostMallocDefault));
    if (!s->bounce_ev[k]) HIP_TRY(s, hipEventCreateWithFlags(&s->bounce_ev[k], hipEventDisableTiming));
  }
  return FCCQP_OK;
}

// the next bounce buffer, free of the DMA that last used it
int take_bounce(fccqp_solver *s, int *k_out) {
  const int rc = ensure_bounce(s);
  if (rc != FCCQP_OK) return rc;
  const int k = s->bounce_next;
  s->bounce_next ^= 1;
  if (s->bounce_busy[k]) {
    HIP_TRY(s, hipEventSynchronize(s->bounce_ev[k]));
    s->bounce_busy[k] = false;
  }
  *k_out = k;
  return FCCQP_OK;
}

// host -> device on `st`, `rows` rows of `row_bytes` (source rows `src_pitch` apart, destination rows dense).  Returns when
// the caller's memory has been read; the DMA out of the bounce buffers may still be running.
int copy_in_rows(fccqp_solver *s, void *dst, const void *src, size_t row_bytes, size_t src_pitch, size_t rows, hipStream_t st) {
  if (!row_bytes || !rows) return FCCQP_OK;
  const char *from = static_cast<const char *>(src);
  char *to = static_cast<char *>(dst);
  if (page_locked(src)) {
    if (src_pitch == row_bytes) HIP_TRY(s, hipMemcpyAsync(to, from, row_bytes * rows, hipMemcpyHostToDevice, st));
    else HIP_TRY(s, hipMemcpy2DAsync(to, row_bytes, from, src_pitch, row_bytes, rows, hipMemcpyHostToDevice, st));
    return FCCQP_OK;
  }
  if (src_pitch == row_bytes) {              // one dense run: cut into bounce-sized pieces
    const size_t bytes = row_bytes * rows;
    for (size_t off = 0; off < bytes; off += kBounceBytes) {
      const size_t len = (bytes - off < kBounceBytes) ? bytes - off : kBounceBytes;
      int k = 0;
      const int rc = take_bounce(s, &k);
      if (rc != FCCQP_OK) return rc;
      bounce_memcpy(s->bounce[k], from + off, len);
      HIP_TRY(s, hipMemcpyAsync(to + off, s->bounce[k], len, hipMemcpyHostToDevice, st));
      HIP_TRY(s, hipEventRecord(s->bounce_ev[k], st));
      s->bounce_busy[k] = true;
    }
    return FCCQP_OK;
  }
  if (row_bytes > kBounceBytes) return fail(s, FCCQP_ERR_INVALID_ARG, "row longer than a bounce buffer");
  const size_t per = kBounceBytes / row_bytes;
  for (size_t r0 = 0; r0 < rows; r0 += per) {
    const size_t nr = (rows - r0 < per) ? rows - r0 : per;
    int k = 0;
    const int rc = take_bounce(s, &k);
    if (rc != FCCQP_OK) return rc;
    for (size_t r = 0; r < nr; ++r) std::memcpy(s->bounce[k] + r * row_bytes, from + (r0 + r) * src_pitch, row_bytes);
    HIP_TRY(s, hipMemcpyAsync(to + r0 * row_bytes, s->bounce[k], nr * row_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(s, hipEventRecord(s->bounce_ev[k], st));
    s->bounce_busy[k] = true;
  }
  return FCCQP_OK;
}
int copy_in(fccqp_solver *s, void *dst, const void *src, size_t bytes, hipStream_t st) {
  return copy_in_rows(s, dst, src, bytes, bytes, 1, st);
}

// device -> host on `st`; returns when the caller's memory holds the data.  (The caller of this function has waited for
// whatever produces `src`, or enqueued this behind it on `st`.)
int copy_out(fccqp_solver *s, void *dst, const void *src, size_t bytes, hipStream_t st) {
  if (!bytes) return FCCQP_OK;
  char *to = static_cast<char *>(dst);
  const char *from = static_cast<const char *>(src);
  if (page_locked(dst)) {
    HIP_TRY(s, hipMemcpyAsync(to, from, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(s, hipStreamSynchronize(st));
    return FCCQP_OK;
  }
  // piece i + 1 is on its way into the other bounce buffer while the CPU copies piece i out of this one
  size_t issued = 0, drained = 0;
  int slot[2] = {0, 0};
  size_t len_of[2] = {0, 0};
  int head = 0, tail = 0, inflight = 0;
  while (drained < bytes) {
    while (inflight < 2 && issued < bytes) {
      const size_t len = (bytes - issued < kBounceBytes) ? bytes - issued : kBounceBytes;
      int k = 0;
      const int rc = take_bounce(s, &k);
      if (rc != FCCQP_OK) return rc;
      HIP_TRY(s, hipMemcpyAsync(s->bounce[k], from + issued, len, hipMemcpyDeviceToHost, st));
      HIP_TRY(s, hipEventRecord(s->bounce_ev[k], st));
      s->bounce_busy[k] = true;
      slot[head] = k; len_of[head] = len; head ^= 1;
      issued += len; ++inflight;
    }
    const int k = slot[tail];
    HIP_TRY(s, hipEventSynchronize(s->bounce_ev[k]));
    s->bounce_busy[k] = false;
    bounce_memcpy(to + drained, s->bounce[k], len_of[tail]);
    drained += len_of[tail];
    tail ^= 1; --inflight;
  }
  return FCCQP_OK;
}

int ensure_staging(fccqp_solver *s) {
  if (s->din_block) return FCCQP_OK;
  const size_t B = (size_t)s->batch, n = (size_t)s->n, m = (size_t)s->m;
  const size_t counts[7] = {B * n * n, B * n, B * m * n, B * m, B * n_mu(s), B * n, B * n};
  size_t off = 0;
  for (int i = 0; i < 7; ++i) {
    s->in_off[i] = off;
    off += align256(sizeof(double) * (counts[i] ? counts[i] : 1));
  }
  s->in_bytes = off;
  HIP_TRY(s, dmalloc(&s->din_block, off));
  double **dst[7] = {&s->dQ, &s->db, &s->dA, &s->dbeq, &s->dmu, &s->dlb, &s->dub};
  for (int i = 0; i < 7; ++i) *dst[i] = reinterpret_cast<double *>(s->din_block + s->in_off[i]);
  if (off <= kPinnedMirrorBytes) {
    // (failure to page-lock is not an error: the solve falls back to the seven separate copies)
    if (hipHostMalloc(reinterpret_cast<void **>(&s->hin_block), off, hipHostMallocDefault) != hipSuccess) s->hin_block = nullptr;
    void *dev_alias = nullptr;
    if (s->hin_block && hipHostGetDevicePointer(&dev_alias, s->hin_block, 0) == hipSuccess) s->hin_dev = static_cast<char *>(dev_alias);
    else s->hin_dev = nullptr;
    (void)hipGetLastError();
  }
  return FCCQP_OK;
}

// f64 buffers for the small vectors of the fp32-storage entry points
int ensure_vec_staging(fccqp_solver *s) {
  if (s->wb) return FCCQP_OK;
  const size_t B = (size_t)s->batch, n = (size_t)s->n, m = (size_t)s->m;
  HIP_TRY(s, dmalloc(&s->wb, B * n));
  HIP_TRY(s, dmalloc(&s->wbeq, B * m));
  HIP_TRY(s, dmalloc(&s->wmu, B * n_mu(s)));
  HIP_TRY(s, dmalloc(&s->wlb, B * n));
  HIP_TRY(s, dmalloc(&s->wub, B * n));
  return FCCQP_OK;
}

// host wait for the last solve of the handle
hipError_t wait_last(fccqp_solver *s) {
  return (s->solved && !s->quiesced) ? hipEventSynchronize(s->ev1) : hipSuccess;
}

// The handle's buffers (input staging, widened vectors, warm-start state, results, fallback list) are shared by
// every solve on it.  Called by every solve entry point BEFORE its first write to any of them: work enqueued on
// `stream` from here on runs after the previous solve's kernels (wherever they were enqueued) and after a result
// copy that fccqp_get_z_async still has in flight on the solver's own stream.
int order_after_previous(fccqp_solver *s, hipStream_t stream) {
  if (s->solved && !s->quiesced && stream != s->last_stream) HIP_TRY(s, hipStreamWaitEvent(stream, s->ev1, 0));
  if (s->copy_pending && stream != s->stream) HIP_TRY(s, hipStreamWaitEvent(stream, s->ev2, 0));
  return FCCQP_OK;
}

int enqueue_impl(fccqp_solver *s, const double *dQ, const double *db, const double *dA,
                 const double *dbeq, const double *dmu, const double *dlb, const double *dub,
                 hipStream_t stream, bool qa_f32, bool defer_fallback) {
  s->hres_fresh = false;
  s->fb_deferred = false;
  s->quiesced = false;
  fccqp::KernelArgs a{};
  a.n = s->n; a.m = s->m; a.nc = s->nc; a.s = s->s; a.batch = s->batch;
  a.warm_start = s->warm_start ? 1 : 0;
  a.max_iter = s->opt.max_iter; a.rho = s->opt.rho;
  a.eps_fcone = s->opt.eps_fcone; a.eps_bound = s->opt.eps_bound;
  a.Q = dQ; a.b = db; a.A = dA; a.b_eq = dbeq; a.mu = dmu; a.lb = dlb; a.ub = dub;
  a.x = s->dx; a.mu_x = s->dmux; a.mu_lc = s->dmulc;
  a.n_iter = s->dniter; a.flags = s->dflags; a.metrics = s->dmetrics;
  a.factor_ticks = s->dticks; a.phase_ticks = s->dphase; a.workspace = s->dwork; a.workspace_stride = s->work_stride;
  // the presolve runs on a cold start, and on a warm one for an equality-only instance
  // (fcc_qp.cpp:159), which needs nc == 0
  const bool fb = s->fallback && (!s->warm_start || s->nc == 0);
  int *cnt = s->dfb_count + (s->solve_seq & 1u), *cnt_next = s->dfb_count + ((s->solve_seq + 1u) & 1u);
  a.fb_list = fb ? s->dfb_list : nullptr; a.fb_count = fb ? cnt : nullptr;
  a.inst_list = nullptr; a.inst_count = nullptr; a.next_count = nullptr;
  a.cod_ws = s->dcod_ws; a.cod_stride = s->cod_stride;

  // kernel choice.  AUTO: the one-wave fused kernel where the shape has one, else the cooperative kernel, else the
  // generic one.  (Where a shape has both specialised kernels -- the walking shape -- four waves per QP are no
  // faster than one even when the chip is otherwise idle: 92.4 against 90.5 us cold with 100 iterations, 43.7
  // against 47.3 us warm with the paper's 15, tools/coop_bench.py: the diagonal tiles, the barriers and the
  // dependent chain of an ADMM iteration are serial either way.)
  // (a run-time compiled kernel stands in for the built-in one of its kind; its float-storage instantiation is
  // compiled on first use, and a shape whose float kernel cannot be had runs that solve on the generic kernel)
  bool jit_ready = s->jit_kind != 0;
  if (jit_ready && qa_f32) {
    if (!s->jit_f32_tried) {
      s->jit_f32_tried = true;
      std::string why;
      s->jit_f32_ok = fccqp::jit::acquire(s->jit_kind == FCCQP_KERNEL_FUSED ? fccqp::jit::kFused : fccqp::jit::kCoop, s->n, s->m,
                                          s->nc, s->s, true, &s->jit_f32, &why);
      if (!s->jit_f32_ok) s->err = "run-time compilation of the fp32-storage kernel failed (" + why + "): generic kernel";
    }
    jit_ready = s->jit_f32_ok;
  }
  const bool can_fuse = (fccqp::fused_supported(s->n, s->m, s->nc, s->s) && fccqp::fused_pointers_ok(a)) ||
                        (jit_ready && s->jit_kind == FCCQP_KERNEL_FUSED);
  const bool can_coop = fccqp::coop_supported(s->n, s->m, s->nc, s->s) || (jit_ready && s->jit_kind == FCCQP_KERNEL_COOP);
  int kernel = s->kernel_request;
  if (kernel == FCCQP_KERNEL_FUSED_SPLIT && s->jit_kind != 0)
    return fail(s, FCCQP_ERR_UNSUPPORTED, "the two-launch form exists only for shapes compiled into the library");
  if ((kernel == FCCQP_KERNEL_FUSED || kernel == FCCQP_KERNEL_FUSED_SPLIT) && !can_fuse)
    return fail(s, FCCQP_ERR_UNSUPPORTED, "no fused kernel for this shape");
  if (kernel == FCCQP_KERNEL_COOP && !can_coop)
    return fail(s, FCCQP_ERR_UNSUPPORTED, "no cooperative kernel for this shape");
  // (the two-launch form of the cooperative kernel: listed shapes with more than 64 variables)
  const long long coop_split_need = (s->jit_kind == 0) ? fccqp::coop_split_workspace_doubles(s->n, s->m, s->nc, s->s) : 0;
  if (kernel == FCCQP_KERNEL_COOP_SPLIT && coop_split_need == 0)
    return fail(s, FCCQP_ERR_UNSUPPORTED, "no two-launch cooperative kernel for this shape (listed shapes with num_vars > 64)");
  if (kernel == FCCQP_KERNEL_AUTO) {
    if (can_fuse) kernel = FCCQP_KERNEL_FUSED;
    else if (can_coop) kernel = (coop_split_need > 0 && s->batch >= kCoopSplitMinBatchPerCu * s->compute_units)
                                    ? FCCQP_KERNEL_COOP_SPLIT : FCCQP_KERNEL_COOP;
    else kernel = FCCQP_KERNEL_GENERIC;
  }
  const bool want_fused = kernel == FCCQP_KERNEL_FUSED || kernel == FCCQP_KERNEL_FUSED_SPLIT;
#ifdef FCCQP_TICKET_DISPATCH
  // tickets only where a launch runs in several rounds (all workgroups of a smaller one are resident at once:
  // 4 per CU of the fused kernel, at least 1 of the generic one)
  const bool tickets = s->batch > (want_fused ? 1024 : 256);
  a.ticket = tickets ? s->dticket : nullptr;
  a.ticket_base = s->ticket_base;
#endif
  // ADMM waves per SIMD of the two-launch form, 0: one launch
  int split = 0;
  if (kernel == FCCQP_KERNEL_FUSED_SPLIT) {
    static const int waves = [] { const char *e = std::getenv("FCCQP_SPLIT_WAVES"); return (e && e[0] == '3') ? 3 : 2; }();
    if (!s->dsplit) {
      s->split_stride = fccqp::fused_split_workspace_doubles(s->n);
      HIP_TRY(s, dmalloc(&s->dsplit, (size_t)s->batch * (size_t)s->split_stride));
    }
    split = waves;
  }
  // workgroups (QPs) per compute unit in the second launch of the cooperative kernel's two-launch form, 0: one launch
  int coop_split = 0;
  if (kernel == FCCQP_KERNEL_COOP_SPLIT) {
    static const int wgs = [] { const char *e = std::getenv("FCCQP_COOP_SPLIT_WGS"); return (e && e[0] == '2') ? 2 : 3; }();
    if (!s->dsplit || s->split_stride < coop_split_need) {
      if (s->dsplit) { HIP_TRY(s, hipStreamSynchronize(stream)); (void)hipFree(s->dsplit); s->dsplit = nullptr; }
      s->split_stride = coop_split_need;
      HIP_TRY(s, dmalloc(&s->dsplit, (size_t)s->batch * (size_t)s->split_stride));
    }
    coop_split = wgs;
  }
  // (ordering against the previous solve: order_after_previous, called by the entry points)
  // (defer_fallback is the synchronous small-batch path: its caller waits on the host and records no events)
  s->timed = s->kernel_timing && !defer_fallback;
  if (s->timed) HIP_TRY(s, hipEventRecord(s->ev0, stream));
  hipError_t e;
  if (jit_ready && kernel == s->jit_kind) {
    s->kernel_used = kernel;
    e = fccqp::jit::launch(qa_f32 ? s->jit_f32 : s->jit_f64, a, stream);
  } else if (kernel == FCCQP_KERNEL_COOP_SPLIT) {
    s->kernel_used = FCCQP_KERNEL_COOP_SPLIT;
    fccqp::KernelArgs ac = a;
    ac.workspace = s->dsplit; ac.workspace_stride = s->split_stride;
    e = fccqp::launch_coop_split(ac, stream, qa_f32, coop_split);
  } else if (kernel == FCCQP_KERNEL_COOP) {
    s->kernel_used = FCCQP_KERNEL_COOP;
    e = fccqp::launch_coop(a, stream, qa_f32);
  } else if (want_fused) {
    s->kernel_used = split ? FCCQP_KERNEL_FUSED_SPLIT : FCCQP_KERNEL_FUSED;
    fccqp::KernelArgs af = a;
    if (split) { af.workspace = s->dsplit; af.workspace_stride = s->split_stride; }
    // a chain's next solve, longest instance first: only where a launch runs in several rounds, and only with the
    // iteration counts of a previous solve on this handle to go by (they are still in place: this launch overwrites them)
    if (!split && s->order_mode == FCCQP_ORDER_PREVIOUS_ITERATIONS && s->solved && s->batch >= 8 * s->compute_units) {
      if (!s->dorder) HIP_TRY(s, hipMalloc(reinterpret_cast<void **>(&s->dorder), sizeof(int) * (size_t)s->batch));
      e = fccqp::launch_order_by_iterations(s->dniter, s->batch, s->dorder, stream);
      if (e != hipSuccess) return fail(s, FCCQP_ERR_HIP, "dispatch-order kernel launch", e);
      af.inst_list = s->dorder;
    }
    e = fccqp::launch_fused(af, stream, qa_f32, split);
  } else {
    s->kernel_used = FCCQP_KERNEL_GENERIC;
    if (!s->dwork && s->n > 0 && s->n <= 128) {
      // LDS-resident shapes: per-instance staging for rho*G (n*n doubles), through which the
      // generic kernel switches long-running instances from triangular sweeps to a mat-vec
      // (fccqp_generic.hip); allocated on the first generic launch only
      // + 2 (n + m): [-b; b_eq] in, x_c out for the one-wave MFMA build
      const size_t nn = (size_t)s->n * (size_t)s->n + 2 * ((size_t)s->n + (size_t)s->m);
      if (dmalloc(&s->dwork, (size_t)s->batch * nn) == hipSuccess) {
        s->work_stride = (long long)nn;
        a.workspace = s->dwork; a.workspace_stride = s->work_stride;
      } else {
        (void)hipGetLastError();   // no staging: the kernel keeps to the sweeps
        s->dwork = nullptr;
      }
    }
    e = fccqp::launch_generic(a, s->lds_limit, stream, qa_f32);
  }
  if (e != hipSuccess) return fail(s, FCCQP_ERR_HIP, "kernel launch", e);
#ifdef FCCQP_TICKET_DISPATCH
  if (tickets) s->ticket_base += (unsigned)s->batch;      // every workgroup of the launch draws exactly one
#endif
  if (fb) {
    // fallback launch over the (normally empty) device list: COD x0, then ADMM from it.  It also
    // zeroes the other counter, which the next solve's main kernel will append to.
    int blocks = s->fb_blocks;
    if (s->fb_seen) {
      if (*s->fb_seen != 0) {
        s->fb_full_left = fccqp_solver::kFbFullFor;
        *s->fb_seen = 0;          // (the device only ever writes non-zero lengths; one that lands right now is seen next time)
      } else if (s->fb_full_left > 0) {
        --s->fb_full_left;
      }
      if (s->fb_full_left == 0 && blocks > fccqp_solver::kFbSmallGrid) blocks = fccqp_solver::kFbSmallGrid;
    }
    fccqp::KernelArgs b = a;
    b.fb_list = nullptr; b.fb_count = s->fb_seen_dev;        // (list mode: where block 0 reports the list's length)
#ifdef FCCQP_TICKET_DISPATCH
    b.ticket = nullptr;
#endif
    b.inst_list = s->dfb_list; b.inst_count = cnt; b.next_count = cnt_next;
    if (defer_fallback) {
      // the caller looks at the flags first (run_deferred_fallback): with nothing appended the counter is still zero
      // and the pair of counters needs no turn
      s->pending_fb = b; s->pending_fb_f32 = qa_f32; s->fb_deferred = true;
      e = hipSuccess;
    } else {
      e = fccqp::launch_generic_list(b, s->lds_limit, blocks, stream, qa_f32);
    }
    if (e != hipSuccess) {
      // the main kernel may have appended to `cnt`, and nothing will consume it: start the next
      // solve from a clean pair of counters
      (void)hipMemsetAsync(s->dfb_count, 0, 2 * sizeof(int), stream);
      return fail(s, FCCQP_ERR_HIP, "fallback kernel launch", e);
    }
    if (!defer_fallback) s->solve_seq++;
  }
  if (s->self_check) {
    // behind the main kernel and (when it is not deferred) the fallback pass, in front of the end event and of any
    // copy of the results; a deferred fallback pass is followed by a second check (run_deferred_fallback)
    s->check_args = a;
    e = fccqp::launch_eq_residual_check(a, stream, qa_f32, kSelfCheckTol);
    if (e != hipSuccess) return fail(s, FCCQP_ERR_HIP, "self-check launch", e);
  }
  if (!defer_fallback) HIP_TRY(s, hipEventRecord(s->ev1, stream));
  s->last_stream = stream;
  s->last_was_warm = s->warm_start;
  s->last_was_f32 = qa_f32;
  s->solved = true;
  return FCCQP_OK;
}

// A solve that failed part-way may have a kernel running on `stream`, a fallback list nobody is going to consume and
// no end event for the next solve to wait on.  Leave the handle as before its first solve (GetSolution answers
// FCCQP_ERR_NOT_SOLVED; the warm-start state is whatever the kernels left): the stream drained, both list counters
// cleared, nothing pending.  Keeps the message of the failure.
void reset_after_failed_solve(fccqp_solver *s, hipStream_t stream) {
  const std::string msg = s->err;
  if (s->dfb_count) (void)hipMemsetAsync(s->dfb_count, 0, 2 * sizeof(int), stream);
  (void)hipStreamSynchronize(stream);
  (void)hipGetLastError();
  s->bounce_busy[0] = s->bounce_busy[1] = false;
  s->fb_deferred = false;
  s->hres_fresh = false;
  s->solved = false;
  s->quiesced = true;
  s->host_device_ms = 0.f;
  s->err = msg;
}

int enqueue(fccqp_solver *s, const double *dQ, const double *db, const double *dA,
            const double *dbeq, const double *dmu, const double *dlb, const double *dub,
            hipStream_t stream, bool qa_f32 = false, bool defer_fallback = false) {
  const int rc = enqueue_impl(s, dQ, db, dA, dbeq, dmu, dlb, dub, stream, qa_f32, defer_fallback);
  if (rc != FCCQP_OK) reset_after_failed_solve(s, stream);
  return rc;
}

// Host wait for a short solve: poll the stream instead of sleeping on its completion signal (the blocking wait
// costs a wake-up of 10-20 us, a fifth of a one-QP solve); falls back to the blocking wait after ~2 ms.
hipError_t spin_synchronize(hipStream_t stream) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(stream);
    if (q != hipErrorNotReady) return q;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) return hipStreamSynchronize(stream);
  }
}

// the fallback-list launch that enqueue(.., defer_fallback = true) left out
int run_deferred_fallback(fccqp_solver *s, hipStream_t stream) {
  if (!s->fb_deferred) return FCCQP_OK;
  s->fb_deferred = false;
  hipError_t e = fccqp::launch_generic_list(s->pending_fb, s->lds_limit, s->fb_blocks, stream, s->pending_fb_f32);
  if (e != hipSuccess) {
    (void)hipMemsetAsync(s->dfb_count, 0, 2 * sizeof(int), stream);
    return fail(s, FCCQP_ERR_HIP, "fallback kernel launch", e);
  }
  s->solve_seq++;
  if (s->self_check) {
    e = fccqp::launch_eq_residual_check(s->check_args, stream, s->pending_fb_f32, kSelfCheckTol);
    if (e != hipSuccess) return fail(s, FCCQP_ERR_HIP, "self-check launch", e);
  }
  return FCCQP_OK;
}

// A failure after enqueue(.., defer_fallback = true): no end event was recorded and the main kernel may have appended
// to the fallback list that nobody is going to consume (reset_after_failed_solve).
int abandon_deferred(fccqp_solver *s, hipStream_t stream, const char *what, hipError_t e) {
  const int rc = fail(s, FCCQP_ERR_HIP, what, e);
  reset_after_failed_solve(s, stream);
  return rc;
}

}  // namespace

extern "C" {

int fccqp_version(void) { return FCCQP_VERSION_MAJOR * 100 + FCCQP_VERSION_MINOR; }

const char *fccqp_strerror(int code) {
  switch (code) {
    case FCCQP_OK: return "ok";
    case FCCQP_ERR_INVALID_ARG: return "invalid argument";
    case FCCQP_ERR_SHAPE: return "invalid problem sizes";
    case FCCQP_ERR_NO_DEVICE: return "no HIP device available (this library has no CPU path)";
    case FCCQP_ERR_HIP: return "HIP runtime error";
    case FCCQP_ERR_ALLOC: return "allocation failed";
    case FCCQP_ERR_FRICTION_COEFFS: return "fewer than nc/3 friction coefficients";
    case FCCQP_ERR_NOT_SOLVED: return "no solve has been run";
    case FCCQP_ERR_UNSUPPORTED: return "shape not supported by the requested kernel";
    default: return "unknown error";
  }
}

int fccqp_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

int fccqp_default_options(fccqp_options *opt) {
  if (!opt) return FCCQP_ERR_INVALID_ARG;
  *opt = fccqp_options{1000, 1e-6, 1e-3, 1e-6};
  return FCCQP_OK;
}

int fccqp_create(int num_vars, int num_equality_constraints, int nc, int lambda_c_start, int batch,
                 int device, fccqp_solver **out) {
  if (!out) return FCCQP_ERR_INVALID_ARG;
  *out = nullptr;
  // reference asserts, src/fcc_qp.cpp:29-33
  if (num_vars < 0 || num_equality_constraints < 0 || nc < 0 || nc % 3 != 0 ||
      lambda_c_start < 0 || lambda_c_start + nc > num_vars || batch < 1)
    return FCCQP_ERR_SHAPE;
  if (num_vars + num_equality_constraints > fccqp::kGenericMaxN) return FCCQP_ERR_UNSUPPORTED;
  const int ndev = fccqp_device_count();
  if (ndev <= 0 || device < 0 || device >= ndev) return FCCQP_ERR_NO_DEVICE;

  fccqp_solver *s = new (std::nothrow) fccqp_solver();
  if (!s) return FCCQP_ERR_ALLOC;
  s->n = num_vars; s->m = num_equality_constraints; s->nc = nc; s->s = lambda_c_start;
  s->batch = batch; s->device = device;
  s->self_check = num_vars + num_equality_constraints > fccqp::kSelfCheckAboveN;
  DeviceGuard g(device);
  if (!g.ok) { delete s; return FCCQP_ERR_NO_DEVICE; }

  int rc = [&]() -> int {
    hipDeviceProp_t prop;
    HIP_TRY(s, hipGetDeviceProperties(&prop, device));
    size_t lds = prop.sharedMemPerBlock;               // 64 KiB default limit
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, device) == hipSuccess &&
        (size_t)optin > lds)
      lds = (size_t)optin;                              // 160 KiB on gfx950
    s->lds_limit = lds;
    s->compute_units = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(s, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    HIP_TRY(s, hipEventCreate(&s->ev0));
    HIP_TRY(s, hipEventCreate(&s->ev1));
    HIP_TRY(s, hipEventCreateWithFlags(&s->ev2, hipEventDisableTiming));
    const size_t B = (size_t)batch, n = (size_t)num_vars;
    {
      // z (= x), metrics, factor ticks, n_iter, flags: one block, so that GetSolution is one copy
      const size_t bytes[5] = {sizeof(double) * (B * n ? B * n : 1), sizeof(double) * 4 * B, sizeof(long long) * B,
                               sizeof(int) * B, sizeof(int) * B};
      size_t off = 0;
      for (int i = 0; i < 5; ++i) {
        s->res_off[i] = off;
        off += align256(bytes[i]);
      }
      s->res_bytes = off;
      HIP_TRY(s, dmalloc(&s->dres_block, off));
      s->dx = reinterpret_cast<double *>(s->dres_block + s->res_off[0]);
      s->dmetrics = reinterpret_cast<double *>(s->dres_block + s->res_off[1]);
      s->dticks = reinterpret_cast<long long *>(s->dres_block + s->res_off[2]);
      s->dniter = reinterpret_cast<int *>(s->dres_block + s->res_off[3]);
      s->dflags = reinterpret_cast<int *>(s->dres_block + s->res_off[4]);
      if (off <= kPinnedMirrorBytes &&
          hipHostMalloc(reinterpret_cast<void **>(&s->hres_block), off, hipHostMallocDefault) != hipSuccess)
        s->hres_block = nullptr;
    }
    HIP_TRY(s, dmalloc(&s->dmux, B * n));
    HIP_TRY(s, dmalloc(&s->dmulc, B * (size_t)nc));
    // reference ctor zero-initialises x_, mu_x_, mu_lambda_c_ (src/fcc_qp.cpp:48-53)
    HIP_TRY(s, hipMemset(s->dx, 0, sizeof(double) * (B * n ? B * n : 1)));
    HIP_TRY(s, hipMemset(s->dmux, 0, sizeof(double) * (B * n ? B * n : 1)));
    HIP_TRY(s, hipMemset(s->dmulc, 0, sizeof(double) * (B * nc ? B * nc : 1)));
    HIP_TRY(s, hipMemset(s->dmetrics, 0, sizeof(double) * 4 * B));
    HIP_TRY(s, hipMemset(s->dniter, 0, sizeof(int) * B));
    HIP_TRY(s, hipMemset(s->dflags, 0, sizeof(int) * B));
    HIP_TRY(s, hipMemset(s->dticks, 0, sizeof(long long) * B));
    s->work_stride = fccqp::generic_workspace_doubles(num_vars, num_equality_constraints, nc, lds);
    if (s->work_stride > 0) HIP_TRY(s, dmalloc(&s->dwork, B * (size_t)s->work_stride));
    HIP_TRY(s, dmalloc(&s->dfb_list, B + 2));
    HIP_TRY(s, hipMemset(s->dfb_list, 0, sizeof(int) * (B + 2)));
    HIP_TRY(s, dmalloc(&s->dfb_count, 2));
    HIP_TRY(s, hipMemset(s->dfb_count, 0, 2 * sizeof(int)));
    // the fills above are enqueued on the null stream and need not have run when hipMemset returns; solves run on
    // non-blocking streams, which do not wait for it: make the zero state a fact before the handle is handed out
    HIP_TRY(s, hipStreamSynchronize(nullptr));
#ifdef FCCQP_TICKET_DISPATCH
    HIP_TRY(s, dmalloc(&s->dticket, 1));
    HIP_TRY(s, hipMemset(s->dticket, 0, sizeof(unsigned)));
#endif
    s->cod_stride = fccqp::cod_workspace_doubles_host(num_vars, num_equality_constraints);
    {
      long long blocks = batch < 1024 ? batch : 1024;
      const long long fit = (256ll << 20) / (8 * (s->cod_stride > 0 ? s->cod_stride : 1));
      if (blocks > fit) blocks = fit;
      s->fb_blocks = (int)(blocks < 1 ? 1 : blocks);
    }
    HIP_TRY(s, dmalloc(&s->dcod_ws, (size_t)s->fb_blocks * (size_t)s->cod_stride));
    {
      void *h = nullptr, *d = nullptr;
      if (hipHostMalloc(&h, sizeof(int), hipHostMallocDefault) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
        s->fb_seen = static_cast<volatile int *>(h);
        s->fb_seen_dev = static_cast<int *>(d);
        *s->fb_seen = 1;                                     // nothing known yet: full grid
      } else if (h) {
        (void)hipHostFree(h);
      }
    }
    // A shape without a compiled-in specialised kernel: instantiate one now (fccqp_jit.cpp) -- here and not in the first
    // solve, which may sit in a control loop.  One-wave kernel first; a shape it does not take, or takes only with
    // register spills, tries the cooperative kernel.  Failure is not an error: the generic kernel serves every shape.
    if (!fccqp::fused_supported(num_vars, num_equality_constraints, nc, lambda_c_start) &&
        !fccqp::coop_supported(num_vars, num_equality_constraints, nc, lambda_c_start) && fccqp::jit::enabled()) {
      std::string why_f, why_c;
      fccqp::jit::Kernel kf, kc;
      const bool qf = fccqp::jit::qualifies(fccqp::jit::kFused, num_vars, num_equality_constraints, nc, lambda_c_start, nullptr);
      const bool qc = fccqp::jit::qualifies(fccqp::jit::kCoop, num_vars, num_equality_constraints, nc, lambda_c_start, nullptr);
      const bool have_f = qf && fccqp::jit::acquire(fccqp::jit::kFused, num_vars, num_equality_constraints, nc, lambda_c_start, false, &kf, &why_f);
      bool have_c = false;
      if (have_f && kf.scratch_bytes == 0) {
        s->jit_kind = FCCQP_KERNEL_FUSED; s->jit_f64 = kf;
      } else if ((have_c = qc && fccqp::jit::acquire(fccqp::jit::kCoop, num_vars, num_equality_constraints, nc, lambda_c_start, false, &kc, &why_c)) &&
                 kc.scratch_bytes == 0) {
        s->jit_kind = FCCQP_KERNEL_COOP; s->jit_f64 = kc;
      } else if (have_f) {
        s->jit_kind = FCCQP_KERNEL_FUSED; s->jit_f64 = kf;       // (spills registers; still well ahead of the generic kernel)
      } else if (have_c) {
        s->jit_kind = FCCQP_KERNEL_COOP; s->jit_f64 = kc;        // (likewise: a few spilled registers against 3x the time)
      } else if (qf || qc) {
        s->err = "no run-time compiled kernel for this shape (" + (why_f.empty() ? why_c : why_f) + "): generic kernel";
      }
      (void)hipGetLastError();
    }
    return FCCQP_OK;
  }();
  if (rc != FCCQP_OK) {
    std::fprintf(stderr, "fccqp_create: %s\n", s->err.c_str());
    fccqp_destroy(s);
    return rc;
  }
  *out = s;
  return FCCQP_OK;
}

int fccqp_destroy(fccqp_solver *s) {
  if (!s) return FCCQP_OK;
  DeviceGuard g(s->device);
  // nothing of the handle may still be in use: the last solve (wherever it was enqueued -- its kernels write the
  // page-locked words and blocks freed below), a result copy on the solver's stream, DMA out of the bounce buffers
  if (s->ev1) (void)wait_last(s);
  if (s->copy_pending && s->ev2) (void)hipEventSynchronize(s->ev2);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  (void)hipGetLastError();
  void *ptrs[] = {s->wb, s->wbeq, s->wmu, s->wlb, s->wub, s->din_block, s->dres_block, s->dmux, s->dmulc,
                  s->dwork, s->dphase, s->dfb_list, s->dfb_count, s->dcod_ws, s->dticket, s->dsplit, s->dorder};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (s->fb_seen) (void)hipHostFree(const_cast<int *>(s->fb_seen));
  if (s->hin_block) (void)hipHostFree(s->hin_block);
  if (s->hres_block) (void)hipHostFree(s->hres_block);
  for (int k = 0; k < 2; ++k) {
    if (s->bounce[k]) (void)hipHostFree(s->bounce[k]);
    if (s->bounce_ev[k]) (void)hipEventDestroy(s->bounce_ev[k]);
  }
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->ev2) (void)hipEventDestroy(s->ev2);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
  return FCCQP_OK;
}

const char *fccqp_last_error(const fccqp_solver *s) { return s ? s->err.c_str() : "null handle"; }

int fccqp_set_options(fccqp_solver *s, const fccqp_options *opt) {
  if (!s || !opt) return FCCQP_ERR_INVALID_ARG;
  s->opt = *opt;  // set_options copies blindly, src/fcc_qp.hpp:85-87
  return FCCQP_OK;
}
int fccqp_get_options(const fccqp_solver *s, fccqp_options *opt) {
  if (!s || !opt) return FCCQP_ERR_INVALID_ARG;
  *opt = s->opt;
  return FCCQP_OK;
}
int fccqp_set_warm_start(fccqp_solver *s, int warm_start) {
  if (!s) return FCCQP_ERR_INVALID_ARG;
  s->warm_start = warm_start != 0;
  return FCCQP_OK;
}
int fccqp_set_kernel(fccqp_solver *s, int kernel) {
  if (!s || kernel < FCCQP_KERNEL_AUTO || kernel > FCCQP_KERNEL_COOP_SPLIT) return FCCQP_ERR_INVALID_ARG;
  s->kernel_request = kernel;
  return FCCQP_OK;
}
int fccqp_set_dispatch_order(fccqp_solver *s, int order) {
  if (!s || (order != FCCQP_ORDER_HASH && order != FCCQP_ORDER_PREVIOUS_ITERATIONS)) return FCCQP_ERR_INVALID_ARG;
  s->order_mode = order;
  return FCCQP_OK;
}
int fccqp_set_presolve_fallback(fccqp_solver *s, int enabled) {
  if (!s) return FCCQP_ERR_INVALID_ARG;
  s->fallback = enabled != 0;
  return FCCQP_OK;
}
int fccqp_set_kernel_timing(fccqp_solver *s, int enabled) {
  if (!s) return FCCQP_ERR_INVALID_ARG;
  s->kernel_timing = enabled != 0;
  return FCCQP_OK;
}
int fccqp_set_self_check(fccqp_solver *s, int enabled) {
  if (!s) return FCCQP_ERR_INVALID_ARG;
  s->self_check = enabled != 0;
  return FCCQP_OK;
}
int fccqp_get_kernel(const fccqp_solver *s, int *kernel_in_use) {
  if (!s || !kernel_in_use) return FCCQP_ERR_INVALID_ARG;
  if (s->solved) {   // what the last solve ran (unaligned Q / A_eq fall back to the generic kernel)
    *kernel_in_use = s->kernel_used;
    return FCCQP_OK;
  }
  const bool can_fuse = fccqp::fused_supported(s->n, s->m, s->nc, s->s) || s->jit_kind == FCCQP_KERNEL_FUSED,
             can_coop = fccqp::coop_supported(s->n, s->m, s->nc, s->s) || s->jit_kind == FCCQP_KERNEL_COOP;
  int k = s->kernel_request;
  if (k == FCCQP_KERNEL_AUTO) {
    k = can_fuse ? FCCQP_KERNEL_FUSED : (can_coop ? FCCQP_KERNEL_COOP : FCCQP_KERNEL_GENERIC);
  }
  *kernel_in_use = k;
  return FCCQP_OK;
}
int fccqp_kernel_for_shape(int num_vars, int num_equality_constraints, int nc, int lambda_c_start) {
  if (num_vars < 0 || num_equality_constraints < 0 || nc < 0 || nc % 3 != 0 || lambda_c_start < 0 ||
      lambda_c_start + nc > num_vars)
    return -FCCQP_ERR_SHAPE;
  if (num_vars + num_equality_constraints > fccqp::kGenericMaxN) return -FCCQP_ERR_UNSUPPORTED;
  if (fccqp::fused_supported(num_vars, num_equality_constraints, nc, lambda_c_start)) return FCCQP_KERNEL_FUSED;
  if (fccqp::coop_supported(num_vars, num_equality_constraints, nc, lambda_c_start)) return FCCQP_KERNEL_COOP;
  if (fccqp::jit::enabled()) {      // what AUTO runs once fccqp_create has compiled the shape's kernel
    if (fccqp::jit::qualifies(fccqp::jit::kFused, num_vars, num_equality_constraints, nc, lambda_c_start, nullptr)) return FCCQP_KERNEL_FUSED;
    if (fccqp::jit::qualifies(fccqp::jit::kCoop, num_vars, num_equality_constraints, nc, lambda_c_start, nullptr)) return FCCQP_KERNEL_COOP;
  }
  return FCCQP_KERNEL_GENERIC;
}
int fccqp_kernel_origin_for_shape(int num_vars, int num_equality_constraints, int nc, int lambda_c_start) {
  const int k = fccqp_kernel_for_shape(num_vars, num_equality_constraints, nc, lambda_c_start);
  if (k < 0) return k;
  if (fccqp::fused_supported(num_vars, num_equality_constraints, nc, lambda_c_start) ||
      fccqp::coop_supported(num_vars, num_equality_constraints, nc, lambda_c_start))
    return FCCQP_ORIGIN_BUILTIN;
  return k == FCCQP_KERNEL_GENERIC ? FCCQP_ORIGIN_GENERIC_ONLY : FCCQP_ORIGIN_JIT;
}
int fccqp_jit_prepare(int num_vars, int num_equality_constraints, int nc, int lambda_c_start) {
  const int k = fccqp_kernel_for_shape(num_vars, num_equality_constraints, nc, lambda_c_start);
  if (k < 0) return k;
  if (fccqp::fused_supported(num_vars, num_equality_constraints, nc, lambda_c_start)) return FCCQP_KERNEL_FUSED;
  if (fccqp::coop_supported(num_vars, num_equality_constraints, nc, lambda_c_start)) return FCCQP_KERNEL_COOP;
  std::string why;
  if (fccqp::jit::qualifies(fccqp::jit::kFused, num_vars, num_equality_constraints, nc, lambda_c_start, nullptr) &&
      fccqp::jit::prepare(fccqp::jit::kFused, num_vars, num_equality_constraints, nc, lambda_c_start, false, &why))
    return FCCQP_KERNEL_FUSED;
  if (fccqp::jit::qualifies(fccqp::jit::kCoop, num_vars, num_equality_constraints, nc, lambda_c_start, nullptr) &&
      fccqp::jit::prepare(fccqp::jit::kCoop, num_vars, num_equality_constraints, nc, lambda_c_start, false, &why))
    return FCCQP_KERNEL_COOP;
  if (!why.empty()) std::fprintf(stderr, "fccqp_jit_prepare(%d, %d, %d, %d): %s\n", num_vars, num_equality_constraints, nc, lambda_c_start, why.c_str());
  return -FCCQP_ERR_UNSUPPORTED;
}
int fccqp_get_kernel_origin(const fccqp_solver *s, int *origin) {
  if (!s || !origin) return FCCQP_ERR_INVALID_ARG;
  if (s->jit_kind != 0) *origin = FCCQP_ORIGIN_JIT;
  else if (fccqp::fused_supported(s->n, s->m, s->nc, s->s) || fccqp::coop_supported(s->n, s->m, s->nc, s->s)) *origin = FCCQP_ORIGIN_BUILTIN;
  else *origin = FCCQP_ORIGIN_GENERIC_ONLY;
  return FCCQP_OK;
}
int fccqp_set_jit(int enabled) {
  fccqp::jit::set_enabled(enabled != 0);
  return FCCQP_OK;
}
int fccqp_debug_jit_forget(void) {
  fccqp::jit::forget();
  return FCCQP_OK;
}
int fccqp_get_shape(const fccqp_solver *s, int *num_vars, int *num_eq, int *nc, int *lambda_c_start,
                    int *batch, int *device) {
  if (!s) return FCCQP_ERR_INVALID_ARG;
  if (num_vars) *num_vars = s->n;
  if (num_eq) *num_eq = s->m;
  if (nc) *nc = s->nc;
  if (lambda_c_start) *lambda_c_start = s->s;
  if (batch) *batch = s->batch;
  if (device) *device = s->device;
  return FCCQP_OK;
}

namespace {
int solve_host(fccqp_solver *s, const double *Q, const double *b, const double *A_eq,
               const double *b_eq, const double *friction_coeffs, int n_friction,
               const double *lb, const double *ub, bool wait);
}

int fccqp_solve(fccqp_solver *s, const double *Q, const double *b, const double *A_eq,
                const double *b_eq, const double *friction_coeffs, int n_friction,
                const double *lb, const double *ub) {
  return solve_host(s, Q, b, A_eq, b_eq, friction_coeffs, n_friction, lb, ub, true);
}

int fccqp_solve_async(fccqp_solver *s, const double *Q, const double *b, const double *A_eq,
                      const double *b_eq, const double *friction_coeffs, int n_friction,
                      const double *lb, const double *ub) {
  if (s && n_friction != (int)n_mu(s)) return fail(s, FCCQP_ERR_FRICTION_COEFFS, "fccqp_solve_async: n_friction must be nc/3");
  return solve_host(s, Q, b, A_eq, b_eq, friction_coeffs, n_friction, lb, ub, false);
}

int fccqp_host_register(void *ptr, size_t bytes) {
  if (!ptr || !bytes) return FCCQP_ERR_INVALID_ARG;
  return hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess ? FCCQP_OK : FCCQP_ERR_HIP;
}
int fccqp_host_unregister(void *ptr) {
  if (!ptr) return FCCQP_ERR_INVALID_ARG;
  return hipHostUnregister(ptr) == hipSuccess ? FCCQP_OK : FCCQP_ERR_HIP;
}

int fccqp_get_z_async(fccqp_solver *s, double *z_pinned) {
  if (!s || !z_pinned) return FCCQP_ERR_INVALID_ARG;
  if (!s->solved) return fail(s, FCCQP_ERR_NOT_SOLVED, "GetSolution before Solve");
  DeviceGuard g(s->device);
  // behind the last solve, wherever it was enqueued
  if (!s->quiesced && s->last_stream != s->stream) HIP_TRY(s, hipStreamWaitEvent(s->stream, s->ev1, 0));
  if (!page_locked(z_pinned)) {
    // not what the contract asks for: the copy is done here, through the bounce buffers, and is complete on return
    const int rc = copy_out(s, z_pinned, s->dx, sizeof(double) * (size_t)s->batch * (size_t)s->n, s->stream);
    if (rc != FCCQP_OK) return rc;
    s->copy_pending = false;
    return FCCQP_OK;
  }
  HIP_TRY(s, hipMemcpyAsync(z_pinned, s->dx, sizeof(double) * (size_t)s->batch * (size_t)s->n,
                            hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipEventRecord(s->ev2, s->stream));
  s->copy_pending = true;
  return FCCQP_OK;
}

namespace {
int solve_host(fccqp_solver *s, const double *Q, const double *b, const double *A_eq,
               const double *b_eq, const double *friction_coeffs, int n_friction,
               const double *lb, const double *ub, bool wait) {
  if (!s) return FCCQP_ERR_INVALID_ARG;
  const size_t B = (size_t)s->batch, n = (size_t)s->n, m = (size_t)s->m, nm = n_mu(s);
  if ((n && (!Q || !b || !lb || !ub)) || (m && n && !A_eq) || (m && !b_eq) || (nm && !friction_coeffs))
    return fail(s, FCCQP_ERR_INVALID_ARG, "null input pointer");
  if (n_friction < (int)nm) return fail(s, FCCQP_ERR_FRICTION_COEFFS, "friction_coeffs too short");
  const auto t0 = std::chrono::high_resolution_clock::now();
  DeviceGuard g(s->device);
  if (!g.ok) return fail(s, FCCQP_ERR_NO_DEVICE, "hipSetDevice failed");
  int rc = ensure_staging(s);
  if (rc != FCCQP_OK) return rc;
  hipStream_t st = s->stream;
  rc = order_after_previous(s, st);
  if (rc != FCCQP_OK) return rc;
  if (s->hin_block) {
    // small batch: gather into the page-locked mirror, one copy.  (The previous solve has finished reading
    // the mirror's device twin only when its kernel is done; the mirror itself is free once its copy is.)
    HIP_TRY(s, wait_last(s));
    char *h = s->hin_block;
    std::memcpy(h + s->in_off[0], Q, sizeof(double) * B * n * n);
    std::memcpy(h + s->in_off[1], b, sizeof(double) * B * n);
    if (m * n) std::memcpy(h + s->in_off[2], A_eq, sizeof(double) * B * m * n);
    if (m) std::memcpy(h + s->in_off[3], b_eq, sizeof(double) * B * m);
    for (size_t i = 0; i < B && nm; ++i)    // keep the first nc/3 of every instance's row
      std::memcpy(h + s->in_off[4] + sizeof(double) * nm * i, friction_coeffs + (size_t)n_friction * i, sizeof(double) * nm);
    std::memcpy(h + s->in_off[5], lb, sizeof(double) * B * n);
    std::memcpy(h + s->in_off[6], ub, sizeof(double) * B * n);
    // The kernel reads the page-locked block in place (its device-side address): no H2D copy in front of it.
    const char *src = s->hin_dev;
    if (!src) {
      HIP_TRY(s, hipMemcpyAsync(s->din_block, h, s->in_bytes, hipMemcpyHostToDevice, st));
      src = s->din_block;
    }
    auto at = [&](int i) { return reinterpret_cast<const double *>(src + s->in_off[i]); };
    const bool defer = wait && s->hres_block != nullptr;      // flags come back with the results: look before launching the list pass
    const auto t_launch = std::chrono::high_resolution_clock::now();
    rc = enqueue(s, at(0), at(1), at(2), at(3), at(4), at(5), at(6), st, /*qa_f32=*/false, defer);
    if (rc != FCCQP_OK) return rc;
    if (!wait) {
      s->last_solve_wall = 0.0;
      return FCCQP_OK;
    }
    if (defer) {
      // results behind the kernel, one synchronisation for both; GetSolution then copies nothing.  (A failure from
      // here on leaves no end event and perhaps a fallback list without a consumer: abandon_deferred.)
      hipError_t he = hipMemcpyAsync(s->hres_block, s->dres_block, s->res_bytes, hipMemcpyDeviceToHost, st);
      if (he == hipSuccess) he = spin_synchronize(st);
      if (he != hipSuccess) return abandon_deferred(s, st, "result copy of a synchronous solve", he);
      bool flagged = false;
      if (s->fb_deferred) {
        const int *fl = reinterpret_cast<const int *>(s->hres_block + s->res_off[4]);
        for (size_t i = 0; i < B; ++i) flagged = flagged || ((fl[i] & FCCQP_FLAG_PRESOLVE_NUMERICAL) != 0);
      }
      if (flagged) {          // rare: the COD pass, then the results again
        rc = run_deferred_fallback(s, st);
        if (rc != FCCQP_OK) {
          reset_after_failed_solve(s, st);
          return rc;
        }
        he = hipMemcpyAsync(s->hres_block, s->dres_block, s->res_bytes, hipMemcpyDeviceToHost, st);
        if (he == hipSuccess) he = hipStreamSynchronize(st);
        if (he != hipSuccess) return abandon_deferred(s, st, "result copy behind the fallback pass", he);
      }
      s->fb_deferred = false;
      s->hres_fresh = true;
      s->quiesced = true;
      s->host_device_ms = std::chrono::duration<float, std::milli>(std::chrono::high_resolution_clock::now() - t_launch).count();
    } else {
      HIP_TRY(s, hipStreamSynchronize(st));
    }
    const auto t1s = std::chrono::high_resolution_clock::now();
    s->last_solve_wall = std::chrono::duration<double>(t1s - t0).count();  // src/fcc_qp.cpp:188-190
    return FCCQP_OK;
  }
  rc = [&]() -> int {
    // (copy_in: page-locked arrays in place, anything else through the handle's bounce buffers -- the GPU never
    // addresses pageable caller memory)
    int rc2 = copy_in(s, s->dQ, Q, sizeof(double) * B * n * n, st);
    if (rc2 == FCCQP_OK) rc2 = copy_in(s, s->db, b, sizeof(double) * B * n, st);
    if (rc2 == FCCQP_OK && m * n) rc2 = copy_in(s, s->dA, A_eq, sizeof(double) * B * m * n, st);
    if (rc2 == FCCQP_OK && m) rc2 = copy_in(s, s->dbeq, b_eq, sizeof(double) * B * m, st);
    if (rc2 == FCCQP_OK && nm)   // keep the first nc/3 of every instance's row
      rc2 = copy_in_rows(s, s->dmu, friction_coeffs, sizeof(double) * nm, sizeof(double) * (size_t)n_friction, B, st);
    if (rc2 == FCCQP_OK) rc2 = copy_in(s, s->dlb, lb, sizeof(double) * B * n, st);
    if (rc2 == FCCQP_OK) rc2 = copy_in(s, s->dub, ub, sizeof(double) * B * n, st);
    if (rc2 != FCCQP_OK) return rc2;
    return enqueue(s, s->dQ, s->db, s->dA, s->dbeq, s->dmu, s->dlb, s->dub, st);
  }();
  if (rc != FCCQP_OK) {
    // the caller's buffers are borrowed for the copies: none of them may still be running when the error returns
    reset_after_failed_solve(s, st);
    return rc;
  }
  if (!wait) {
    s->last_solve_wall = 0.0;   // reported as the kernel time, like a device-pointer solve
    return FCCQP_OK;
  }
  HIP_TRY(s, hipStreamSynchronize(st));
  const auto t1 = std::chrono::high_resolution_clock::now();
  s->last_solve_wall = std::chrono::duration<double>(t1 - t0).count();  // src/fcc_qp.cpp:188-190
  return FCCQP_OK;
}
}  // namespace

int fccqp_solve_device(fccqp_solver *s, const double *dQ, const double *db, const double *dA_eq,
                       const double *db_eq, const double *dfriction, const double *dlb,
                       const double *dub, void *stream) {
  if (!s) return FCCQP_ERR_INVALID_ARG;
  const size_t n = (size_t)s->n, m = (size_t)s->m, nm = n_mu(s);
  if ((n && (!dQ || !db || !dlb || !dub)) || (m && n && !dA_eq) || (m && !db_eq) || (nm && !dfriction))
    return fail(s, FCCQP_ERR_INVALID_ARG, "null device pointer");
  DeviceGuard g(s->device);
  if (!g.ok) return fail(s, FCCQP_ERR_NO_DEVICE, "hipSetDevice failed");
  hipStream_t st = (stream == FCCQP_STREAM_OWN) ? s->stream : reinterpret_cast<hipStream_t>(stream);
  s->last_solve_wall = 0.0;
  const int rc = order_after_previous(s, st);
  if (rc != FCCQP_OK) return rc;
  return enqueue(s, dQ, db, dA_eq, db_eq, dfriction, dlb, dub, st);
}

int fccqp_solve_async_f32(fccqp_solver *s, const float *Q, const float *b, const float *A_eq,
                          const float *b_eq, const float *friction_coeffs, int n_friction,
                          const float *lb, const float *ub) {
  if (!s) return FCCQP_ERR_INVALID_ARG;
  const size_t B = (size_t)s->batch, n = (size_t)s->n, m = (size_t)s->m, nm = n_mu(s);
  if ((n && (!Q || !b || !lb || !ub)) || (m && n && !A_eq) || (m && !b_eq) || (nm && !friction_coeffs))
    return fail(s, FCCQP_ERR_INVALID_ARG, "null input pointer");
  if (n_friction != (int)nm) return fail(s, FCCQP_ERR_FRICTION_COEFFS, "n_friction must equal nc/3 here");
  DeviceGuard g(s->device);
  if (!g.ok) return fail(s, FCCQP_ERR_NO_DEVICE, "hipSetDevice failed");
  int rc = ensure_staging(s);
  if (rc != FCCQP_OK) return rc;
  rc = ensure_vec_staging(s);
  if (rc != FCCQP_OK) return rc;
  hipStream_t st = s->stream;
  rc = order_after_previous(s, st);
  if (rc != FCCQP_OK) return rc;
  // Everything travels as float.  Q and A_eq land in the front halves of their f64 staging buffers and
  // are read there by the kernel; the five small vectors land in the unused back half of Q's buffer
  // (B n^2 floats: far more than B (3n + m + nc/3)) and are widened on the device.
  float *const back = reinterpret_cast<float *>(s->dQ) + B * n * n;
  float *const dv[5] = {back, back + B * n, back + B * (n + m), back + B * (n + m + nm), back + B * (2 * n + m + nm)};
  const float *const hv[5] = {b, b_eq, friction_coeffs, lb, ub};
  const size_t cnt[5] = {B * n, B * m, B * nm, B * n, B * n};
  if (B * (3 * n + m + nm) > B * n * n) return fail(s, FCCQP_ERR_UNSUPPORTED, "shape too small for the float staging layout");
  rc = [&]() -> int {
    int rc2 = copy_in(s, s->dQ, Q, sizeof(float) * B * n * n, st);
    if (rc2 == FCCQP_OK && m * n) rc2 = copy_in(s, s->dA, A_eq, sizeof(float) * B * m * n, st);
    for (int i = 0; i < 5 && rc2 == FCCQP_OK; ++i)
      if (cnt[i]) rc2 = copy_in(s, dv[i], hv[i], sizeof(float) * cnt[i], st);
    if (rc2 != FCCQP_OK) return rc2;
    const float *const src[5] = {dv[0], dv[1], dv[2], dv[3], dv[4]};
    double *const dst[5] = {s->wb, s->wbeq, s->wmu, s->wlb, s->wub};
    HIP_TRY(s, fccqp::launch_widen5(src, dst, cnt, st));
    s->last_solve_wall = 0.0;   // reported as the kernel time, like a device-pointer solve
    return enqueue(s, s->dQ, s->wb, s->dA, s->wbeq, s->wmu, s->wlb, s->wub, st, /*qa_f32=*/true);
  }();
  if (rc != FCCQP_OK) reset_after_failed_solve(s, st);      // (no copy out of the caller's buffers outlives the error)
  return rc;
}

int fccqp_solve_device_f32(fccqp_solver *s, const float *dQ, const float *db, const float *dA_eq,
                           const float *db_eq, const float *dfriction, const float *dlb,
                           const float *dub, void *stream) {
  if (!s) return FCCQP_ERR_INVALID_ARG;
  const size_t B = (size_t)s->batch, n = (size_t)s->n, m = (size_t)s->m, nm = n_mu(s);
  if ((n && (!dQ || !db || !dlb || !dub)) || (m && n && !dA_eq) || (m && !db_eq) || (nm && !dfriction))
    return fail(s, FCCQP_ERR_INVALID_ARG, "null device pointer");
  DeviceGuard g(s->device);
  if (!g.ok) return fail(s, FCCQP_ERR_NO_DEVICE, "hipSetDevice failed");
  int rc = ensure_vec_staging(s);
  if (rc != FCCQP_OK) return rc;
  hipStream_t st = (stream == FCCQP_STREAM_OWN) ? s->stream : reinterpret_cast<hipStream_t>(stream);
  rc = order_after_previous(s, st);          // the widened vectors are shared staging: before the first write
  if (rc != FCCQP_OK) return rc;
  const float *const src[5] = {db, db_eq, dfriction, dlb, dub};
  double *const dst[5] = {s->wb, s->wbeq, s->wmu, s->wlb, s->wub};
  const size_t cnt[5] = {B * n, B * m, B * nm, B * n, B * n};
  HIP_TRY(s, fccqp::launch_widen5(src, dst, cnt, st));
  s->last_solve_wall = 0.0;
  return enqueue(s, reinterpret_cast<const double *>(dQ), s->wb, reinterpret_cast<const double *>(dA_eq),
                 s->wbeq, s->wmu, s->wlb, s->wub, st, /*qa_f32=*/true);
}

int fccqp_solve_f32(fccqp_solver *s, const float *Q, const float *b, const float *A_eq,
                    const float *b_eq, const float *friction_coeffs, int n_friction,
                    const float *lb, const float *ub) {
  if (!s) return FCCQP_ERR_INVALID_ARG;
  const size_t B = (size_t)s->batch, n = (size_t)s->n, m = (size_t)s->m, nm = n_mu(s);
  if ((n && (!Q || !b || !lb || !ub)) || (m && n && !A_eq) || (m && !b_eq) || (nm && !friction_coeffs))
    return fail(s, FCCQP_ERR_INVALID_ARG, "null input pointer");
  if (n_friction < (int)nm) return fail(s, FCCQP_ERR_FRICTION_COEFFS, "friction_coeffs too short");
  const auto t0 = std::chrono::high_resolution_clock::now();
  DeviceGuard g(s->device);
  if (!g.ok) return fail(s, FCCQP_ERR_NO_DEVICE, "hipSetDevice failed");
  int rc = ensure_staging(s);
  if (rc != FCCQP_OK) return rc;
  hipStream_t st = s->stream;
  rc = order_after_previous(s, st);
  if (rc != FCCQP_OK) return rc;
  // Q and A_eq travel and stay as float (in the front half of the f64 staging buffers); the small
  // vectors are widened here on the host
  std::vector<double> wb(b, b + B * n), wbeq(b_eq, b_eq + (m ? B * m : 0)), wlb(lb, lb + B * n),
      wub(ub, ub + B * n), wmu(B * nm);
  for (size_t i = 0; i < B; ++i)
    for (size_t c = 0; c < nm; ++c) wmu[i * nm + c] = (double)friction_coeffs[i * (size_t)n_friction + c];
  rc = [&]() -> int {
    int rc2 = copy_in(s, s->dQ, Q, sizeof(float) * B * n * n, st);
    if (rc2 == FCCQP_OK && m * n) rc2 = copy_in(s, s->dA, A_eq, sizeof(float) * B * m * n, st);
    if (rc2 == FCCQP_OK) rc2 = copy_in(s, s->db, wb.data(), sizeof(double) * B * n, st);
    if (rc2 == FCCQP_OK && m) rc2 = copy_in(s, s->dbeq, wbeq.data(), sizeof(double) * B * m, st);
    if (rc2 == FCCQP_OK && nm) rc2 = copy_in(s, s->dmu, wmu.data(), sizeof(double) * B * nm, st);
    if (rc2 == FCCQP_OK) rc2 = copy_in(s, s->dlb, wlb.data(), sizeof(double) * B * n, st);
    if (rc2 == FCCQP_OK) rc2 = copy_in(s, s->dub, wub.data(), sizeof(double) * B * n, st);
    if (rc2 != FCCQP_OK) return rc2;
    return enqueue(s, s->dQ, s->db, s->dA, s->dbeq, s->dmu, s->dlb, s->dub, st, /*qa_f32=*/true);
  }();
  if (rc != FCCQP_OK) {
    reset_after_failed_solve(s, st);      // (drains the stream: the widened vectors above outlive their copies)
    return rc;
  }
  HIP_TRY(s, hipStreamSynchronize(st));   // also keeps the host vectors alive until the copies ran
  const auto t1 = std::chrono::high_resolution_clock::now();
  s->last_solve_wall = std::chrono::duration<double>(t1 - t0).count();
  return FCCQP_OK;
}

int fccqp_synchronize(fccqp_solver *s) {
  if (!s) return FCCQP_ERR_INVALID_ARG;
  DeviceGuard g(s->device);
  HIP_TRY(s, wait_last(s));
  if (s->copy_pending) {
    HIP_TRY(s, hipEventSynchronize(s->ev2));
    s->copy_pending = false;
  }
  return FCCQP_OK;
}

int fccqp_get_solution(fccqp_solver *s, double *z, fccqp_details *details) {
  if (!s) return FCCQP_ERR_INVALID_ARG;
  if (!s->solved) return fail(s, FCCQP_ERR_NOT_SOLVED, "GetSolution before Solve");
  DeviceGuard g(s->device);
  HIP_TRY(s, wait_last(s));
  const size_t B = (size_t)s->batch, n = (size_t)s->n;
  const bool one_copy = s->hres_block != nullptr;
  if (one_copy) {   // small batch: the whole result block in one copy into page-locked memory
    if (!s->hres_fresh) HIP_TRY(s, hipMemcpy(s->hres_block, s->dres_block, s->res_bytes, hipMemcpyDeviceToHost));
    if (z && n) std::memcpy(z, s->hres_block + s->res_off[0], sizeof(double) * B * n);
  } else if (z && n) {
    const int rc = copy_out(s, z, s->dx, sizeof(double) * B * n, s->stream);
    if (rc != FCCQP_OK) return rc;
  }
  if (details) {
    std::vector<int> it(B), fl(B);
    std::vector<double> met(4 * B);
    std::vector<long long> tk(B);
    if (one_copy) {
      std::memcpy(met.data(), s->hres_block + s->res_off[1], sizeof(double) * 4 * B);
      std::memcpy(tk.data(), s->hres_block + s->res_off[2], sizeof(long long) * B);
      std::memcpy(it.data(), s->hres_block + s->res_off[3], sizeof(int) * B);
      std::memcpy(fl.data(), s->hres_block + s->res_off[4], sizeof(int) * B);
    } else {
      int rc = copy_out(s, it.data(), s->dniter, sizeof(int) * B, s->stream);
      if (rc == FCCQP_OK) rc = copy_out(s, fl.data(), s->dflags, sizeof(int) * B, s->stream);
      if (rc == FCCQP_OK) rc = copy_out(s, met.data(), s->dmetrics, sizeof(double) * 4 * B, s->stream);
      if (rc == FCCQP_OK) rc = copy_out(s, tk.data(), s->dticks, sizeof(long long) * B, s->stream);
      if (rc != FCCQP_OK) return rc;
    }
    double wall = s->last_solve_wall;
    if (wall == 0.0) {  // device-pointer solve: report the kernel time instead
      float ms = 0.f;
      if (s->timed && hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) wall = 1e-3 * (double)ms;
    }
    for (size_t i = 0; i < B; ++i) {
      fccqp_details &d = details[i];
      d.n_iter = it[i];
      d.admm_residual_bounds = met[0 * B + i];
      d.admm_residual_friction_cone = met[1 * B + i];
      d.bounds_viol = met[2 * B + i];
      d.friction_cone_viol = met[3 * B + i];
      d.solve_time = wall;
      d.factorization_time = 1e-8 * (double)tk[i];  // wall_clock64() ticks at 100 MHz
      // src/fcc_qp.cpp:203-204
      d.solve_status = (it[i] == s->opt.max_iter) ? FCCQP_STATUS_MAX_ITERATIONS : FCCQP_STATUS_SUCCESS;
      d.flags = fl[i];
    }
  }
  return FCCQP_OK;
}

int fccqp_get_solution_device(fccqp_solver *s, const double **z, const int **n_iter,
                              const int **flags, const double **metrics,
                              const long long **factor_ticks) {
  if (!s) return FCCQP_ERR_INVALID_ARG;
  if (!s->solved) return fail(s, FCCQP_ERR_NOT_SOLVED, "GetSolution before Solve");
  if (z) *z = s->dx;
  if (n_iter) *n_iter = s->dniter;
  if (flags) *flags = s->dflags;
  if (metrics) *metrics = s->dmetrics;
  if (factor_ticks) *factor_ticks = s->dticks;
  return FCCQP_OK;
}

int fccqp_get_state(fccqp_solver *s, double *x, double *mu_x, double *mu_lambda_c) {
  if (!s) return FCCQP_ERR_INVALID_ARG;
  DeviceGuard g(s->device);
  HIP_TRY(s, wait_last(s));
  const size_t B = (size_t)s->batch, n = (size_t)s->n, nc = (size_t)s->nc;
  int rc = FCCQP_OK;
  if (x && n) rc = copy_out(s, x, s->dx, sizeof(double) * B * n, s->stream);
  if (rc == FCCQP_OK && mu_x && n) rc = copy_out(s, mu_x, s->dmux, sizeof(double) * B * n, s->stream);
  if (rc == FCCQP_OK && mu_lambda_c && nc) rc = copy_out(s, mu_lambda_c, s->dmulc, sizeof(double) * B * nc, s->stream);
  return rc;
}

int fccqp_set_state(fccqp_solver *s, const double *x, const double *mu_x, const double *mu_lambda_c) {
  if (!s) return FCCQP_ERR_INVALID_ARG;
  DeviceGuard g(s->device);
  HIP_TRY(s, wait_last(s));
  s->hres_fresh = false;               // x doubles as z: the page-locked copy of the results is stale now
  const size_t B = (size_t)s->batch, n = (size_t)s->n, nc = (size_t)s->nc;
  int rc = FCCQP_OK;
  if (x && n) rc = copy_in(s, s->dx, x, sizeof(double) * B * n, s->stream);
  if (rc == FCCQP_OK && mu_x && n) rc = copy_in(s, s->dmux, mu_x, sizeof(double) * B * n, s->stream);
  if (rc == FCCQP_OK && mu_lambda_c && nc) rc = copy_in(s, s->dmulc, mu_lambda_c, sizeof(double) * B * nc, s->stream);
  if (rc != FCCQP_OK) return rc;
  // (the state is in place before the call returns, as with the blocking copies this replaces; a following solve on
  // another stream is ordered behind it by that)
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  return FCCQP_OK;
}

int fccqp_last_kernel_ms(fccqp_solver *s, float *ms) {
  if (!s || !ms) return FCCQP_ERR_INVALID_ARG;
  if (!s->solved) return fail(s, FCCQP_ERR_NOT_SOLVED, "no solve has been run");
  DeviceGuard g(s->device);
  if (s->quiesced) {       // low-latency path: no events were recorded (see fccqp_solver::quiesced)
    *ms = s->host_device_ms;
    return FCCQP_OK;
  }
  if (!s->timed) return fail(s, FCCQP_ERR_UNSUPPORTED, "kernel timing is switched off (fccqp_set_kernel_timing)");
  HIP_TRY(s, hipEventSynchronize(s->ev1));
  HIP_TRY(s, hipEventElapsedTime(ms, s->ev0, s->ev1));
  return FCCQP_OK;
}

int fccqp_last_timing(fccqp_solver *s, float *ms, int *measured_by) {
  if (!s || !ms || !measured_by) return FCCQP_ERR_INVALID_ARG;
  if (!s->solved) return fail(s, FCCQP_ERR_NOT_SOLVED, "no solve has been run");
  DeviceGuard g(s->device);
  *ms = 0.f;
  *measured_by = FCCQP_TIMING_NONE;
  if (s->quiesced) {
    *ms = s->host_device_ms;
    *measured_by = FCCQP_TIMING_HOST_CLOCK;
  } else if (s->timed) {
    HIP_TRY(s, hipEventSynchronize(s->ev1));
    HIP_TRY(s, hipEventElapsedTime(ms, s->ev0, s->ev1));
    *measured_by = FCCQP_TIMING_DEVICE_EVENTS;
  }
  return FCCQP_OK;
}

int fccqp_phase_seconds(fccqp_solver *s, double out[8]) {
  if (!s || !out) return FCCQP_ERR_INVALID_ARG;
  for (int i = 0; i < 8; ++i) out[i] = 0.0;
  DeviceGuard g(s->device);
  const size_t B = (size_t)s->batch;
  if (!s->dphase) {
    HIP_TRY(s, wait_last(s));
    HIP_TRY(s, dmalloc(&s->dphase, 8 * B));
    HIP_TRY(s, hipMemset(s->dphase, 0, sizeof(long long) * 8 * B));
    HIP_TRY(s, hipStreamSynchronize(nullptr));       // (see fccqp_create: the fill runs on the null stream)
    return FCCQP_OK;
  }
  if (!s->solved) return FCCQP_OK;
  HIP_TRY(s, wait_last(s));
  std::vector<long long> h(8 * B);
  { const int rc = copy_out(s, h.data(), s->dphase, sizeof(long long) * 8 * B, s->stream); if (rc != FCCQP_OK) return rc; }
  for (int p = 0; p < 8; ++p) {
    double acc = 0.0;
    for (size_t i = 0; i < B; ++i) acc += (double)h[(size_t)p * B + i];
    out[p] = 1e-8 * acc / (double)B;
  }
  return FCCQP_OK;
}

int fccqp_debug_phase_ticks(fccqp_solver *s, long long *out, long long count) {
  if (!s || !out || count < 0) return FCCQP_ERR_INVALID_ARG;
  DeviceGuard g(s->device);
  const long long have = s->dphase ? 8ll * s->batch : 0;
  if (!s->solved || count > have) return FCCQP_ERR_INVALID_ARG;
  HIP_TRY(s, wait_last(s));
  { const int rc = copy_out(s, out, s->dphase, sizeof(long long) * (size_t)count, s->stream); if (rc != FCCQP_OK) return rc; }
  return FCCQP_OK;
}

int fccqp_debug_block_instance(int batch, int block) {
  if (batch < 1 || block < 0 || block >= batch) return -1;
  return (int)fccqp::permuted_index((unsigned)block, (unsigned)batch);
}

int fccqp_debug_read_workspace(fccqp_solver *s, double *out, long long count, long long *available) {
  if (!s) return FCCQP_ERR_INVALID_ARG;
  DeviceGuard g(s->device);
  HIP_TRY(s, wait_last(s));
  const long long have = s->dwork ? (long long)s->batch * s->work_stride : 0;
  if (available) *available = have;
  const long long take = count < have ? count : have;
  if (out && take > 0) { const int rc = copy_out(s, out, s->dwork, sizeof(double) * (size_t)take, s->stream); if (rc != FCCQP_OK) return rc; }
  return FCCQP_OK;
}

int fccqp_algorithmic_bytes(const fccqp_solver *s, long long *bytes) {
  if (!s || !bytes) return FCCQP_ERR_INVALID_ARG;
  const long long n = s->n, m = s->m, nc = s->nc, B = s->batch;
  const long long in_elems = n * n + m * n + 3 * n + m + nc / 3;   // the inputs ...
  long long io_elems = n;                                          // ... and z
  if (s->warm_start) io_elems += 2 * (2 * n + nc);                 // state read + written
  *bytes = B * ((s->last_was_f32 ? 4 : 8) * in_elems + 8 * io_elems);
  return FCCQP_OK;
}

}  // extern "C"
