// fccqp_device.h -- kernel argument block and launch entry points shared by the C ABI
// implementation (fccqp_capi.cpp) and the gfx950 kernels (*.hip).  Internal header.
#pragma once
#ifndef __HIPCC_RTC__            // (hiprtc, fccqp_jit.cpp: the runtime's device header is built in)
#include <hip/hip_runtime.h>
#endif

namespace fccqp {

// One block of arguments for every kernel variant.  All pointers are device pointers,
// instance-major (see include/fccqp.h).
struct KernelArgs {
  int n;        // num_vars                      (reference fcc_qp.hpp:128)
  int m;        // num_equality_constraints      (:131)
                // (The one-wave shape-specialised kernel takes its sizes from template parameters and never reads n / m /
                // nc / s; its launchers put the number of its workgroups that are resident at once -- 4 x compute units:
                // one_wave_resident_workgroups() -- into THIS field for the staggered start of fused_kernel.  A field of
                // its own would cost that kernel 3 % through its kernarg layout, see the note on Q / A below.)
  int nc;       // contact force variables       (:129)
  int s;        // lambda_c_start                (:130)
  int batch;
  int warm_start;                               // (:133)
  int max_iter;                                 // FCCQPOptions (:30-35)
  double rho, eps_fcone, eps_bound;
  // Q and A_eq address float arrays instead when a launcher is called with qa_f32 = true (the
  // fp32-storage entry points): values are widened on load, all arithmetic is fp64.  The flag is a
  // template parameter of the kernels, not a field here: the fused kernel's register allocation is
  // sensitive to the kernarg layout (+3% run time from one extra int).
  const double *Q, *b, *A, *b_eq, *mu, *lb, *ub;
  // warm-start state, also the outputs: x doubles as z (reference GetSolution: z = x_, cpp:205)
  double *x, *mu_x, *mu_lc;
  int *n_iter;            // [batch]
  int *flags;             // [batch]  FCCQP_FLAG_*
  double *metrics;        // [4][batch] residual_bounds, residual_cone, bounds_viol, cone_viol
  long long *factor_ticks;  // [batch] 100 MHz ticks spent factorising
  // rank-deficient presolve fallback (fccqp_cod.hip): instances whose presolve LDL^T met a zero /
  // non-finite pivot are appended to fb_list (count in *fb_count) and left for the fallback pass
  int *fb_list;            // [batch] or null (then such instances just carry the flag)
  int *fb_count;
  // list mode (generic kernel only, the fallback launch): for every instance in
  // inst_list[0..*inst_count) compute the COD x0 in cod_ws, then run ADMM from it; *next_count is
  // zeroed for the following solve (the two counters are used alternately: no memset)
  // (one-wave kernel, batch launch: inst_list != nullptr with inst_count == nullptr is the block -> instance map of this
  // launch -- block b solves instance inst_list[b] -- instead of the fixed permutation of block_instance(); the launcher
  // then takes fused_kernel<.., ORDERED = true>)
  const int *inst_list;
  const int *inst_count;
  int *next_count;
  double *cod_ws;
  long long cod_stride;
  long long *phase_ticks;  // optional [8][batch]: 100 MHz ticks per phase (fused kernel), or null
  double *workspace;      // generic kernel, large shapes: per-block packed KKT in global memory
  long long workspace_stride;  // doubles per block
#ifdef FCCQP_TICKET_DISPATCH
  // experiment (see block_instance below): every workgroup of a batch launch draws a ticket from *ticket, a
  // counter that only ever grows; the launch's first ticket is ticket_base (the host adds the grid size after
  // every launch, so nothing is zeroed between launches)
  unsigned *ticket;
  unsigned ticket_base;
#endif
};

// Which instance a workgroup of a batch launch solves: a fixed pseudo-random permutation of blockIdx.x.
// Workgroups are dealt round-robin over the eight XCDs (block b runs on XCD b % 8; every XCD takes its grid / 8
// blocks in order) and, inside an XCD, onto its CUs in a fixed order too, so with instance = blockIdx.x a
// workload whose cost is periodic in the instance index -- W-mix of fcc_qp_amd/problems.py: even instances leave
// after one ADMM iteration, odd ones run all 100 -- puts the expensive instances on the same XCDs / CUs round after
// round.  The permutation is a bijection of [0, batch) (an invertible integer hash on ceil(log2 batch) bits,
// re-applied while it lands past the batch: cycle walking); results do not depend on it.  Returns -1 past the batch.
//
// Measured on MI355X (tools/dispatch_order.py, profiles/r03*_dispatch_order.json; walking shape, batch 4096, kernel
// ms): W-mix in index order 0.407 with instance = blockIdx.x; 0.389 with the batch shuffled on the host; mixing
// only the low three bits of the index into the XCD choice (b ^ ((b >> 3) & 7)): 0.404 -- the imbalance just moves
// from the XCDs to the CUs of an XCD.  None of them reaches the 0.335 of a perfectly packed schedule: with four
// rounds of 57 / 95 us jobs a greedy schedule ends about (longest job - mean remaining job) after the work runs
// out, whatever the order.  A ticketed DYNAMIC map (-DFCCQP_TICKET_DISPATCH: one returning atomicAdd per workgroup,
// instance = order of arrival) was built and measured too: 0.394-0.397 on W-mix, no better than a static shuffle
// (each XCD still runs exactly grid / 8 workgroups), and 1.2 % slower on W-sat (the atomic's round trip sits in
// front of the first load).  Off by default.
__host__ __device__ __forceinline__ unsigned mix_bits(unsigned x, const unsigned mask, const int half) {
  x ^= x >> half; x = (x * 0x9E3779B1u) & mask;
  x ^= x >> half; x = (x * 0x85EBCA77u) & mask;
  x ^= x >> half;
  return x;
}
// the permutation itself (also exported to the host tests as fccqp_debug_block_instance): t < batch
__host__ __device__ __forceinline__ unsigned permuted_index(unsigned t, const unsigned batch) {
  if (batch >= 128u) {                                     // (smaller launches are resident all at once)
    const int bits = 32 - __builtin_clz(batch - 1u);       // ceil(log2 batch), >= 7
    const unsigned mask = (1u << bits) - 1u;
    const int half = (bits + 1) >> 1;
    // cycle walking, expected < 2 rounds.  The first round stands outside the loop: the kernels are built with every
    // loop header aligned to 256 bytes, and the padding in front of a loop on the way into the kernel is ~50 s_nop
    // that a lone wave pays 6 cycles each for
    t = mix_bits(t, mask, half);
    if (__builtin_expect(t >= batch, 0)) {
      do t = mix_bits(t, mask, half); while (t >= batch);
    }
  }
  return t;
}
__device__ __forceinline__ int block_instance(const KernelArgs &a) {
  unsigned t = blockIdx.x;
#ifdef FCCQP_TICKET_DISPATCH
  if (a.ticket) {
    if (threadIdx.x == 0) t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.ticket_base;
    t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
  }
#endif
  const unsigned batch = (unsigned)a.batch;
  if (t >= batch) return -1;
  return (int)permuted_index(t, batch);
}

// ---- memory fences with their waits WRITTEN OUT ----------------------------------------------------------------
// Lanes (and waves) hand data to each other through global memory in the generic kernel's large shapes (the packed
// KKT matrix in a per-block workspace: the owner of an entry changes from phase to phase) and through a staging
// buffer.  Between phases the writing wave must have its stores COMPLETE, then drop its L1 lines:
//     s_waitcnt vmcnt(0) | buffer_wbl2 sc1 | s_waitcnt vmcnt(0) | buffer_inv sc1 | s_waitcnt vmcnt(0)
// __builtin_amdgcn_fence(seq_cst, "agent") alone is NOT reliable for that with ROCm 7.2's LLVM: its wait-count pass can
// lose track of stores issued before an EXEC-skipped region (s_cbranch_execz over the code that holds the
// `s_waitcnt vmcnt(0)` it placed) and then emits `buffer_wbl2 sc1; s_waitcnt lgkmcnt(0); buffer_inv sc1` -- no vmcnt
// wait on that path (seen in the disassembly of generic_kernel<8, false>: flat_store ... s_cbranch_execz ... fence).
// The stores are then still in flight when the L1 is invalidated, and the next phase's loads are served the PREVIOUS
// phase's values from L2.  This is the cause of the "sporadically wrong iterates" that made round 2 withdraw N > 256:
// ADMM sweeps that read the presolve's factor (rho = 0) instead of the K_rho factor just written over it -- results
// off by O(rho), different from run to run, correct whenever no presolve phase ran before (tools/n512_bisect*.py,
// DESIGN.md section 7).  Inline asm is opaque to the pass, so the waits below always survive
// (MI355X_MICROARCH.md "Compiler hazard" describes the same trap for cross-CU hand-offs).
// (-DFCCQP_NO_EXPLICIT_WAITS builds the compiler's own lowering again: tools/n512_variants.sh, for the record)
#ifdef FCCQP_NO_EXPLICIT_WAITS
#define FCCQP_EXPLICIT_WAIT(what)
#else
#define FCCQP_EXPLICIT_WAIT(what) asm volatile(what ::: "memory")
#endif
__device__ __forceinline__ void agent_release() {
  FCCQP_EXPLICIT_WAIT("s_waitcnt vmcnt(0) lgkmcnt(0)");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  FCCQP_EXPLICIT_WAIT("s_waitcnt vmcnt(0)");
}
__device__ __forceinline__ void agent_acquire() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  FCCQP_EXPLICIT_WAIT("s_waitcnt vmcnt(0)");      // (the invalidate completes asynchronously)
}
__device__ __forceinline__ void agent_release_acquire() {
  agent_release();
  agent_acquire();
}
// workgroup barrier whose LDS and global traffic is complete on arrival, whatever the pass thinks is pending
__device__ __forceinline__ void workgroup_sync() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
}

// flag bits (mirror include/fccqp.h)
constexpr int kFlagPresolveRan = 1;
constexpr int kFlagPresolveNumerical = 2;
constexpr int kFlagAdmmNumerical = 4;
constexpr int kFlagEqualityOnly = 8;
constexpr int kFlagPresolveCod = 16;
constexpr int kFlagEqResidual = 32;
constexpr int kFlagPresolveCg = 64;

#ifndef __HIPCC_RTC__            // host-side launch entry points: not part of a run-time compiled kernel
// ---- generic kernel (fccqp_generic.hip) -----------------------------------------------------
// Largest KKT size N = n + m the generic kernel handles (8 register slots x 64 lanes; the reference's
// shapes are N = 98 and, in BASELINE configs[3], N = 174).  Round 2 had withdrawn N > 256 after sporadically
// wrong iterates whose cause it could not establish; the cause is the fence lowering described at
// agent_release() above (it was latent in the 4-slot global-workspace path, 198 < N <= 256, too), and the range is
// back.  fccqp_create returns FCCQP_ERR_UNSUPPORTED beyond this size.
constexpr int kGenericMaxN = 512;
// Bytes of dynamic LDS the generic kernel needs when the packed KKT lives in LDS.
size_t generic_lds_bytes(int n, int m, int nc);
// Doubles of global workspace per block when the packed KKT does not fit in LDS (else 0).
long long generic_workspace_doubles(int n, int m, int nc, size_t lds_limit);
hipError_t launch_generic(const KernelArgs &a, size_t lds_limit, hipStream_t stream, bool qa_f32);
// list mode: a.inst_list / a.inst_count set; `blocks` persistent one-wave workgroups
hipError_t launch_generic_list(const KernelArgs &a, size_t lds_limit, int blocks, hipStream_t stream,
                               bool qa_f32);

// widens five float arrays to double in one launch (the small per-instance vectors of the
// fp32-storage entry points)
hipError_t launch_widen5(const float *const src[5], double *const dst[5], const size_t count[5],
                         hipStream_t stream);

// ---- self-check behind a solve (fccqp_check.hip): raises kFlagEqResidual on every instance whose z is not finite or
// violates A_eq z = b_eq by more than `tol` relative to the terms of the row (a.A / a.b_eq: the solve's inputs, a.x: z)
hipError_t launch_eq_residual_check(const KernelArgs &a, hipStream_t stream, bool qa_f32, double tol);
// order[0 .. batch) = the instances sorted by n_iter (of the previous solve), longest first: the block -> instance map
// of the one-wave kernel for the next solve of a chain (KernelArgs::inst_list with inst_count == nullptr)
hipError_t launch_order_by_iterations(const int *n_iter, int batch, int *order, hipStream_t stream);
// shapes whose solves are always checked (the global-workspace instantiations of the generic kernel beyond this size)
constexpr int kSelfCheckAboveN = 256;

// ---- presolve fallback workspace (fccqp_cod_impl.h) ---------------------------------------------
long long cod_workspace_doubles_host(int n, int m);

// ---- fused shape-specialised kernels (fccqp_fused.hip) ---------------------------------------
// One-wave workgroups resident at once on the current device (four per compute unit: one per SIMD); cached per device.
inline int one_wave_resident_workgroups() {
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1024;
  if (cached[dev] == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
      (void)hipGetLastError();
      cus = 256;
    }
    cached[dev] = 4 * cus;
  }
  return cached[dev];
}
bool fused_supported(int n, int m, int nc, int s);
bool fused_pointers_ok(const KernelArgs &a);  // (always true since the tile loader: natural alignment suffices)
// split == 0: the whole solve in one launch.  split == 2 or 3: the two-launch form -- fused_kernel<.., SPLIT> up to
// the presolve, then admm_kernel at `split` waves per SIMD; a.workspace must then hold
// fused_split_workspace_doubles(n) doubles per instance (a.workspace_stride)
hipError_t launch_fused(const KernelArgs &a, hipStream_t stream, bool qa_f32, int split);
long long fused_split_workspace_doubles(int n);
hipError_t launch_fused_split(const KernelArgs &a, hipStream_t stream, bool qa_f32, int waves_per_simd);

// ---- cooperative multi-wave kernel (fccqp_coop.hip) ----------------------------------------------
bool coop_supported(int n, int m, int nc, int s);
int coop_resident_workgroups(int n, int m, int nc, int s, int compute_units);
hipError_t launch_coop(const KernelArgs &a, hipStream_t stream, bool qa_f32);
// two-launch form (shapes with num_vars > 64: the row-split ADMM layout): everything up to the presolve at one QP per
// CU, then the ADMM iterations + epilogue at wgs_per_cu (2 or 3) QPs per CU; a.workspace must hold
// coop_split_workspace_doubles() doubles per instance (0: the shape has no two-launch form)
long long coop_split_workspace_doubles(int n, int m, int nc, int s);
hipError_t launch_coop_split(const KernelArgs &a, hipStream_t stream, bool qa_f32, int wgs_per_cu);

#endif  // !__HIPCC_RTC__

}  // namespace fccqp
