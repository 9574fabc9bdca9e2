/*
 * fccqp.h -- C ABI of the MI355X (gfx950) batched friction-cone QP solver.
 *
 * This is the drop-in boundary for the hot path of Brian-Acosta/fcc_qp:
 *     FCCQP::Solve  -> FCCQP::DoADMM          (reference src/fcc_qp.cpp:57-191)
 *     projections / violation metrics          (reference src/constraint_utils.cpp:5-65)
 * The reference has no C ABI of its own (its boundary is the C++ class fcc_qp::FCCQP,
 * src/fcc_qp.hpp:54-172, exported to Python by src/main.cpp:19-56).  The functions below are
 * what that class binds to when its Eigen body is replaced by the HIP path: the C++ class
 * fcc_qp::FCCQP / fcc_qp::FCCQPBatch in fcc_qp_amd/csrc/fcc_qp.hpp is implemented purely on top
 * of this header, and INTEGRATION.md shows the same binding for the reference's own sources.
 *
 * Data layout ("instance-major"): the batch index is the slowest dimension and every instance
 * is one contiguous row-major (C order) block --
 *     Q[batch][n][n]   b[batch][n]   A_eq[batch][m][n]   b_eq[batch][m]
 *     friction_coeffs[batch][nc/3]   lb[batch][n]   ub[batch][n]   z[batch][n]
 * so the 64 lanes of the wavefront that owns an instance read consecutive 8-byte words.
 * Only the lower triangle of Q is read (Q[i][j], i >= j), as Eigen's LDLT does in the
 * reference (src/fcc_qp.cpp:68,161 with the default UpLo = Lower).
 *
 * All arithmetic is IEEE fp64.  There is no CPU fallback: every entry point that needs the GPU
 * returns FCCQP_ERR_NO_DEVICE when none is present.
 *
 * Threading: a solver handle is stateful (warm-start state, workspace) and not re-entrant, like
 * the reference class; distinct handles are independent (one per GPU / per thread).
 */
#ifndef FCCQP_H
#define FCCQP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FCCQP_VERSION_MAJOR 0
#define FCCQP_VERSION_MINOR 1

/* ---- status codes (the reference reports nothing: void returns + asserts compiled out in
 * Release, src/fcc_qp.cpp:29-33,122-130; shape errors here are a documented superset) ------- */
enum {
  FCCQP_OK = 0,
  FCCQP_ERR_INVALID_ARG = 1,     /* NULL pointer, bad enum value, bad handle                    */
  FCCQP_ERR_SHAPE = 2,           /* sizes violate fcc_qp.cpp:29-33 (nc%3, lambda_c_start+nc<=n) */
  FCCQP_ERR_NO_DEVICE = 3,       /* no HIP device / device index out of range                   */
  FCCQP_ERR_HIP = 4,             /* a HIP runtime call failed; see fccqp_last_error()           */
  FCCQP_ERR_ALLOC = 5,
  FCCQP_ERR_FRICTION_COEFFS = 6, /* fewer than nc/3 coefficients: the reference throws
                                    std::out_of_range from .at() (constraint_utils.cpp:32,55)   */
  FCCQP_ERR_NOT_SOLVED = 7,      /* get_solution before any solve, or after a solve that failed */
  FCCQP_ERR_UNSUPPORTED = 8      /* shape too large for the device workspace, or no kernel of
                                    the requested kind (fccqp_set_kernel) for this shape         */
};
/* A solve entry point that returns an error leaves the handle as before its first solve: nothing of that solve is
 * still running or copying when the call returns (the caller's arrays are free), fccqp_get_solution answers
 * FCCQP_ERR_NOT_SOLVED rather than the previous solve's z, and the handle can solve again.  The warm-start state is
 * whatever the kernels that did run left: start cold after an error. */

/* reference: enum FCCQPSolveStatus, src/fcc_qp.hpp:14-17 */
enum { FCCQP_STATUS_SUCCESS = 0, FCCQP_STATUS_MAX_ITERATIONS = 1 };

/* per-instance flag bits (diagnostics beyond the reference struct) */
enum {
  FCCQP_FLAG_PRESOLVE_RAN = 1,          /* presolve branch of fcc_qp.cpp:159 executed            */
  FCCQP_FLAG_PRESOLVE_NUMERICAL = 2,    /* presolve LDL^T met a zero / non-finite pivot: where
                                           the reference switches to its COD fallback
                                           (fcc_qp.cpp:164-166).  Without FLAG_PRESOLVE_COD
                                           (fallback disabled) z is NOT comparable              */
  FCCQP_FLAG_ADMM_NUMERICAL = 4,        /* ADMM LDL^T met a zero / non-finite pivot              */
  FCCQP_FLAG_EQUALITY_ONLY = 8,         /* nc == 0 and all bounds infinite: no ADMM (:132-133)   */
  FCCQP_FLAG_PRESOLVE_COD = 16,         /* x0 came from the device COD fallback (min-norm least
                                           squares, fcc_qp.cpp:165,176): z IS comparable again   */
  FCCQP_FLAG_PRESOLVE_CG = 64,          /* the presolve's system (I - rho G) x0 = x_c was solved by conjugate gradients
                                           (one-wave kernel, cold start; same x0 to ~1e-13).  Clear with PRESOLVE_RAN
                                           set: the direct factorisation ran (other kernels; or CG declined: singular
                                           diagonal, vanishing curvature, no convergence in n - m + 10 steps)           */
  FCCQP_FLAG_EQ_RESIDUAL = 32           /* self-check (fccqp_set_self_check): the returned z is not finite, or
                                           |A_eq z - b_eq| exceeds 1e-6 of the terms of some row.  Every primal
                                           iterate is the head of a KKT solve (fcc_qp.cpp:86-87), so this never
                                           happens for consistent equality rows unless the solve went wrong      */
};

/* kernel selection */
enum {
  FCCQP_KERNEL_AUTO = 0,     /* specialised kernel when the shape has one, else generic          */
  FCCQP_KERNEL_GENERIC = 1,  /* any shape: LDS/global-resident KKT, triangular solves per iter   */
  FCCQP_KERNEL_FUSED = 2,    /* shape-specialised (fused_shapes.def), one wavefront per QP, one launch:
                                K_rho = U^T D U by 16 x 16 register tiles on the fp64 matrix cores with
                                the forward solve fused in, rho G as a Gram product, presolve from
                                W = I - rho G, mat-vec ADMM iterations with DPP-fed FMAs               */
  FCCQP_KERNEL_FUSED_SPLIT = 3, /* the same in two launches: everything up to the presolve at one wave
                                per SIMD, then the ADMM iterations + epilogue at two waves per SIMD
                                (rho G takes a round trip through a device workspace of 8 * 64 * (n + 1)
                                bytes per instance).  Bit-identical results; measured 3-4 % faster (round 3; 7 % slower since round 4) at
                                batch 4096 with 100 iterations, 8 % slower with the paper's 15: AUTO
                                never picks it                                  */
  FCCQP_KERNEL_COOP = 4,     /* shape-specialised (coop_shapes.def): one workgroup of 2-8 wavefronts per QP,
                                the register tiles shared out over the waves.  Shapes with num_vars >= 64
                                (BASELINE configs[3]: 3x the generic kernel).  AUTO: a listed shape that
                                has no one-wave kernel (for the walking shape four waves per QP are no
                                faster than one, even at batch 1)                                     */
  FCCQP_KERNEL_COOP_SPLIT = 5 /* the cooperative kernel in two launches (listed shapes with num_vars > 64):
                                everything up to the presolve at one QP per compute unit, then the ADMM
                                iterations + epilogue at two or three QPs per compute unit, whose waits
                                (the exchange of w, the barrier) they fill for each other; rho G takes a
                                round trip through a device workspace of ~1 KB per variable per
                                instance (allocated on first use).  Bit-identical to FCCQP_KERNEL_COOP.
                                AUTO takes it for such a shape when the batch is several times the
                                number of compute units, FCCQP_KERNEL_COOP otherwise (one launch: the
                                shorter way for a few QPs)                                            */
};

/* reference: struct FCCQPOptions, src/fcc_qp.hpp:30-35 (same defaults) */
typedef struct fccqp_options {
  int max_iter;      /* 1000 */
  double rho;        /* 1e-6 */
  double eps_fcone;  /* 1e-3 */
  double eps_bound;  /* 1e-6 */
} fccqp_options;

/* reference: struct FCCQPDetails, src/fcc_qp.hpp:19-28, one per instance */
typedef struct fccqp_details {
  int n_iter;
  double admm_residual_bounds;         /* python: eps_bounds        (main.cpp:24) */
  double admm_residual_friction_cone;  /* python: eps_friction_cone (main.cpp:25) */
  double solve_time;          /* seconds: host wall clock of the solve call that produced it     */
  double factorization_time;  /* seconds: device time this instance spent in its factorisations  */
  double bounds_viol;
  double friction_cone_viol;
  int solve_status;           /* FCCQP_STATUS_*  (n_iter == max_iter ? MAX_ITERATIONS : SUCCESS) */
  int flags;                  /* FCCQP_FLAG_* */
} fccqp_details;

typedef struct fccqp_solver fccqp_solver;

/* ---- library ------------------------------------------------------------------------------ */
int fccqp_version(void);                 /* major*100 + minor */
const char *fccqp_strerror(int code);
int fccqp_device_count(void);            /* number of HIP devices visible (0 if none)           */

/* ---- life cycle   (replaces FCCQP::FCCQP, reference src/fcc_qp.cpp:24-55) ------------------ */
/* `batch` independent instances of one shape, resident on HIP device `device`.
 * The reference constructor is batch = 1.
 * Size limit (the reference has none: it allocates any N x N, src/fcc_qp.cpp:38-41): num_vars +
 * num_equality_constraints <= 512, FCCQP_ERR_UNSUPPORTED beyond.  512 is eight KKT rows per lane of the any-shape
 * kernel; the reference's own shapes are N = 98 (its harness) and N = 174 (36 DoF, eight contacts).  A caller with a
 * larger KKT system has two ways that keep this library: (i) eliminate the equality constraints on the host when
 * they are what makes it large: pick num_equality_constraints variables x_d that carry neither a finite bound nor a
 * contact force and whose columns A_d of A_eq are invertible (the accelerations of an operational-space QP, which the
 * dynamics rows determine, are such a set), substitute x_d = A_d^-1 (b_eq - A_f x_f), and solve for the remaining
 * variables x_f -- the bounded ones and lambda_c among them, still coordinates, so box and cone apply unchanged --
 * with Q_f = [-A_d^-1 A_f; I]^T Q [-A_d^-1 A_f; I], the matching b_f and num_equality_constraints = 0 (served by the
 * any-shape kernel); N drops by 2 num_equality_constraints, the minimiser is the same, the ADMM iterates on the way
 * to it are not; (ii) split a block-diagonal problem -- several robots, several
 * time steps without coupling -- into its blocks, which is what `batch` is for.  What this library does not do is
 * factorise one dense KKT matrix beyond 512 rows: at that size the matrix (2 MB and more) no longer lives near the
 * lanes that eliminate it, and a blocked out-of-core LDL^T is a different kernel, not a larger instance of this one. */
int fccqp_create(int num_vars, int num_equality_constraints, int nc, int lambda_c_start,
                 int batch, int device, fccqp_solver **out);
int fccqp_destroy(fccqp_solver *s);
const char *fccqp_last_error(const fccqp_solver *s);

/* ---- options      (replaces set_rho/set_max_iter/set_options/set_warm_start,
 *                    reference src/fcc_qp.hpp:75-91) ---------------------------------------- */
int fccqp_default_options(fccqp_options *opt);
int fccqp_set_options(fccqp_solver *s, const fccqp_options *opt);
int fccqp_get_options(const fccqp_solver *s, fccqp_options *opt);
int fccqp_set_warm_start(fccqp_solver *s, int warm_start);
int fccqp_set_kernel(fccqp_solver *s, int kernel);   /* FCCQP_KERNEL_* */
/* Which instance the k-th workgroup of a batch launch solves (one-wave kernel of a shape compiled into the library,
 * batches of at least eight instances per compute unit; no effect elsewhere, and never an effect on results).  The workgroups of a launch take the SIMDs in
 * order as slots fall free, and an instance is one indivisible job whose length goes with its iteration count:
 *   FCCQP_ORDER_HASH                 a fixed pseudo-random permutation (default): no structure of the batch -- every
 *                                    other instance easy, say -- lines up with the hardware's dealing order;
 *   FCCQP_ORDER_PREVIOUS_ITERATIONS  longest first, by the iteration counts of the PREVIOUS solve on this handle: for a
 *                                    chain of solves (the same robots a tick later, fcc_qp_test.py:77-89) the last
 *                                    count is a good guess at the next, and a launch then ends when its work runs out
 *                                    instead of a job's length later.  One ~5 us sorting launch per solve.
 *                                    Measured (profiles/r05_dispatch_order_ab.json): + 6 % on a batch that is half
 *                                    short, half long instances; - 2 .. - 5 % where nearly all instances run the
 *                                    same number of iterations (nothing to pack, the sort is pure cost). */
enum { FCCQP_ORDER_HASH = 0, FCCQP_ORDER_PREVIOUS_ITERATIONS = 1 };
int fccqp_set_dispatch_order(fccqp_solver *s, int order);
/* Rank-deficient presolve fallback on the device (replaces the CompleteOrthogonalDecomposition
 * branch, reference src/fcc_qp.cpp:164-166,176).  On by default; costs two near-empty kernel
 * launches per solve when no instance needs it. */
int fccqp_set_presolve_fallback(fccqp_solver *s, int enabled);
/* The start-of-kernel event of a solve (what fccqp_last_kernel_ms and, for device-pointer solves,
 * fccqp_details.solve_time are measured from).  On by default.  A hipEventRecord in front of the kernel makes the
 * kernel wait for the stream to drain -- ~5 us per solve, 1.5 % of a 4096-instance walking-shape batch -- so a
 * caller that times its solves itself can switch it off: fccqp_last_kernel_ms then returns FCCQP_ERR_UNSUPPORTED
 * and solve_time of a device-pointer solve is 0 (small synchronous host-pointer solves are timed on the host clock
 * whatever this switch says, see fccqp_last_timing).  (No counterpart in the reference, whose solve_time is a host
 * chrono pair, src/fcc_qp.cpp:115,188.) */
int fccqp_set_kernel_timing(fccqp_solver *s, int enabled);
/* Self-check behind every solve: one more small launch that re-computes A_eq z - b_eq from the solve's inputs and
 * raises FCCQP_FLAG_EQ_RESIDUAL on the instances that fail it.  Default: on for num_vars + num_equality_constraints
 * > 256 (the generic kernel's global-workspace instantiations), off otherwise.  The
 * reference has no counterpart: it reports nothing about the quality of a solve beyond n_iter. */
int fccqp_set_self_check(fccqp_solver *s, int enabled);
int fccqp_get_kernel(const fccqp_solver *s, int *kernel_in_use);
/* Which kernel FCCQP_KERNEL_AUTO takes for a shape in THIS build of the library (no handle, no device needed):
 * FCCQP_KERNEL_FUSED or FCCQP_KERNEL_COOP for the constructor arguments listed in fused_shapes.def / coop_shapes.def
 * when the library was built and for every shape that qualifies for run-time compilation (below),
 * FCCQP_KERNEL_GENERIC for every other valid shape (several times slower), a negative status code
 * (-FCCQP_ERR_SHAPE, -FCCQP_ERR_UNSUPPORTED) for arguments fccqp_create would refuse. */
int fccqp_kernel_for_shape(int num_vars, int num_equality_constraints, int nc, int lambda_c_start);

/* ---- run-time shape specialisation -------------------------------------------------------------------------------
 * The reference's constructor takes its sizes at run time (FCCQP::FCCQP, src/fcc_qp.cpp:24-55): nothing is a build
 * option there.  Here the fast kernels are templates on the sizes; for a shape that is not compiled in, fccqp_create
 * instantiates the template IN PROCESS with hiprtc (the kernel headers travel inside the library as text; no child
 * process), loads the code object and keeps it in a disk cache ($FCCQP_JIT_CACHE_DIR, else $XDG_CACHE_HOME/fccqp_jit,
 * else ~/.cache/fccqp_jit, else /tmp/fccqp_jit_<uid>) keyed by shape, storage type, library version, compile flags, a
 * hash of the kernel sources and the hiprtc version: the first create of a new shape costs a few seconds, every later
 * one (any process) a file read.  One-wave kernel for num_vars <= 63, num_equality_constraints <= 64, <= 16 contacts;
 * else the cooperative kernel for num_vars <= ~118 (<= 8 contacts above 64 variables); nc > 0 in both.  If the shape
 * does not qualify or compilation is not possible (no libhiprtc, a requirement of the kernel's Shape not met), the
 * handle uses the generic kernel and fccqp_last_error(handle) says why -- fccqp_create still returns FCCQP_OK.
 * FCCQP_JIT=0 in the environment, or fccqp_set_jit(0), switches the mechanism off (process-wide). */
enum {
  FCCQP_ORIGIN_BUILTIN = 0,   /* compiled into the library (fused_shapes.def / coop_shapes.def), or the generic kernel */
  FCCQP_ORIGIN_JIT = 1,       /* instantiated at run time for this shape */
  FCCQP_ORIGIN_GENERIC_ONLY = 2  /* no specialised kernel can be had for the shape: generic kernel */
};
/* Where the specialised kernel of a shape comes from in this build (no handle, no device, nothing compiled):
 * FCCQP_ORIGIN_BUILTIN for a listed shape, FCCQP_ORIGIN_JIT for a shape that qualifies for run-time compilation (and
 * the mechanism is on), FCCQP_ORIGIN_GENERIC_ONLY otherwise; negative status codes as fccqp_kernel_for_shape.
 * fccqp_kernel_for_shape answers FCCQP_KERNEL_FUSED / _COOP for a FCCQP_ORIGIN_JIT shape: what AUTO runs once the
 * kernel is compiled. */
int fccqp_kernel_origin_for_shape(int num_vars, int num_equality_constraints, int nc, int lambda_c_start);
/* Compiles the specialised kernel of a shape into the disk cache now (or finds it there): what fccqp_create would do,
 * without a handle and without a GPU -- e.g. at install time.  Returns FCCQP_KERNEL_FUSED / FCCQP_KERNEL_COOP, or a
 * negative status code (-FCCQP_ERR_UNSUPPORTED: the shape has no run-time kernel or compilation failed). */
int fccqp_jit_prepare(int num_vars, int num_equality_constraints, int nc, int lambda_c_start);
/* FCCQP_ORIGIN_* of the specialised kernel this handle holds (FCCQP_ORIGIN_GENERIC_ONLY: none). */
int fccqp_get_kernel_origin(const fccqp_solver *s, int *origin);
int fccqp_set_jit(int enabled);          /* process-wide; affects handles created afterwards */
/* tests: forget the code objects held in memory, so that the next create reads the disk cache */
int fccqp_debug_jit_forget(void);
int fccqp_get_shape(const fccqp_solver *s, int *num_vars, int *num_eq, int *nc,
                    int *lambda_c_start, int *batch, int *device);

/* ---- solve        (replaces FCCQP::Solve, reference src/fcc_qp.cpp:114-191) ---------------- */
/* Host-pointer form: borrows the seven host arrays for the duration of the call (as the
 * reference borrows its Eigen::Ref arguments), copies them to the device, solves the whole
 * batch and waits.  n_friction = coefficients supplied PER INSTANCE (must be >= nc/3; the row
 * stride of friction_coeffs is n_friction).
 * Where the data goes through: arrays that are page-locked (hipHostMalloc, fccqp_host_register) are read by the DMA
 * engines in place; anything else is copied by the CPU into page-locked buffers of the handle first (two 8 MiB bounce
 * buffers, double-buffered against the DMA; small batches: one gather block) -- the GPU never addresses pageable caller
 * memory.  (Handing pageable arrays of 1 MiB and more to hipMemcpy makes the HIP runtime page-lock the caller's pages on
 * the fly and cache that lock by address; the two process aborts this library has seen -- a GPU memory fault on a heap
 * address -- were on that path: DESIGN.md section 7.3.)  The same holds for every entry point that takes or fills a
 * caller array: fccqp_solve_f32, fccqp_get_solution, fccqp_get_state / fccqp_set_state.  Pageable arrays cost a CPU
 * copy by up to four threads (measured: 42 GB/s into the device against 52 GB/s from page-locked arrays, 0.86 against
 * 1.06 M walking-shape solves/s at batch 4096); page-lock them, or use the device-pointer form, where the transfer rate matters. */
int fccqp_solve(fccqp_solver *s, const double *Q, const double *b, const double *A_eq,
                const double *b_eq, const double *friction_coeffs, int n_friction,
                const double *lb, const double *ub);

/* Asynchronous host-pointer form: the copies and the kernel(s) are enqueued on the solver's own
 * stream and the call returns at once.  The seven host arrays must stay valid and unmodified until
 * fccqp_synchronize() / fccqp_get_solution() on this handle, and should be page-locked
 * (fccqp_host_register below, or hipHostMalloc) -- pageable arrays are copied by the CPU through the
 * handle's bounce buffers inside the call (see fccqp_solve) and only the kernel overlaps.  This is what lets a caller keep several handles in flight:
 * the H2D copy of one batch, the kernel of another and the D2H copy of a third overlap (bench.py's
 * `end_to_end` leg; SURVEY.md 8d timing protocol).  n_friction must equal nc/3 here.
 * Small batches (staged inputs of at most 1 MiB, e.g. up to ~20 walking-shape instances) are gathered
 * into one page-locked block inside the handle first: for those the call WAITS for the handle's
 * previous kernel (it reads the block's device twin) before it returns, and the caller's arrays are
 * free again as soon as it has. */
int fccqp_solve_async(fccqp_solver *s, const double *Q, const double *b, const double *A_eq,
                      const double *b_eq, const double *friction_coeffs, int n_friction,
                      const double *lb, const double *ub);
/* The same for fp32-STORAGE inputs (see fccqp_solve_f32 below): all seven arrays travel as float -- half the
 * PCIe bytes of the fp64 form, which is what bounds a host-fed pipeline -- and the small vectors are widened
 * on the device.  Results are bit-identical to fccqp_solve_f32 on the same arrays.  Shapes with
 * num_vars^2 < 3 num_vars + num_equality_constraints + nc/3 are not supported here (FCCQP_ERR_UNSUPPORTED). */
int fccqp_solve_async_f32(fccqp_solver *s, const float *Q, const float *b, const float *A_eq,
                          const float *b_eq, const float *friction_coeffs, int n_friction,
                          const float *lb, const float *ub);
/* Page-lock / release a host range for DMA (thin wrappers over hipHostRegister/Unregister, so
 * that a caller of this ABI needs no HIP headers).  The reference has no counterpart: its inputs
 * never leave the host (src/fcc_qp.cpp:146-150 copies them into Eigen members). */
int fccqp_host_register(void *ptr, size_t bytes);
int fccqp_host_unregister(void *ptr);

/* Device-pointer form: the seven arrays are already resident in HBM on the solver's device
 * (instance-major, friction stride nc/3).  Enqueues on `stream` and returns without waiting.
 * `stream` is a hipStream_t: NULL is HIP's default (null) stream, as everywhere in HIP;
 * FCCQP_STREAM_OWN selects the solver's private non-blocking stream.  The arrays need only the
 * natural alignment of their element type (the kernels read Q and A_eq one element per lane). */
#define FCCQP_STREAM_OWN ((void *)(intptr_t)-1)
int fccqp_solve_device(fccqp_solver *s, const double *dQ, const double *db, const double *dA_eq,
                       const double *db_eq, const double *dfriction, const double *dlb,
                       const double *dub, void *stream);

/* fp32-STORAGE forms (BASELINE.json configs[2]).  Same shapes and layout, float arrays.  Every
 * value is widened to double as it is loaded and all arithmetic stays fp64, so the result is bit
 * for bit that of fccqp_solve on the widened values, while the HBM / PCIe traffic of the call is
 * halved (Q and A_eq, 98% of the bytes, are read as float by the kernels; the small vectors are
 * widened into solver-owned buffers first).  The reference has no float path (Eigen::MatrixXd
 * throughout, src/fcc_qp.hpp:114-117); these replace the same code as fccqp_solve
 * (src/fcc_qp.cpp:114-191) for callers that log or generate QPs in float. */
int fccqp_solve_f32(fccqp_solver *s, const float *Q, const float *b, const float *A_eq,
                    const float *b_eq, const float *friction_coeffs, int n_friction,
                    const float *lb, const float *ub);
int fccqp_solve_device_f32(fccqp_solver *s, const float *dQ, const float *db, const float *dA_eq,
                           const float *db_eq, const float *dfriction, const float *dlb,
                           const float *dub, void *stream);
int fccqp_synchronize(fccqp_solver *s);

/* ---- results      (replaces FCCQP::GetSolution, reference src/fcc_qp.cpp:194-207) ----------- */
/* Copies z[batch][n] and details[batch] to the host (either may be NULL).  Waits for the solve. */
int fccqp_get_solution(fccqp_solver *s, double *z, fccqp_details *details);

/* z only, into page-locked host memory, enqueued behind the solve on the solver's stream; returns
 * at once.  fccqp_synchronize() waits for it. */
int fccqp_get_z_async(fccqp_solver *s, double *z_pinned);

/* Device-resident results of the last solve (valid until the next solve / destroy):
 * z[batch][n] (f64); n_iter[batch], flags[batch] (i32);
 * metrics[4][batch] (f64) = residual_bounds, residual_cone, bounds_viol, cone_viol;
 * factor_ticks[batch] (i64, 100 MHz device clock ticks).  Any out pointer may be NULL. */
int fccqp_get_solution_device(fccqp_solver *s, const double **z, const int **n_iter,
                              const int **flags, const double **metrics,
                              const long long **factor_ticks);

/* ---- warm-start state (x, mu_x, mu_lambda_c of reference src/fcc_qp.hpp:145-150) ------------ */
int fccqp_get_state(fccqp_solver *s, double *x, double *mu_x, double *mu_lambda_c);
int fccqp_set_state(fccqp_solver *s, const double *x, const double *mu_x,
                    const double *mu_lambda_c);

/* ---- measurement -------------------------------------------------------------------------- */
/* Milliseconds the last solve's kernel(s) took on the device (hipEvent pair on the launch
 * stream); waits for the solve.  Exception: a synchronous host-pointer solve of a small batch (staged
 * inputs <= 1 MiB: the one-QP-per-tick path) records no events -- each costs ~5 us of stream time, a tenth
 * of such a solve -- and reports the host-measured time from the launch to the completion of the device
 * work instead (a few microseconds more than the kernel itself). */
int fccqp_last_kernel_ms(fccqp_solver *s, float *ms);
/* The same figure together with HOW it was measured, so that a caller does not average two different things:
 *   FCCQP_TIMING_DEVICE_EVENTS  HIP events around the solve's launches on their stream (device time);
 *   FCCQP_TIMING_HOST_CLOCK     small synchronous host-pointer solves record no events (each costs ~5 us of stream
 *                               time): host clock from the launch to the completion of the device work, i.e. launch
 *                               latency, kernel, the copy of the results and, if an instance was flagged, the COD pass;
 *   FCCQP_TIMING_NONE           nothing was measured: fccqp_set_kernel_timing(s, 0) and a solve that records events
 *                               (device-pointer, asynchronous, or a large host-pointer batch); *ms = 0.  The switch
 *                               is about the start-of-kernel EVENT: a small synchronous host-pointer solve records
 *                               none anyway and keeps reporting its host clock (FCCQP_TIMING_HOST_CLOCK), which costs
 *                               nothing. */
enum { FCCQP_TIMING_NONE = 0, FCCQP_TIMING_DEVICE_EVENTS = 1, FCCQP_TIMING_HOST_CLOCK = 2 };
int fccqp_last_timing(fccqp_solver *s, float *ms, int *measured_by);
/* Mean device time (seconds) the instances of the last solve spent per phase (fused kernel; the generic
 * kernel fills the slots it has): out[0] issue of the matrix loads (assembly on the generic kernel),
 * [1] factorisations (K_rho with the fused forward solve, and W), [2] presolve (W's factorisation and the
 * two sweeps), [3] Gram product: rho G, x_c, W, [4] ADMM iterations, [5] whole kernel per instance;
 * [6],[7] reserved.  Profiling aid: the first call enables the stamps for subsequent solves and returns
 * zeros. */
int fccqp_phase_seconds(fccqp_solver *s, double out[8]);
/* The raw stamps behind fccqp_phase_seconds: out[p * batch + i], p = 0..7, 100 MHz ticks; slot 7 is the
 * device clock at which instance i started, slot 5 its duration (so a launch can be laid out on a time
 * line).  count <= 8 * batch; the stamps must have been enabled by a fccqp_phase_seconds call before the
 * solve.  Debugging aid. */
int fccqp_debug_phase_ticks(fccqp_solver *s, long long *out, long long count);
/* Compulsory HBM bytes one solve of the whole batch moves:
 * 8 * batch * (n^2 + m*n + 4n + m + nc/3), plus the warm-start state when warm.  After an
 * fp32-storage solve the seven inputs count 4 bytes per element (z and the state stay 8). */
int fccqp_algorithmic_bytes(const fccqp_solver *s, long long *bytes);

/* ---- debugging aids ------------------------------------------------------------------------- */
/* The block -> instance map of a batch launch (fccqp_device.h: a fixed pseudo-random bijection of blockIdx.x, so
 * that a workload whose cost is periodic in the instance index does not load the XCDs / CUs unevenly): the instance
 * that workgroup `block` of a launch of `batch` instances solves, or -1 if block >= batch.  Pure integer arithmetic,
 * no device needed: the host tests check the bijection with it. */
int fccqp_debug_block_instance(int batch, int block);

/* Copies up to `count` doubles of the solver's global workspace to the host (the per-block KKT
 * factor of shapes too large for LDS, N > 198; the staged rho*G of the generic kernel's operator
 * iterations otherwise) and returns the number available in *available (0: no workspace).  Lets a
 * test compare the factor a kernel left behind with one computed on the host.  Environment:
 * FCCQP_DEBUG_POISON=1 fills every fresh device allocation with NaN bit patterns. */
int fccqp_debug_read_workspace(fccqp_solver *s, double *out, long long count, long long *available);

#ifdef __cplusplus
}
#endif
#endif /* FCCQP_H */
