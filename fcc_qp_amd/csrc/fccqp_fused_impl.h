// fccqp_fused_impl.h -- shape-specialised gfx950 kernel: ONE WAVEFRONT PER QP, four one-wave
// workgroups per CU (one per SIMD), the whole solve fused in one launch.
//
// Hot path replaced (reference /root/reference):
//   FCCQP::Solve   src/fcc_qp.cpp:114-191   FCCQP::DoADMM  src/fcc_qp.cpp:57-112
//   projections    src/constraint_utils.cpp:5-65      Eigen LDLT call sites fcc_qp.cpp:68,86,161,174
//
// Why it looks the way it does (numbers from tools/fp64_issue_probe.hip, mfma_f64_probe.hip, vgpr_bank_probe.hip
// and lane_ops_probe.hip on MI355X):
//   * one wave per SIMD: a v_fma_f64 issues every 5.8-8.2 cycles (repeated / fresh register operands) whether
//     or not it depends on the previous one, a v_mfma_f64_16x16x4 (1024 FMAs) every 64 cycles back to back:
//     the matrix cores are the cheaper way to the same 78.6 TFLOP/s fp64 unit (they do NOT run beside the
//     fp64 VALU: the two serialise), and -- what matters -- they take their operands ONE ELEMENT PER LANE
//     and hand the result tile back as the next product's operand.
//   * an operand every lane needs costs an LDS broadcast (2 LDS cycles per double per CU), a
//     v_readlane pair (~9 cycles per double) -- or nothing: the fp64 ALU's one DPP mode, row_newbcast,
//     lets an FMA read lane k of its 16-lane row.  Every sequential piece of the path (diagonal-tile
//     elimination, tile inverse, the mat-vec of an ADMM iteration) is laid out so that the operand it
//     needs sits in a lane of the same row.
//   * 512 registers per lane (256 VGPR + 256 AGPR, MFMA reads and writes both) hold the upper triangle
//     of the leading 96 x 96 block of the KKT matrix as 21 accumulator tiles plus the right-hand-side
//     tiles; LDS (32 KB per QP) holds exchange scratch, rho G and the presolve factor -- never the matrix.
//
// Phases, N = n + m (walking shape: n = 60, m = 38, N = 98):
//   P0  K_rho = [[Q + rho I, A^T],[A, 0]] (fcc_qp.cpp:146-150,62-64) from HBM straight into accumulator-layout
//       tiles, ONCE (issue_kkt_loads: only the lower triangle of Q is read; the loads of row panel 0
//       first, so that its diagonal tile is factorised while the rest is in flight).  The last N % 16
//       rows, when they are few, are not tiled but bordered (Shape::kBorder).
//   P1  factorise K_rho (fcc_qp.cpp:68) with the forward solve of [e_0 .. e_{n-1} | -b; b_eq] fused in
//       (tile_factor_forward): unpivoted, natural order, right-looking by 16 x 16 register tiles in
//       the form U^T D U, in which every trailing update contracts over the ROWS of the panel's
//       tiles -- both MFMA operands are accumulator registers, nothing is read from or written to
//       LDS except the 16 x 16 diagonal tile, which takes a trip to a column-per-lane layout to be
//       eliminated and inverted with DPP-fed FMAs (tile_ldl).  Pivot rule: 1/d_j = 0 for
//       |d_j| <= DBL_MIN (Eigen's pseudo-inverse rule in LDLT::solve).
//   P3  rho G = rho (K_rho^-1)[0:n,0:n] and x_c = (K_rho^-1 [-b; b_eq])[0:n] as the Gram product
//       Y^T D^-1 Y of the forward result (gram_output), then through LDS to the layout of P4: a lane
//       holds one row of rho G and one entry of x_c.  The tiles of W = I - rho G stay in registers.
//   Pw  presolve, skipped on warm start (fcc_qp.cpp:159-178): the reference's x0 = (K^-1 [-b; b_eq])
//       .head(n) without a factorisation of K: x0 = (I - rho G)^-1 x_c (Woodbury; see the kernel
//       body); W is factorised by the same register tiles, its factor written to LDS for the two
//       triangular sweeps of the single right-hand side.  One assembly, one pass over HBM.
//   P4  ADMM iterations (fcc_qp.cpp:79-110): the KKT solve of an iteration is
//       x = x_c + (rho G) w,  w = x_bar - mu_x (lambda_c segment overwritten, fcc_qp.cpp:82):
//       a 60x60 mat-vec (DPP-fed FMAs, one LDS exchange of w); projections, residuals, dual update
//       and the exit test are per-lane work, quad_perm moves inside a contact and one ballot.
//   P5  epilogue metrics (fcc_qp.cpp:184-186) and write-back of z and the warm-start state.
//
// Pivot order: natural (no pivoting).  For rho > 0, Q >= 0 and A_eq of full row rank K_rho is
// symmetric quasi-definite ((1,1) block positive definite, Schur complement negative definite), so
// the factorisation exists in every symmetric order; the reference's Eigen LDLT orders pivots by the
// largest remaining diagonal entry, which only reorders the same block elimination.  A duplicated
// equality row gives an exactly zero pivot with an exactly zero column, in Eigen as here; both drop
// it (pseudo-inverse rule).  Parity with the reference-order oracle is ~1e-13 relative on the
// walking-shape workloads.
#pragma once
// (run-time compilation, fccqp_jit.cpp: hiprtc has no system headers; what this file needs of them is spelled out)
#ifndef __HIPCC_RTC__
#include <cfloat>
#else
#ifndef DBL_MIN
#define DBL_MIN 2.2250738585072014e-308
#endif
#ifndef HUGE_VAL
#define HUGE_VAL __builtin_huge_val()
#endif
#endif

#include "fccqp_device.h"

namespace fccqp {
namespace fused {

constexpr int kWave = 64;
// compile-time loop index handed to the body of static_for (std::integral_constant without <type_traits>)
template <int V> struct int_c { static constexpr int value = V; };
// presolve: smallest pivot of W = I - rho G (eigenvalues in (0, 1]) still taken as non-singular.
// A pivot p amplifies the ~1e-16 error of rho G by 1/p; below this the instance goes to the COD
// fallback, whose minimum-norm solution is what the reference returns for a singular K.
constexpr double kPresolveMinPivot = 1e-10;
using d2 = double __attribute__((ext_vector_type(2)));
// explicit LDS address space: keeps every access a ds_* instruction (a generic pointer decays to
// flat_load / flat_store as soon as it goes through a select or a volatile cast)
using lds_f64 = __attribute__((address_space(3))) double;
using lds_d2 = __attribute__((address_space(3))) d2;

template <int I, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < E) {
    f(int_c<I>{});
    static_for<I + 1, E>(f);
  }
}

__device__ __forceinline__ void lds_fence() {
  // one-wave workgroup: LDS ops of a wave complete in order; this only pins the compiler's order
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Development aid: -DFCCQP_FINE_STAMPS reports (100 MHz ticks, per QP) the time in the diagonal tiles of
// K_rho (phase slot 0) and of W (slot 2) and in the Gram product with its stores (slot 7); tools/quick_bench.py
// prints the slots.
#ifdef FCCQP_FINE_STAMPS
#define FCCQP_FINE(slot, expr) do { const long long t_ = wall_clock64(); expr; if (fine) fine[slot] += wall_clock64() - t_; } while (0)
#else
#define FCCQP_FINE(slot, expr) do { expr; } while (0)
#endif
__device__ __forceinline__ double bcast_lane(double v, int src_lane) {  // src_lane wave-uniform
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// Selects on 64-bit values, written out as VOP3-encoded v_cndmask_b32 with the lane mask in an SGPR pair.  Left to
// itself the compiler turns a compare with one user into v_cmp -> VCC and the VOP2 form `v_cndmask_b32_e32 ..., vcc`,
// and on gfx950 that form costs a lone wave 19 cycles an instruction where every other VALU instruction -- the VOP3
// form with the very same VCC as its mask included -- costs 5 to 7 (tools/issue_probe.hip).
__device__ __forceinline__ double select_lanes(const unsigned long long mask, const double if_set, const double if_clear) {
  int lo, hi;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"(__double2loint(if_clear)), "v"(__double2loint(if_set)), "s"(mask));
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"(__double2hiint(if_clear)), "v"(__double2hiint(if_set)), "s"(mask));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double select_lanes_or_zero(const unsigned long long mask, const double if_set) {
  int lo, hi;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(lo) : "v"(__double2loint(if_set)), "s"(mask));
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(hi) : "v"(__double2hiint(if_set)), "s"(mask));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int select32_lanes(const unsigned long long mask, const int if_set, const int if_clear) {
  int r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
  return r;
}
__device__ __forceinline__ double select64(const bool c, const double if_true, const double if_false) {
  return select_lanes(__builtin_amdgcn_ballot_w64(c), if_true, if_false);
}
__device__ __forceinline__ double max2(double a, double b) { return select64(a < b, b, a); }  // std::max
__device__ __forceinline__ double min2(double a, double b) { return select64(b < a, b, a); }  // std::min

// ---- DPP wave reductions (no LDS traffic).  gfx9 DPP controls: quad_perm 0x00..0xff,
// row_half_mirror 0x141, row_mirror 0x140, row_bcast15 0x142, row_bcast31 0x143 ----------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get(double v, double old) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
// every lane reads lane K of its quad (all lanes active: no `old` value, no copy)
template <int K>
__device__ __forceinline__ double quad_bcast(double v) {
  constexpr int ctrl = K * 0x55;   // quad_perm [K,K,K,K]
  int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
  int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// component K of this lane's contact force: lane K of the quad, or -- one contact per 16-lane row -- of the row
template <int K, bool ROW>
__device__ __forceinline__ double contact_bcast(double v) {
  if constexpr (ROW) return __builtin_amdgcn_mov_dpp(v, 0x150 + K, 0xf, 0xf, true);    // v_mov_b64_dpp row_newbcast:K
  else return quad_bcast<K>(v);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_get<0xB1, 0xf>(v, 0.0);   // quad_perm [1,0,3,2]
  v += dpp_get<0x4E, 0xf>(v, 0.0);   // quad_perm [2,3,0,1]
  v += dpp_get<0x141, 0xf>(v, 0.0);  // row_half_mirror
  v += dpp_get<0x140, 0xf>(v, 0.0);  // row_mirror: every lane holds its row-of-16 sum
  v += dpp_get<0x142, 0xa>(v, 0.0);  // row_bcast15 -> rows 1,3
  v += dpp_get<0x143, 0xc>(v, 0.0);  // row_bcast31 -> rows 2,3
  return bcast_lane(v, 63);
}
__device__ __forceinline__ double wave_max(double v) {
  v = max2(v, dpp_get<0xB1, 0xf>(v, v));
  v = max2(v, dpp_get<0x4E, 0xf>(v, v));
  v = max2(v, dpp_get<0x141, 0xf>(v, v));
  v = max2(v, dpp_get<0x140, 0xf>(v, v));
  v = max2(v, dpp_get<0x142, 0xa>(v, v));
  v = max2(v, dpp_get<0x143, 0xc>(v, v));
  return bcast_lane(v, 63);
}

// reference: project_to_friction_cone(Vector3d, mu), constraint_utils.cpp:5-25, evaluated for ONE
// component of the contact force (every lane of a contact triple calls it with the same
// (fx, fy, fz) and its own component v = f[comp]).  Same three branches and the same f_z == 0
// quirk.  The side-of-cone branch of the reference normalises the ray r = (k f_x, k f_y, f_z),
// k = mu f_z / |f_xy|, and returns (r^ . f) r^ (:20-23).  With |r|^2 = f_z^2 (1 + mu^2) and
// r . f = f_z (mu |f_xy| + f_z) the same vector is
//     t (mu f_xy / |f_xy|, 1),     t = (mu |f_xy| + f_z) / (1 + mu^2)
// -- the textbook second-order-cone projection: one reciprocal square root per iteration
// (v_rsq_f64 + two Newton steps, <= 1-2 ulp) and a per-lane constant kc = 1 / (1 + mu^2) (z
// component) or mu / (1 + mu^2) (x, y) computed once, instead of the rsqrt + reciprocal + ~12 more
// dependent operations of the literal form, on every lane of the wave on every ADMM iteration.
// f_z == 0 exactly: Eigen's normalize() leaves the zero ray alone and the reference returns 0.
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
#ifdef FCCQP_RCP_TWO_NEWTON
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
#else
  // one cubic step, r <- r (1 + e + e^2), e = 1 - d r: three dependent operations instead of the four of two
  // Newton steps, on the critical path of every pivot of a diagonal tile (the raw v_rcp_f64 is good to ~1e-8)
  const double e = fma(-d, r, 1.0);
  r = fma(r, fma(e, e, e), r);
#endif
  return r;
}
__device__ __forceinline__ double fast_rsqrt(double a) {  // a > 0
  double r = __builtin_amdgcn_rsq(a);
#ifdef FCCQP_RSQRT_TWO_NEWTON
  // Newton for 1/sqrt(a): r <- r + r * (1 - a r^2) / 2, twice (8 operations)
  double e = fma(-a * r, r, 1.0);
  r = fma(0.5 * r, e, r);
  e = fma(-a * r, r, 1.0);
  r = fma(0.5 * r, e, r);
#else
  // one Halley step, r <- r (1 + e (1/2 + 3/8 e)), e = 1 - a r^2: cubic, so the raw v_rsq_f64 (good to 5e-8) lands
  // at ~1e-22 before rounding -- the same <= 1-2 ulp as two Newton steps in 5 operations instead of 8 (the projection
  // runs on every lane in every ADMM iteration)
  const double e = fma(-a * r, r, 1.0);
  r = fma(r, e * fma(0.375, e, 0.5), r);
#endif
  return r;
}
__device__ __forceinline__ double cone_lane_constant(double mu, bool is_z) {
  const double c = 1.0 / fma(mu, mu, 1.0);
  return is_z ? c : mu * c;
}
__device__ __forceinline__ double project_cone_component(double fx, double fy, double fz, double mu, double kc,
                                                         double v, const unsigned long long z_lanes) {
  const double nxy2 = fx * fx + fy * fy;
  // f_xy == 0 (inside if mu f_z >= 0, else the origin): the clamp keeps the rsqrt finite, and then
  // nxy = 0 * rs = 0 and dir = rs * 0 = 0 exactly, with no select
  const double rs = fast_rsqrt(__builtin_fmax(nxy2, DBL_MIN));
  const double nxy = nxy2 * rs;
  const bool inside = mu * fz >= nxy;                   // :10-12
  // :15-17, f_z < -mu |f_xy|, read off the numerator of t (one rounding instead of the product's: the two differ
  // only where f_z + mu |f_xy| vanishes to the last bit, and there the other branch returns t dir = O(1e-16) |f|)
  const double tnum = fma(mu, nxy, fz);
  const bool polar = tnum < 0.0;
  const double t = tnum * kc;                           // :20-23, see above
  const double dir = select_lanes(z_lanes, 1.0, rs * v);   // (VOP3 select on the loop-invariant lane mask of the z components)
  // zero ray (f_z == 0): Eigen's normalize() no-op -> 0, like the polar cone.  Straight-line VOP3 selects: whether the
  // backend makes this a branch around the outside-the-cone work or two VOP2 selects on VCC changed from build to build
  const unsigned long long keep = __builtin_amdgcn_ballot_w64(fz != 0.0) & __builtin_amdgcn_ballot_w64(!polar);
  const double outside = select_lanes_or_zero(keep, t * dir);
  return select64(inside, v, outside);
}

// The same projection in fewer issue slots, for the one-wave kernel's ADMM loop (every instruction of that loop is
// 0.27 % of a solve): |f_xy|^2 with one fma; the polar cone by clamping the numerator of t at zero (t = 0: the
// origin) instead of a compare, a mask operation and part of a select; the z lanes' direction 1 by a per-lane pair
// of constants (dxy, dz) = (1, 0) on x / y lanes, (0, 1) on z lanes, dir = (rs v) dxy + dz, instead of a select.
// Same values as project_cone_component for finite input, up to the sign of a zero result and the one rounding
// the fma saves.
__device__ __forceinline__ double project_cone_component_lean(double fx, double fy, double fz, double mu, double kc,
                                                              double v, double dxy, double dz) {
  const double nxy2 = fma(fy, fy, fx * fx);
  const double rs = fast_rsqrt(__builtin_fmax(nxy2, DBL_MIN));
  const double nxy = nxy2 * rs;
  const bool inside = mu * fz >= nxy;                   // :10-12
  const double t = __builtin_fmax(fma(mu, nxy, fz), 0.0) * kc;   // :15-17 (polar cone: 0) and :20-23
  const double dir = fma(rs * v, dxy, dz);
  // zero ray (f_z == 0): Eigen's normalize() no-op -> 0
  const double outside = select_lanes_or_zero(__builtin_amdgcn_ballot_w64(fz != 0.0), t * dir);
  return select64(inside, v, outside);
}

__device__ __forceinline__ double pivot_inverse(double dd) {   // Eigen's LDLT::solve drops |d| <= DBL_MIN
  const double r = fast_rcp(dd);
  return select_lanes_or_zero(__builtin_amdgcn_ballot_w64(fabs(dd) > DBL_MIN), r);
}
// a / d from r = pivot_inverse(d): the product a r corrected by one residual step, i.e. the correctly rounded
// quotient in all but rare cases and EXACTLY 1 for a == d.  That is what makes a duplicated equality row
// eliminate to an exactly zero column and an exactly zero pivot, as it does with the reference's true
// division (Eigen LDLT: A21 /= pivot), instead of a pivot of rounding-error size whose reciprocal
// wrecks the solve; r == 0 (dropped pivot) stays 0.
__device__ __forceinline__ double pivot_quotient(double a, double d, double r) {
  const double q = a * r;
  return fma(fma(-d, q, a), r, q);
}
// (fast_rcp above is also the pivot reciprocal of the factorisation: <= 1 ulp from the correctly
// rounded quotient, a third of the dependent-instruction chain of an IEEE division.)

// ---- compile-time shape description ------------------------------------------------------------
// Packed lower triangle of an NN x NN symmetric matrix, column-major: element (i,j), i >= j, lives
// at cbase(j) + i.  Column starts are padded so that cbase(j) is even: an element's address parity
// is its row parity and ds_read_b128 of rows (2r, 2r+1) is always 16-byte aligned.
template <int NN>
struct Packed {
  static constexpr int N = NN;
  static constexpr int cstart(int j) {
    return j * N - (j * (j - 1)) / 2 + ((N % 2 == 0) ? (j + 1) / 2 : j / 2);
  }
  static constexpr int cbase(int j) { return cstart(j) - j; }
  // cbase(j + 1) - cbase(j): lets loops walk the columns with one scalar add
  static constexpr int cstep(int j) { return N - j - 1 + (((j & 1) == (N & 1)) ? 1 : 0); }
  static constexpr int kPacked = (cstart(N) + 1) & ~1;
  static constexpr bool kZeroFilled = false;
  // exchange scratch of the diagonal tiles (tile_factor_forward, tile_factor_store): the tile, its inverse,
  // 16 reciprocal pivots (+ padding)
  // (columns 18 doubles apart, not 16: with a stride of 128 bytes the sixteen lanes that store or load one row of
  // the tile hit one LDS bank -- 16-way conflicts on every store, 8-way on every 16-byte load, 500 LDS cycles per
  // tile instead of 50; 144 bytes keeps the 16-byte alignment of the loads)
  static constexpr int kTileStride = 18, kTileLen = 16 * kTileStride;
  static constexpr int kScratch = 2 * kTileLen + 32;
  static constexpr int kRowsPadded = 16 * ((N + 15) / 16);
  static constexpr int kNp = (N + 1) & ~1;
};

// The factor of W = I - rho G as the presolve's two sweeps want it: a full NN x NN square, column-major with an odd
// column stride (a lane walking along its own column -- the backward sweep -- then hits all LDS banks), L strictly
// below the diagonal and ZEROS on and above it, so that a sweep step multiplies by the entry it loads without a
// "row below the pivot?" select, and the reciprocal pivots in NN slots behind the square.  Element (i, j), i > j,
// at cbase(j) + i as in Packed.
template <int NN>
struct ZeroFilledSquare {
  static constexpr int N = NN;
  static constexpr bool kZeroFilled = true;
  static constexpr int kStride = NN | 1;
  static constexpr int cbase(int j) { return j * kStride; }
  static constexpr int kDinv = (NN * kStride + 1) & ~1;
  static constexpr int kPacked = (kDinv + NN + 1) & ~1;
};

template <int N_VARS, int N_EQ, int NC, int LC_START>
struct Shape : Packed<N_VARS + N_EQ> {
  using KKT = Packed<N_VARS + N_EQ>;
  using KKT::N; using KKT::kNp;
  // the packed n x n factor of W = I - rho G for the presolve's two sweeps (see fused_kernel)
  using Reduced = ZeroFilledSquare<N_VARS>;
  static constexpr int n = N_VARS, m = N_EQ, nc = NC, s = LC_START;
  static_assert(n >= 1 && n < kWave, "one lane per variable plus one lane for the rhs column");
  static_assert(m >= 0 && m <= kWave, "one lane per equality row when the right-hand side is staged");
  static_assert(nc % 3 == 0 && s + nc <= n, "fcc_qp.cpp:32-33");
  // nc == 0 with all-infinite bounds is the reference's equality-only case (fcc_qp.cpp:132-133: one
  // direct solve, no ADMM); such shapes go to the generic kernel, which keeps the direct presolve
  static_assert(nc > 0, "fused shapes have contacts");
  // where tile_factor_forward leaves 1/d_row for gram_output: behind the exchange scratch and behind the rho G
  // image (n x n, row-major) that gram_output writes
  // (row stride of that image: even -- the rows are read back 16 bytes at a time -- and 2 mod 4, so that the mirror
  // stores of a tile, which walk down a column, spread over eight LDS banks instead of four or fewer)
  static constexpr int kImgStride = ((n + 1) & ~1) + ((((n + 1) & ~1) % 4 == 2) ? 0 : 2);
  static constexpr int kRcpOffset = (n * kImgStride > KKT::kScratch) ? n * kImgStride : KKT::kScratch;
  // Bordering.  16 x 16 tiles waste most of a last row block that holds only N % 16 rows (walking shape:
  // 2 of 16, and a fifth of all MFMAs touch it).  When those last rows lie in the zero block and the
  // right-hand-side tile column that carries [-b; b_eq] has free columns for them, the tiles factorise
  // only the leading N1 = N - kBorder rows, the border columns K_rho[0:N1, N1:N] ride along as extra
  // right-hand sides, and the last kBorder pivots are the LDL^T of the small Schur complement read off
  // the Gram product (gram_output) -- the same elimination order, the same pivots.
  static constexpr int kBorder =
      (N % 16 != 0 && N - N % 16 >= n && (n % 16) + (N % 16) <= 15 && N % 16 <= 3) ? N % 16 : 0;
  static constexpr int N1 = N - kBorder;
  static constexpr int kTileRows = 16 * ((N1 + 15) / 16);
  // LDS staging of the border columns (kBorder x kTileRows, behind the reciprocal pivots) and, later, of
  // the Gram columns n .. n + kBorder (rows 0 .. n + kBorder)
  static constexpr int kBorderOffset = kRcpOffset + KKT::kRowsPadded;
  // (the Gram columns are stored by whole row blocks: 16 (n / 16 + 1) rows)
  static constexpr int kBorderDoubles =
      (kBorder * kTileRows > 16 * (n / 16 + 1) * (kBorder + 1)) ? kBorder * kTileRows : 16 * (n / 16 + 1) * (kBorder + 1);
  // LDS carve (doubles): work area M[kMat] | vbuf[max(kNp, 64)].  M holds, in turn: the exchange scratch,
  // 1/d and the border staging during the factorisation; the rho G image; the packed factor of W and its
  // exchange scratch.  (The matrix itself never is in LDS: it lives in registers.)
  static constexpr int kMatFactor = kBorderOffset + kBorderDoubles;
  static constexpr int kMatPresolve = ((Reduced::kPacked + 1) & ~1) + KKT::kScratch;
  static constexpr int kMat = (((kMatFactor > kMatPresolve) ? kMatFactor : kMatPresolve) + 1) & ~1;
  static constexpr int kVbuf = (kNp > kWave) ? kNp : kWave;
  static constexpr int kLdsDoubles = kMat + kVbuf;
  static_assert(kLdsDoubles * 8 <= 160 * 1024 / 4, "four one-wave workgroups per CU");
  // Lane <-> variable map of the ADMM loop.  A lane gets the three components of its contact's force with DPP
  // moves instead of an LDS round trip per iteration: contact c owns lanes 16c, 16c+1, 16c+2 (x, y, z) of 16-lane row c
  // when there are at most four contacts (kRowContacts below), else lanes 4c, 4c+1, 4c+2 of quad c.  The other
  // variables fill the remaining lanes in increasing order.  -1: the lane has no variable.
  static constexpr int kContacts = nc / 3;
  static_assert(4 * kContacts <= kWave, "one quad of lanes per contact");
  // Up to four contacts: contact c owns lanes 16c, 16c+1, 16c+2 -- one contact per 16-lane row -- and a lane gets a
  // component of its contact's force with ONE v_mov_b64_dpp (row_newbcast, the fp64 DPP mode) instead of the two
  // v_mov_b32 quad_perm moves a 64-bit value needs in the quad layout: three issue slots fewer per ADMM iteration.
  static constexpr bool kRowContacts = kContacts <= 4;
  __device__ static __forceinline__ bool lane_is_cone(const int lane) {
    return kRowContacts ? ((lane >> 4) < kContacts && (lane & 15) < 3) : ((lane >> 2) < kContacts && (lane & 3) < 3);
  }
  __device__ static __forceinline__ int lane_contact(const int lane) { return kRowContacts ? (lane >> 4) : (lane >> 2); }
  __device__ static __forceinline__ int lane_component(const int lane) { return kRowContacts ? (lane & 15) : (lane & 3); }
  __device__ static __forceinline__ int lane_variable(const int lane) {
    if constexpr (kRowContacts) {
      const int row = lane >> 4, pos = lane & 15;
      const bool in_row = row < kContacts;
      // contact lanes with a smaller index than this one
      const int below = 3 * (in_row ? row : kContacts) + (in_row ? ((pos < 3) ? pos : 3) : 0);
      const int k = lane - below;                            // index among the non-contact variables
      const int other = (k < s) ? k : k + nc;
      return (in_row && pos < 3) ? s + 3 * row + pos : ((k < n - nc) ? other : -1);
    } else {
      const int q = lane >> 2, r = lane & 3;
      const bool in_quad = q < kContacts;
      const int k = in_quad ? q : lane - 3 * kContacts;      // index among the non-contact variables
      const int other = (k < s) ? k : k + nc;
      return (in_quad && r < 3) ? s + 3 * q + r : ((k < n - nc) ? other : -1);
    }
  }
};


using d4 = double __attribute__((ext_vector_type(4)));

// rho G = rho (K_rho^-1)[0:n,0:n] and x_c = (K_rho^-1 [-b; b_eq])[0:n] from the forward result Y (tiles X) and the
// reciprocal pivots (left in LDS by tile_factor_forward).  rho G goes through LDS (over the dead matrix
// image) to Mrow = row `grow` of rho G; the upper tiles of W = I - rho G (identity-padded to whole tiles)
// stay in registers for the presolve (tile_factor_store).  Returns true if one of the bordered pivots
// (Shape::kBorder) was zero or not finite.
template <class SH>
__device__ __forceinline__ bool gram_output(lds_f64 *M, lds_f64 *vbuf, const int lane, const double rho, const int grow,
                                            const d4 (&X)[(SH::N1 + 15) / 16][kWave / 16],
                                            double (&Mrow)[SH::n],
                                            double &xc, double &gdiag, long long *fine = nullptr) {
  constexpr int N = SH::N1, n = SH::n, NRB = (N + 15) / 16, NW = (n + 15) / 16, NB = SH::kBorder;
  const int col = lane & 15, grp = lane >> 4;
#ifdef FCCQP_FINE_STAMPS
  const long long tg_ = wall_clock64();
#endif
  // ---- Gram product P = Y^T D^-1 Y with Y = L^-1 [E | r | B] (the forward result) ----------------------
  // (K1^-1)[0:n, 0:n] = E^T L^-T D^-1 L^-1 E (K1: the rows the tiles eliminated): instead of a backward
  // sweep with L^T -- MFMAs chained through the row blocks -- the product of the forward result with
  // itself: register r of tile Y[K][cb] is at once the B operand (B[k][n] = Y[4r+k][n]) and, for the
  // transposed factor, the A operand (A[i][k] = Y[4r+k][i]) of k-step r, so all operands come out of
  // the accumulators, the tile products are independent of each other, only the upper tiles (J >= I)
  // are computed and Y's zero tiles (rows above the unit entries) are skipped.  Column n of the product
  // is K1^-1 r, columns n+1 .. n+kBorder are K1^-1 B.
  constexpr int cbd = n / 16;                            // the column block that holds the dense columns
  constexpr int NI = (NB > 0) ? cbd + 1 : NW;            // row blocks of the product that are needed
  auto kmin = [](int cb) { return (cb == cbd) ? 0 : cb; };   // first row block in which Y[.][cb] is not all zero
  // 1/d per tile row, left behind the rho G image by tile_factor_forward (kept there rather than in 56
  // registers across the factorisation)
  double rd[NRB][4];
#pragma unroll
  for (int K = 0; K < NRB; ++K)
#pragma unroll
    for (int r = 0; r < 4; ++r) rd[K][r] = M[SH::kRcpOffset + 16 * K + grp + 4 * r];
  lds_fence();
  d4 Gt[NI][cbd + 1];
#pragma unroll
  for (int I = 0; I < NI; ++I) {
#pragma unroll
    for (int J = I; J <= cbd; ++J) Gt[I][J] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int K = 0; K < NRB; ++K) {
      if (K >= kmin(I)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (4 * (4 * K + r) < N) {
            const double aop = X[K][I][r] * rd[K][r];
#pragma unroll
            for (int J = I; J <= cbd; ++J)
              if (K >= kmin(J)) Gt[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, X[K][J][r], Gt[I][J], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- the bordered rows: K_rho = [[K1, B],[B^T, 0]] (fused_kernel keeps them in the zero block), so
  //   S = -B^T K1^-1 B (kBorder x kBorder) carries the last kBorder pivots of the same elimination order,
  //   (K_rho^-1)[0:n,0:n] = (K1^-1)[0:n,0:n] + Z S^-1 Z^T with Z = (K1^-1 B)[0:n],
  //   x_c = u - Z x2,  u = (K1^-1 r1)[0:n],  x2 = S^-1 (r2 - B^T K1^-1 r1).
  // Everything on the right is in columns n .. n+kBorder of the product: they go through LDS so that each
  // lane can pick up the entries of its own rows and column.
  bool bad = false;
  double corr_r[NI][4][NB > 0 ? NB : 1];   // (Z S^-1)[row][b] for this lane's rows of row block I
  double x2[NB > 0 ? NB : 1];
  lds_f64 *const Eb = M + SH::kBorderOffset;     // Eb[row * (NB + 1) + k] = P[row][n + k]
  if constexpr (NB > 0) {
    // (one lane predicate around all the stores -- the columns n .. n + NB of the product -- and every row of the
    // NI row blocks stored, the ones past n + NB into padding: a predicate per store costs an EXEC round trip each)
    static_assert(SH::kBorderDoubles >= 16 * NI * (NB + 1), "Eb holds whole row blocks");
    {
      const int cc = 16 * cbd + col;
      if (cc >= n && cc <= n + NB) {
#pragma unroll
        for (int I = 0; I < NI; ++I)
#pragma unroll
          for (int r = 0; r < 4; ++r) Eb[(16 * I + grp + 4 * r) * (NB + 1) + (cc - n)] = Gt[I][cbd][r];
      }
    }
    lds_fence();
    // S = -P[n+1.., n+1..] (symmetric; lower part from the upper tile), its LDL^T with the pivot rules of tile_ldl,
    // S^-1 and x2 -- a handful of scalar operations, done by every lane
    double Sm[NB][NB], Sinv[NB][NB], tvec[NB];
#pragma unroll
    for (int aa = 0; aa < NB; ++aa) {
      tvec[aa] = vbuf[N + aa] - Eb[n * (NB + 1) + 1 + aa];          // r2 - B^T K1^-1 r1  (P[n][n+1+a], upper part)
#pragma unroll
      for (int bb = 0; bb < NB; ++bb) {
        const int lo = (aa < bb) ? aa : bb, hi = (aa < bb) ? bb : aa;
        Sm[aa][bb] = -Eb[(n + 1 + lo) * (NB + 1) + 1 + hi];
      }
    }
    // in place, lower triangle only: unit lower L below the diagonal, d on it
    double dinv[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const double d = Sm[k][k];
      bad |= __builtin_amdgcn_class(d, 0x3 | 0x4 | 0x20 | 0x40 | 0x200);
      dinv[k] = pivot_inverse(d);
      double lk[NB];
#pragma unroll
      for (int i = k + 1; i < NB; ++i) lk[i] = pivot_quotient(Sm[i][k], d, dinv[k]);
#pragma unroll
      for (int i = k + 1; i < NB; ++i)
#pragma unroll
        for (int j = k + 1; j <= i; ++j) Sm[i][j] = fma(-lk[i], Sm[j][k], Sm[i][j]);   // (column k still unscaled)
#pragma unroll
      for (int i = k + 1; i < NB; ++i) Sm[i][k] = lk[i];
    }
    // S^-1 = L^-T D^-1 L^-1, column by column (unit vectors through the two sweeps)
#pragma unroll
    for (int cI = 0; cI <= NB; ++cI) {               // cI == NB: the right-hand side tvec -> x2
      double y[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) y[i] = (cI == NB) ? tvec[i] : ((i == cI) ? 1.0 : 0.0);
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int k = 0; k < i; ++k) y[i] -= Sm[i][k] * y[k];
#pragma unroll
      for (int i = 0; i < NB; ++i) y[i] *= dinv[i];
#pragma unroll
      for (int i = NB - 1; i >= 0; --i)
#pragma unroll
        for (int k = i + 1; k < NB; ++k) y[i] -= Sm[k][i] * y[k];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if (cI == NB) x2[i] = y[i];
        else Sinv[i][cI] = y[i];
      }
    }
    constexpr int kDummyStride = 16 * NI + 1;          // (odd: the sixteen columns spread over the LDS banks)
    static_assert(kDummyStride * 15 + 16 * NI <= SH::kRcpOffset, "the dummy slots lie in the dead image / exchange area");
    lds_f64 *const xc_base = (16 * cbd + col == n) ? vbuf : M + kDummyStride * col;    // (a slot of its own for every lane)
#pragma unroll
    for (int I = 0; I < NI; ++I) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + grp + 4 * r;
        double zr[NB];
#pragma unroll
        for (int aa = 0; aa < NB; ++aa) zr[aa] = Eb[row * (NB + 1) + 1 + aa];
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
          double acc = 0.0;
#pragma unroll
          for (int aa = 0; aa < NB; ++aa) acc += zr[aa] * Sinv[aa][bb];
          corr_r[I][r][bb] = acc;
        }
        // x_c (column n of the product, corrected): every lane computes it, the lanes that hold column n store it --
        // the others to the dead head of the work area, which the rho G image overwrites next (one select of the
        // base address for all sixteen stores instead of an EXEC round trip around each)
        double v = Gt[I][cbd][r];
#pragma unroll
        for (int aa = 0; aa < NB; ++aa) v -= zr[aa] * x2[aa];
        if (16 * I + 4 * r + 3 < n) xc_base[row] = v;                     // (whole group of four rows below n)
        else if (16 * I + 4 * r < n) { if (row < n) xc_base[row] = v; }
      }
    }
  }
  lds_fence();
#pragma unroll
  for (int I = 0; I < NW; ++I) {
#pragma unroll
    for (int J = I; J <= cbd; ++J) {
      double zc[NB > 0 ? NB : 1];
      if constexpr (NB > 0) {
        const int c = 16 * J + col;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) zc[bb] = Eb[((c < n) ? c : 0) * (NB + 1) + 1 + bb];
      }
      // (a tile that lies wholly inside the n x n block needs no masks: I and J are constants here; an edge tile
      // computes its four values first and stores them under ONE column predicate, the row condition decided at
      // compile time wherever a whole group of four rows lies on one side of n)
      const bool interior = (16 * I + 15 < n) && (16 * J + 15 < n);
      const int c = 16 * J + col;
      double g4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + grp + 4 * r;                        // Gt[I][J][r] = (K1^-1)[row][c]
        double gv = Gt[I][J][r];
        if constexpr (NB > 0) {
#pragma unroll
          for (int bb = 0; bb < NB; ++bb) gv = fma(corr_r[I][r][bb], zc[bb], gv);
        }
        g4[r] = rho * gv;
        if (NB == 0 && J == cbd && c == n && row < n) vbuf[row] = Gt[I][J][r];     // x_c
      }
      if (interior || c < n) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * I + grp + 4 * r;
          const bool rows_in = interior || (16 * I + 4 * r + 3 < n), rows_out = !interior && (16 * I + 4 * r >= n);
          if (rows_in || (!rows_out && row < n)) {
            M[row * SH::kImgStride + c] = g4[r];
            if (J != I) M[c * SH::kImgStride + row] = g4[r];         // the mirror tile
          }
        }
      }
    }
  }
  lds_fence();
#ifdef FCCQP_FINE_STAMPS
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (fine) fine[2] += wall_clock64() - tg_;
#endif
  // this lane takes row `grow` of rho G and entry `grow` of x_c (the ADMM loop's lane <-> variable map,
  // Shape::lane_variable; -1: none)
  const bool isvar = grow >= 0;
  const int lr = isvar ? grow : 0;
  xc = isvar ? vbuf[lr] : 0.0;
  gdiag = isvar ? M[lr * SH::kImgStride + lr] : 0.0;       // (rho G)[var][var]: the diagonal of W = I - rho G, for the presolve
  // (a lane without a variable reads a row of zeros laid over the reciprocal pivots, which are in registers by now:
  // one select of the row address instead of two per entry of the row)
  static_assert(SH::KKT::kRowsPadded >= n, "room for a zero row behind the image");
  if (lane < n) M[SH::kRcpOffset + lane] = 0.0;
  lds_fence();
  const lds_f64 *const rowp = isvar ? M + lr * SH::kImgStride : M + SH::kRcpOffset;
  if constexpr (n % 2 == 0 && SH::kRcpOffset % 2 == 0) {
    const lds_d2 *g2 = reinterpret_cast<const lds_d2 *>(rowp);
#pragma unroll
    for (int jp = 0; jp < n / 2; ++jp) {
      const d2 g = g2[jp];
      Mrow[2 * jp] = g.x;
      Mrow[2 * jp + 1] = g.y;
    }
  } else {
#pragma unroll
    for (int jj = 0; jj < n; ++jj) Mrow[jj] = rowp[jj];
  }
  lds_fence();
  return bad;
}

// ---- K_rho = U^T D U by 16 x 16 tiles held in registers, with the forward solve fused in ---------------
// Right-looking, upper-triangular form.  The tiles C[I][J] (J >= I) of the upper triangle and the
// right-hand-side tiles X[I][cb] of [e_0 .. e_{n-1} | -b; b_eq] live in MFMA accumulator layout (col =
// lane & 15, row = (lane >> 4) + 4 r) for the whole factorisation.  For row panel p:
//   (1) the diagonal tile C_pp = L D L^T (L = U_pp^T) is factorised and L^-1 formed on the VALU
//       (tile_ldl below: a trip through LDS to a column-per-lane layout and back);
//   (2) V_pj = L^-1 C_pj (= D U_pj) for the tiles right of it and Y_p = L^-1 X_p:  MFMAs whose A operand is
//       the inverse tile and whose B operand IS register r of the tile being solved;
//   (3) C_ij -= V_pi^T D^-1 V_pj and X_i -= V_pi^T D^-1 Y_p for the rows below:  both contract over the ROWS
//       of panel-p tiles, so A (= register r of V_pi: A[i][k] = V_pi[4r+k][i]) and B (= register r of
//       U_pj / Y_p) come straight out of the accumulators -- no operand is ever read from LDS, no tile is
//       ever written back to it, and the panel's tiles are dead once step (3) is done (the factor itself
//       is not kept: G = Y^T D^-1 Y needs only Y and D).
// Against the left-looking in-LDS panel factorisation of rounds 1-2a (and its separate forward sweep) this
// removes the panel loads / stores (10 us per QP), the per-k-step operand fetches and the 98-row-high
// in-panel updates with their v_readlane pairs (the serial part shrinks to the 16 x 16 diagonal tiles).
// Same elimination order (natural, unpivoted) and the same pivot rules.
template <int K>
__device__ __forceinline__ double row_bcast(double v) {   // lane K of this lane's 16-lane row
  int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + K, 0xf, 0xf, true);
  int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + K, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// t += t(lane K of the row) * f   /   y += l(lane K of the row) * f.   volatile: these keep their program
// order, which is what spaces a DPP read of a VGPR from the VALU write of it (2 wait states; the
// hazard recogniser does not look inside asm statements) -- see the s_nop notes in tile_ldl.
template <int K>
__device__ __forceinline__ void dpp_self_fma(double &t, const double f) {
  asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(f), "n"(K));
}
template <int K>
__device__ __forceinline__ void dpp_fma(double &y, const double l, const double f) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(y) : "v"(l), "v"(f), "n"(K));
}
// Lane masks that depend only on the column c = lane & 15 are compile-time constants (the tile is replicated in the
// four 16-lane rows of the wave): an SGPR-pair literal instead of a v_cmp per use, whose results the compiler used
// to hoist out of the sixteen steps and then spill to VGPR lanes.
__host__ __device__ constexpr unsigned long long rows4(const unsigned m16) { return (m16 & 0xffffull) * 0x0001000100010001ull; }
template <int K> struct ColMask {
  static constexpr unsigned long long eq = rows4(1u << K), gt = rows4(0xffffu << (K + 1));
};
// lane K of this lane's 16-lane row, one instruction; volatile: it DPP-reads a register a volatile FMA has just written
// and must keep its place behind it (2 wait states, see tile_ldl)
template <int K>
__device__ __forceinline__ double row_bcast_ordered(const double &v) {
  double d;
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v), "n"(K));
  return d;
}
// -(a / d) with the rounding of pivot_quotient (negations are exact and ride on the source modifiers)
__device__ __forceinline__ double neg_pivot_quotient(const double a, const double d, const double r) {
  const double nq = -a * r;
  return fma(-fma(d, nq, a), r, nq);
}
// rows I = LO .. HI of column K of L^-1's recursion: Y_I += S_IK (lane K) * f, f = -Y_K / d_K
template <int K, int LO, int HI>
__device__ __forceinline__ void tile_inverse_rows(double (&Y)[16], const double (&T)[16], const double f) {
  if constexpr (LO <= HI) {
    dpp_fma<K>(Y[LO], T[LO], f);
    tile_inverse_rows<K, LO + 1, HI>(Y, T, f);
  }
}
// Elimination step K of the diagonal tile, d = d_K in every lane.  Column c (this lane) takes
// S_ic -= S_iK (S_Kc / d_K), i > K, for c > K (lanes c <= K add zero), and -- WITH_INVERSE -- the rows below K of L^-1
// take column K's contribution in the same step: y_i -= l_iK y_K with l_iK d_K = S_iK read from lane K, whose column is
// final since step K - 1, and the 1/d_K folded into the other factor (no separate scaling pass over the tile).
// Order inside a step (the FMAs are volatile asm and keep it): the update of row K + 1 first, then the L^-1 rows that
// do not touch it, then the broadcast of the next pivot out of it (VALU write -> DPP read: two wait states, s_nop
// where a step has no such work left), L^-1's row K + 1 and the remaining rows.  The pivot chain
// (rcp, refinement, quotient, selects) stays in one piece in front of the DPP block: spreading its links between the
// FMAs to cover their latency was measured 1.2 % SLOWER per solve -- the wave is issue-bound and alternating DPP and
// plain fp64 operations costs more than the dependent issue does.
template <int K, int LO, int HI>
__device__ __forceinline__ void tile_step_rows(double (&T)[16], const double negm) {
  if constexpr (LO <= HI) {
    dpp_self_fma<K>(T[LO], negm);
    tile_step_rows<K, LO + 1, HI>(T, negm);
  }
}
// (NV valid rows: the padding rows / columns NV .. 15 of a factorisation's last tile are the identity -- no step for
// them, and no update of them: their entries in the columns being eliminated are zero)
// KEEP_L = false (the caller does not read T afterwards): the multiplier is NOT masked to the columns right of K.
// Lane c <= K then "updates" its own, finished column with a multiplier of its own -- lane K's is -S_KK / d_K = -1,
// which zeroes that column -- and nobody minds: L^-1 has read lane K's column by then (its row K + 1 moves in front
// of the first update for that), the pivots come from columns right of K only, and two v_cndmask + two s_mov per
// step are gone.
// (eqm / gtm: the lane masks "column == K" and "column > K" of the step, carried from step to step in SGPR pairs --
// one s_lshl_b64 / s_andn2_b64 each instead of two s_mov_b32 literals per mask and step; tile_ldl hides the first
// pair from constant folding)
template <int K, bool WITH_INVERSE, int NV, bool KEEP_L, bool RULE>
__device__ __forceinline__ void tile_steps(double (&T)[16], double (&Y)[16], const double d, double &rmine,
                                           const unsigned long long eqm, const unsigned long long gtm) {
  const double r = RULE ? pivot_inverse(d) : fast_rcp(d);
  rmine = select_lanes(eqm, r, rmine);
  if constexpr (K + 1 < NV) {
    const double nq = neg_pivot_quotient(T[K], d, r);
    const double negm = KEEP_L ? select_lanes_or_zero(gtm, nq) : nq;
    double f = 0.0;
    if constexpr (WITH_INVERSE) {
      // the unit entry of row K of L^-1 goes in now, with the mask of this step (Y starts as all zeros: lane K of
      // Y[K] has only ever accumulated multiples of the zeros right of the diagonal in the rows above)
      {
        int hi;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"(__double2hiint(Y[K])), "v"(__double2hiint(1.0)), "s"(eqm));
        Y[K] = __hiloint2double(hi, __double2loint(Y[K]));
      }
      f = -r * Y[K];
      if constexpr (!KEEP_L) dpp_fma<K>(Y[K + 1], T[K + 1], f);
    }
    dpp_self_fma<K>(T[K + 1], negm);
    if constexpr (WITH_INVERSE) tile_inverse_rows<K, K + 2, NV - 1>(Y, T, f);
    if constexpr (!WITH_INVERSE || K + 3 >= NV) asm volatile("s_nop 1" : "+v"(T[K + 1]));
    const double dn = row_bcast_ordered<K + 1>(T[K + 1]);
    if constexpr (WITH_INVERSE && KEEP_L) dpp_fma<K>(Y[K + 1], T[K + 1], f);
    tile_step_rows<K, K + 2, NV - 1>(T, negm);
    const unsigned long long eqn = eqm << 1;
    tile_steps<K + 1, WITH_INVERSE, NV, KEEP_L, RULE>(T, Y, dn, rmine, eqn, KEEP_L ? (gtm & ~eqn) : 0ull);
  }
}
// Diagonal tile, column-per-lane (every 16-lane row of the wave holds a copy and does the same work):
// in  T[i] = S[i][c], the symmetric 16 x 16 block;
// out rmine = 1/d_c (pseudo-inverse rule: 0 for |d_c| <= DBL_MIN), KEEP_L: T[i] = l_ic d_c for i > c (the column of L
//     still times its pivot: a caller that stores L multiplies by rmine; without KEEP_L T is scratch), Y[i] = (L^-1)[i][c] for i > c (Y[c] = 1,
//     Y[i < c] = 0).  MAX_RCP: the largest reciprocal pivot still accepted (a definite matrix whose pivots
//     must stay above 1 / MAX_RCP), or 0: then only a dropped or non-finite pivot (rmine 0 or NaN: d was
//     zero, subnormal, infinite or NaN) is reported.  Padding rows are identity.
// (WITH_INVERSE = false: the last tile of a factorisation without right-hand sides -- nothing is solved with it)
// (The tile arrives through LDS -- `colp`: this lane's column, written in accumulator layout by the caller -- and is run
// OPTIMISTICALLY first: without the |d| <= DBL_MIN test and its two selects in each of the sixteen steps.  A pivot that
// the rule would have dropped shows in its reciprocal (|1/d| >= 2^1022, infinite or NaN, which is also what a zero
// or subnormal d gives); then -- rank-deficient equality rows, practically never otherwise -- the column is read
// again and the tile redone with the rule in every step.  Conservative: a pivot within an ulp of DBL_MIN may take
// the second pass and come out the same.)
template <int NVALID, bool WITH_INVERSE, bool KEEP_L, bool RULE>
__device__ __forceinline__ void tile_run(double (&T)[16], double (&Y)[16], double &rmine, const lds_d2 *colp) {
#pragma unroll
  for (int ip = 0; ip < 8; ++ip) {
    const d2 t = colp[ip];
    T[2 * ip] = t.x;
    T[2 * ip + 1] = t.y;
  }
  lds_fence();
  rmine = 1.0;
  if constexpr (WITH_INVERSE) {
    // Y = 0; the unit diagonal arrives row by row in the steps (rows NVALID - 1 .. 15, which no step reaches as
    // "row K", here)
#pragma unroll
    for (int i = 0; i < 16; ++i) Y[i] = 0.0;
    static_for<(NVALID > 0 ? NVALID - 1 : 0), 16>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      int hi;
      asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(hi) : "v"(__double2hiint(1.0)), "s"(ColMask<i>::eq));
      Y[i] = __hiloint2double(hi, 0);
    });
  }
  const double d0 = row_bcast<0>(T[0]);
  unsigned long long eq0 = ColMask<0>::eq, gt0 = ColMask<0>::gt;
  asm volatile("" : "+s"(eq0), "+s"(gt0));
  tile_steps<0, WITH_INVERSE, NVALID, KEEP_L, RULE>(T, Y, d0, rmine, eq0, gt0);
}
template <int NVALID, bool WITH_INVERSE = true, bool KEEP_L = false>
__device__ __forceinline__ bool tile_ldl(double (&T)[16], double (&Y)[16], double &rmine, const int c, const lds_d2 *colp,
                                         const double max_rcp = 0.0) {
  tile_run<NVALID, WITH_INVERSE, KEEP_L, false>(T, Y, rmine, colp);
  if (__builtin_expect(__any((c < NVALID) && !(fabs(rmine) < 0x1p1022)), 0))
    tile_run<NVALID, WITH_INVERSE, KEEP_L, true>(T, Y, rmine, colp);
  // judged once per tile: lane c holds 1/d_c
  const bool lane_bad = (max_rcp > 0.0) ? !(rmine > 0.0 && rmine < max_rcp) : __builtin_amdgcn_class(rmine, 0x3 | 0x20 | 0x40);
  return __any((c < NVALID) && lane_bad);
}

// P0: the upper triangle of K_rho = [[Q + rho I, A^T],[A, 0]] (fcc_qp.cpp:146-150,62-64) from global memory
// straight into accumulator-layout tiles: C[I][J][r] = K_rho[i][j], i = 16 I + (lane >> 4) + 4 r, j = 16 J + (lane & 15).
// Only Q[a][b] with a >= b is read (like Eigen's Lower LDLT), so element (i, j), i <= j, comes from row j of Q
// (or row j - n of A_eq): the 16 lanes of a lane-row read 16 rows, the 4 lane-rows 4 consecutive entries,
// and the four loads r = 0 .. 3 of a tile finish each 128-byte piece of a row.  Rows >= n are the zero
// block; rows / columns >= N pad the last tile with the identity.  All loads are issued before the first
// value is used (issue_kkt_loads, then convert_kkt_tiles): the first panel's diagonal tile is factorised
// while the rest is still in flight.
template <class SH, class TIN, int I0, int I1>
__device__ __forceinline__ void issue_kkt_loads(const TIN *__restrict__ Q, const TIN *__restrict__ A, const int lane,
                                                TIN (&raw)[(SH::N1 + 15) / 16][(SH::N1 + 15) / 16][4]) {
  constexpr int N = SH::N1, n = SH::n, NRB = (N + 15) / 16;   // (the rows the tiles eliminate: Shape::kBorder)
  const int col = lane & 15, grp = lane >> 4;
#pragma unroll
  for (int I = I0; I < I1; ++I) {
    if (16 * I < n) {
#pragma unroll
      for (int J = I; J < NRB; ++J) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * I + grp + 4 * r, j = 16 * J + col;
          const int lo = (I == J && j < i) ? j : i, hi = (I == J && j < i) ? i : j;
          const int loc = (lo < n) ? lo : n - 1;                  // clamped: in-bounds reads for masked entries
          const int hic = (hi < N) ? hi : N - 1;
          if (16 * J + 15 < n) {
            raw[I][J][r] = Q[hic * n + loc];
          } else if (16 * J >= n) {
            raw[I][J][r] = A[(hic - n) * n + loc];
          } else {
            const TIN *src = (hic < n) ? Q + hic * n : A + (hic - n) * n;
            raw[I][J][r] = src[loc];
          }
        }
      }
    }
  }
}
template <class SH, class TIN>
__device__ __forceinline__ void convert_kkt_tiles(const TIN (&raw)[(SH::N1 + 15) / 16][(SH::N1 + 15) / 16][4], const double rho,
                                                  const int lane, d4 (&C)[(SH::N1 + 15) / 16][(SH::N1 + 15) / 16]) {
  constexpr int N = SH::N1, n = SH::n, NRB = (N + 15) / 16;
  const int col = lane & 15, grp = lane >> 4;
#pragma unroll
  for (int I = 0; I < NRB; ++I) {
#pragma unroll
    for (int J = I; J < NRB; ++J) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + grp + 4 * r, j = 16 * J + col;
        const int lo = (I == J && j < i) ? j : i, hi = (I == J && j < i) ? i : j;
        double v = (i == j && i >= N) ? 1.0 : 0.0;
        if (16 * I < n) {
          const double q = (double)raw[I][J][r] + ((I == J && i == j) ? rho : 0.0);
          v = (lo < n && hi < N) ? q : v;
        }
        C[I][J][r] = v;
      }
    }
  }
}

template <class SH>
__device__ __forceinline__ bool tile_factor_forward(lds_f64 *M, lds_f64 *vbuf, const int lane,
                                                    d4 (&C)[(SH::N1 + 15) / 16][(SH::N1 + 15) / 16],
                                                    d4 (&X)[(SH::N1 + 15) / 16][kWave / 16], long long *fine = nullptr) {
  constexpr int N = SH::N1, n = SH::n, NRB = (N + 15) / 16, NCB = kWave / 16;   // N: the rows eliminated by tiles
  static_assert(n < kWave, "n + 1 right-hand sides in 64 columns");

  const int col = lane & 15, grp = lane >> 4;
  // tile X[K][cb] is still all zero while every unit entry of column block cb lies below row block K; the
  // block that holds the dense column [-b; b_eq] is the exception
  auto zero_tile = [](int K, int cb) { return K < cb && cb != n / 16; };
  // right-hand sides: column c < n is e_c, column n is [-b; b_eq] (staged in vbuf), the rest zero.  A tile is
  // materialised when panel kfirst(cb) first touches its column block (28 more live tiles otherwise)
  constexpr int cbd = n / 16;
  auto kfirst = [](int cb) { return (cb == cbd) ? 0 : cb; };
  auto x_initial = [&](const int I, const int cb) {
    d4 t;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * I + grp + 4 * r, c = 16 * cb + col;
      // (I and cb are constants once the callers' loops are unrolled: a tile off the diagonal that is not in
      // the dense column block is the literal zero tile -- no registers, srcC = 0 in its first MFMA)
      double v = 0.0;
      if (I == cb) v = (row == c && c < n) ? 1.0 : 0.0;
      if (cb == cbd) {
        v = (c == n) ? ((row < N) ? vbuf[row < N ? row : 0] : 0.0) : v;
        if constexpr (SH::kBorder > 0) {   // columns n+1 .. n+kBorder: the border columns of K_rho (staged by the caller)
          const int bc = (c > n && c <= n + SH::kBorder) ? c - n - 1 : 0;
          const double bv = M[SH::kBorderOffset + bc * SH::kTileRows + row];
          v = (c > n && c <= n + SH::kBorder) ? bv : v;
        }
      }
      t[r] = v;
    }
    return t;
  };
  constexpr int TS = SH::kTileStride;
  lds_f64 *const Sblk = M;                        // exchange scratch: diagonal tile, column-major 16 x 16 (column stride TS)
  lds_f64 *const Sinv = M + SH::kTileLen;         // its L^-1, column-major
  lds_f64 *const Srcp = M + 2 * SH::kTileLen;     // 1 / d of the tile's rows
  bool bad = false;
  static_for<0, NRB>([&](auto pc) {
    constexpr int P = decltype(pc)::value;
    constexpr int kValid = (N - 16 * P < 16) ? N - 16 * P : 16;
    // (1) diagonal tile
#pragma unroll
    for (int r = 0; r < 4; ++r) Sblk[TS * col + grp + 4 * r] = C[P][P][r];
    lds_fence();
    double Ainv[4], nrv[4];
    {
      double T[16], Y[16], rmine;
      const lds_d2 *colp = reinterpret_cast<const lds_d2 *>(Sblk + TS * col);
      FCCQP_FINE(0, bad |= tile_ldl<kValid>(T, Y, rmine, col, colp));
      if (lane < 16) {
        lds_d2 *outp = reinterpret_cast<lds_d2 *>(Sinv + TS * col);
#pragma unroll
        for (int ip = 0; ip < 8; ++ip) outp[ip] = d2{Y[2 * ip], Y[2 * ip + 1]};
        Srcp[col] = rmine;
        M[SH::kRcpOffset + 16 * P + col] = rmine;
      }
      lds_fence();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // A[i][k] = (L^-1)[i][4r + k], strictly below the diagonal (the unit diagonal is the accumulator's start)
      const double li = Sinv[TS * (4 * r + grp) + col];
      Ainv[r] = (col > 4 * r + grp) ? li : 0.0;
      nrv[r] = -Srcp[grp + 4 * r];
    }
    lds_fence();
    // (2) row panel: V_pj = L^-1 C_pj (in place); Y_p = L^-1 X_p
#pragma unroll
    for (int J = P + 1; J < NRB; ++J) {
      d4 t = C[P][J];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * P + 4 * r < N) t = __builtin_amdgcn_mfma_f64_16x16x4f64(Ainv[r], C[P][J][r], t, 0, 0, 0);
      C[P][J] = t;
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      if (!zero_tile(P, cb)) {
        if (cb == P && cb != cbd) {
          // the tile of the unit columns 16 P .. 16 P + 15 is still the identity when its panel comes up (the
          // panels above it only had zero tiles in this column block): L^-1 I is the tile tile_ldl has just
          // left in the exchange area -- four LDS reads instead of four MFMAs
#pragma unroll
          for (int r = 0; r < 4; ++r) X[P][cb][r] = Sinv[TS * col + grp + 4 * r];
        } else {
          if (P == kfirst(cb)) X[P][cb] = x_initial(P, cb);
          d4 t = X[P][cb];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * P + 4 * r < N) t = __builtin_amdgcn_mfma_f64_16x16x4f64(Ainv[r], X[P][cb][r], t, 0, 0, 0);
          X[P][cb] = t;
        }
      }
    }
    // (3) the rows below
#pragma unroll
    for (int I = P + 1; I < NRB; ++I) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
        if (!zero_tile(P, cb) && P == kfirst(cb)) X[I][cb] = x_initial(I, cb);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (16 * P + 4 * r < N) {
          const double aop = C[P][I][r] * nrv[r];        // A[i][k] = -(V_pi[4r + k][i] / d_{4r + k})
#pragma unroll
          for (int J = I; J < NRB; ++J)
            C[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, C[P][J][r], C[I][J], 0, 0, 0);
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
            if (!zero_tile(P, cb)) X[I][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, X[P][cb][r], X[I][cb], 0, 0, 0);
        }
      }
    }
  });
  return bad;
}

// upper tiles of W = I - rho G in accumulator layout from the rho G image gram_output left in LDS (row stride
// Shape::kImgStride), rows / columns past n the identity
template <class SH>
__device__ __forceinline__ void w_tiles_from_image(const lds_f64 *M, const int lane, d4 (&Wt)[(SH::n + 15) / 16][(SH::n + 15) / 16]) {
  constexpr int n = SH::n, NW = (n + 15) / 16;
  const int col = lane & 15, grp = lane >> 4;
#pragma unroll
  for (int I = 0; I < NW; ++I) {
#pragma unroll
    for (int J = I; J < NW; ++J) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + grp + 4 * r, c = 16 * J + col;
        const bool inside = row < n && c < n;
        const double g = M[(inside ? row : 0) * SH::kImgStride + (inside ? c : 0)];
        const double unit = (I == J && row == c) ? 1.0 : 0.0;
        Wt[I][J][r] = inside ? unit - g : unit;
      }
    }
  }
  lds_fence();
}

// W = U^T D U by register tiles (the presolve, see fused_kernel), same scheme as tile_factor_forward without
// right-hand sides; the factor is WRITTEN to LDS in the packed layout solve_lds_one_slot reads (L = U^T below the
// diagonal, 1/d in the diagonal slots).  Returns true if a pivot was not above min_pivot (or not finite): W is
// positive definite unless K is singular to working precision.
// The exchange scratch sits behind the packed n x n image.
template <class SH>
__device__ __forceinline__ bool tile_factor_store(lds_f64 *M, const int lane, const double min_pivot,
                                                  d4 (&C)[(SH::n + 15) / 16][(SH::n + 15) / 16], long long *fine = nullptr) {
  using SW = typename SH::Reduced;
  constexpr int n = SH::n, NW = (n + 15) / 16;
  constexpr int kOff = (SW::kPacked + 1) & ~1;
  static_assert(kOff + SH::kScratch <= SH::kMat, "exchange scratch behind the packed W");
  const int col = lane & 15, grp = lane >> 4;
  constexpr int TS = SH::kTileStride;
  lds_f64 *const Sblk = M + kOff;
  lds_f64 *const Sinv = Sblk + SH::kTileLen;
  lds_f64 *const Srcp = Sblk + 2 * SH::kTileLen;
  if constexpr (SW::kZeroFilled) {
    // zeros on and above the diagonal of the square: the whole of it first (the rho G image it lies over is in
    // registers by now), L then goes on top
    static_assert(SW::kDinv % 2 == 0 && SW::kDinv <= kOff, "the square ends in front of the exchange scratch");
    lds_d2 *const z2 = reinterpret_cast<lds_d2 *>(M);
#pragma unroll
    for (int k = 0; k < (SW::kDinv / 2 + kWave - 1) / kWave; ++k)
      if ((k + 1) * kWave <= SW::kDinv / 2 || k * kWave + lane < SW::kDinv / 2) z2[k * kWave + lane] = d2{0.0, 0.0};
    lds_fence();
  }
  bool bad = false;
  static_for<0, NW>([&](auto pc) {
    constexpr int P = decltype(pc)::value;
    constexpr int kValid = (n - 16 * P < 16) ? n - 16 * P : 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) Sblk[TS * col + grp + 4 * r] = C[P][P][r];
    lds_fence();
    double Ainv[4], rv[4];
    {
      double T[16], Y[16], rmine;
      const lds_d2 *colp = reinterpret_cast<const lds_d2 *>(Sblk + TS * col);
      constexpr bool kLast = (P + 1 == NW);
      FCCQP_FINE(0, (bad |= tile_ldl<kValid, !kLast, true>(T, Y, rmine, col, colp, 1.0 / min_pivot)));
      if (lane < 16) {
        if constexpr (!kLast) {
          lds_d2 *outp = reinterpret_cast<lds_d2 *>(Sinv + TS * col);
#pragma unroll
          for (int ip = 0; ip < 8; ++ip) outp[ip] = d2{Y[2 * ip], Y[2 * ip + 1]};
          Srcp[col] = rmine;
        }
        // column 16 P + col of L (tile_ldl left T = L d), its 1/d
        if (col < kValid) {
          const int jcol = 16 * P + col, cb = SW::cbase(jcol);
          M[SW::kZeroFilled ? SW::kDinv + jcol : cb + jcol] = rmine;
          // (entries above the diagonal go to a per-lane dummy slot of the dead exchange tile: one select
          // of the address instead of an exec-mask round trip per store)
          // (the select written out as VOP3 on a ballot: the compiler's own `v_cmp -> VCC, s_nop 1, v_cndmask_b32_e32`
          // costs this lone wave 32 cycles per address, 5 + 5 this way; `colv`: hidden from CSE, or the fifteen masks
          // are hoisted out of the panels and spilled to VGPR lanes)
          int colv = col;
          asm volatile("" : "+v"(colv));
          const int below0 = cb + 16 * P, dummy = kOff + colv;
#pragma unroll
          for (int i = 1; i < kValid; ++i) {
            const int idx = select32_lanes(__builtin_amdgcn_ballot_w64(i > colv), below0 + i, dummy);
            M[idx] = T[i] * rmine;
          }
        }
      }
      lds_fence();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double li = Sinv[TS * (4 * r + grp) + col];
      Ainv[r] = (col > 4 * r + grp) ? li : 0.0;
      rv[r] = Srcp[grp + 4 * r];
    }
    lds_fence();
#pragma unroll
    for (int J = P + 1; J < NW; ++J) {
      d4 t = C[P][J];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * P + 4 * r < n) t = __builtin_amdgcn_mfma_f64_16x16x4f64(Ainv[r], C[P][J][r], t, 0, 0, 0);
      C[P][J] = t;
      // L[16 J + col][16 P + row] = U_pj[row][col] = V_pj[row][col] / d_row
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jrow = 16 * P + grp + 4 * r, irow = 16 * J + col;
        if (16 * J + 15 < n || irow < n) M[SW::cbase(jrow) + irow] = t[r] * rv[r];      // (jrow < n: P < J)
      }
    }
#pragma unroll
    for (int I = P + 1; I < NW; ++I) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double aop = C[P][I][r] * -rv[r];
#pragma unroll
        for (int J = I; J < NW; ++J)
          C[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, C[P][J][r], C[I][J], 0, 0, 0);
      }
    }
  });
  lds_fence();
  return bad;
}

// x := (L D L^T)^-1 x for N <= 64 (one row per lane), M as left by tile_factor_store.  Column-oriented forward
// and backward substitution; the only dependent chain is v_readlane(x_k) -> fma.  The L operands of the next
// U steps are fetched while the current U run (two register buffers): the single-buffered version waited
// for an LDS round trip in front of every group of four steps.
template <class SH>
__device__ __forceinline__ void solve_lds_one_slot(const lds_f64 *M, const int lane, double &x) {
  constexpr int N = SH::N, U = 8;
  constexpr bool kZ = SH::kZeroFilled;       // zeros on and above the diagonal: no "row below the pivot?" selects
  static_assert(N <= kWave, "one row per lane");
  const int row = lane;
  const int rc = (row < N) ? row : N - 1;        // clamped: in-bounds reads for idle lanes
  const int cbr = SH::cbase(rc);
  // ---- forward, unit lower: x_i -= L[i][k] x_k, i > k.  L[row][k] = M[cbase(k) + row] -------------------------
  {
    double cur[U], nxt[U];
    auto load_fwd = [&](double (&buf)[U], const int k0) {
#pragma unroll
      for (int u = 0; u < U; ++u) buf[u] = M[SH::cbase((k0 + u < N) ? k0 + u : N - 1) + rc];
    };
    load_fwd(cur, 0);
#pragma unroll
    for (int k0 = 0; k0 < N; k0 += U) {
      if (k0 + U < N) load_fwd(nxt, k0 + U);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + u;
        if (k < N - 1) {                           // (column N - 1 has nothing below its diagonal)
          const double a = kZ ? cur[u] : ((row > k) ? cur[u] : 0.0);
          const double xk = bcast_lane(x, k);
          x = fma(-a, xk, x);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
  }
  // ---- D^-1: 1/d with Eigen's pseudo-inverse rule folded in (the diagonal slot of the packed layout) ---------
  if constexpr (kZ) x *= M[SH::kDinv + rc];
  else x *= M[cbr + rc];
  // ---- backward with L^T: x_i -= L[k][i] x_k, i < k.  L[k][row] = M[cbase(row) + k]: this lane's own column ---
  {
    double cur[U], nxt[U];
    auto load_bwd = [&](double (&buf)[U], const int k0) {     // k = k0, k0 - 1, ...
#pragma unroll
      for (int u = 0; u < U; ++u) buf[u] = M[cbr + ((k0 - u >= 1) ? k0 - u : 1)];
    };
    load_bwd(cur, N - 1);
#pragma unroll
    for (int k0 = N - 1; k0 >= 1; k0 -= U) {
      if (k0 - U >= 1) load_bwd(nxt, k0 - U);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 - u;
        if (k >= 1) {
          const double a = kZ ? cur[u] : ((row < k) ? cur[u] : 0.0);
          const double xk = bcast_lane(x, k);
          x = fma(-a, xk, x);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
  }
}

// acc[i % kMvAcc] += w[16 q + i] * Mrow[16 q + i], i = 0 .. L-1: one 16-column block of the mat-vec of an
// ADMM iteration, the vector operand fetched by the FMA itself (DPP row_newbcast from lane i of this
// lane's 16-lane row of Wq, where Wq[lane] = w[16 q + (lane & 15)]).  ONE asm statement per block: the
// hazard recogniser counts an inline-asm statement as zero wait states, so with one statement per FMA
// it put an s_nop in front of every reuse of an accumulator whatever the distance (a dependent fp64 FMA
// needs none).  Unused operands of the shorter last block are legal.
// Accumulators of the mat-vec: the DPP FMA issues every ~8.2 cycles with a fresh matrix operand whether it depends on
// the previous one or not and whatever the number of accumulators (tools/vgpr_bank_probe.hip), so ONE accumulator
// costs nothing and saves the zeroing of the others and the three adds of the final sum.  (-DFCCQP_MV_FOUR_ACC: the
// four rotating accumulators of round 2.)
#ifdef FCCQP_MV_FOUR_ACC
constexpr bool kMvOneAcc = false;
#define FCCQP_A0 0
#define FCCQP_A1 1
#define FCCQP_A2 2
#define FCCQP_A3 3
#else
constexpr bool kMvOneAcc = true;
#define FCCQP_A0 0
#define FCCQP_A1 0
#define FCCQP_A2 0
#define FCCQP_A3 0
#endif
constexpr int kMvAcc = 4;
__device__ __forceinline__ double mv_total(const double (&acc)[kMvAcc]) {
  if constexpr (kMvOneAcc) return acc[0];
  else return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
#define FCCQP_DPPF_X(A, M, L) "v_fmac_f64_dpp %" #A ", %4, %" #M " row_newbcast:" #L " row_mask:0xf bank_mask:0xf\n\t"
#define FCCQP_DPPF(A, M, L) FCCQP_DPPF_X(A, M, L)
#define FCCQP_S1 FCCQP_DPPF(FCCQP_A0, 5, 0)
#define FCCQP_S2 FCCQP_S1 FCCQP_DPPF(FCCQP_A1, 6, 1)
#define FCCQP_S3 FCCQP_S2 FCCQP_DPPF(FCCQP_A2, 7, 2)
#define FCCQP_S4 FCCQP_S3 FCCQP_DPPF(FCCQP_A3, 8, 3)
#define FCCQP_S5 FCCQP_S4 FCCQP_DPPF(FCCQP_A0, 9, 4)
#define FCCQP_S6 FCCQP_S5 FCCQP_DPPF(FCCQP_A1, 10, 5)
#define FCCQP_S7 FCCQP_S6 FCCQP_DPPF(FCCQP_A2, 11, 6)
#define FCCQP_S8 FCCQP_S7 FCCQP_DPPF(FCCQP_A3, 12, 7)
#define FCCQP_S9 FCCQP_S8 FCCQP_DPPF(FCCQP_A0, 13, 8)
#define FCCQP_S10 FCCQP_S9 FCCQP_DPPF(FCCQP_A1, 14, 9)
#define FCCQP_S11 FCCQP_S10 FCCQP_DPPF(FCCQP_A2, 15, 10)
#define FCCQP_S12 FCCQP_S11 FCCQP_DPPF(FCCQP_A3, 16, 11)
#define FCCQP_S13 FCCQP_S12 FCCQP_DPPF(FCCQP_A0, 17, 12)
#define FCCQP_S14 FCCQP_S13 FCCQP_DPPF(FCCQP_A1, 18, 13)
#define FCCQP_S15 FCCQP_S14 FCCQP_DPPF(FCCQP_A2, 19, 14)
#define FCCQP_S16 FCCQP_S15 FCCQP_DPPF(FCCQP_A3, 20, 15)
// (with one accumulator the other three are not operands of the statement at all: as "+v" operands they would
// have to be materialised -- three v_mov_b64 per iteration -- although the instructions never name them; the
// operand numbers %1 .. %3 are kept by three dummy inputs that cost nothing)
#ifdef FCCQP_MV_FOUR_ACC
#define FCCQP_DPP_BLOCK(STR)                                                                          \
  asm(STR : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])                                    \
      : "v"(w), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), \
        "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]), "v"(m[12]), "v"(m[13]), "v"(m[14]), "v"(m[15]))
#else
#define FCCQP_DPP_BLOCK(STR)                                                                          \
  asm(STR : "+v"(acc[0])                                                                              \
      : "n"(0), "n"(0), "n"(0),                                                                       \
        "v"(w), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), \
        "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]), "v"(m[12]), "v"(m[13]), "v"(m[14]), "v"(m[15]))
#endif
template <int L>
__device__ __forceinline__ void dpp_block(double (&acc)[kMvAcc], const double w, const double (&m)[16]) {
  static_assert(L >= 1 && L <= 16, "one 16-lane row");
  if constexpr (L == 16) FCCQP_DPP_BLOCK(FCCQP_S16);
  else if constexpr (L == 15) FCCQP_DPP_BLOCK(FCCQP_S15);
  else if constexpr (L == 14) FCCQP_DPP_BLOCK(FCCQP_S14);
  else if constexpr (L == 13) FCCQP_DPP_BLOCK(FCCQP_S13);
  else if constexpr (L == 12) FCCQP_DPP_BLOCK(FCCQP_S12);
  else if constexpr (L == 11) FCCQP_DPP_BLOCK(FCCQP_S11);
  else if constexpr (L == 10) FCCQP_DPP_BLOCK(FCCQP_S10);
  else if constexpr (L == 9) FCCQP_DPP_BLOCK(FCCQP_S9);
  else if constexpr (L == 8) FCCQP_DPP_BLOCK(FCCQP_S8);
  else if constexpr (L == 7) FCCQP_DPP_BLOCK(FCCQP_S7);
  else if constexpr (L == 6) FCCQP_DPP_BLOCK(FCCQP_S6);
  else if constexpr (L == 5) FCCQP_DPP_BLOCK(FCCQP_S5);
  else if constexpr (L == 4) FCCQP_DPP_BLOCK(FCCQP_S4);
  else if constexpr (L == 3) FCCQP_DPP_BLOCK(FCCQP_S3);
  else if constexpr (L == 2) FCCQP_DPP_BLOCK(FCCQP_S2);
  else FCCQP_DPP_BLOCK(FCCQP_S1);
}
template <int Q, int NN, int NQ>
__device__ __forceinline__ void dpp_matvec(double (&acc)[kMvAcc], const double (&Wq)[NQ], const double (&Mrow)[NN]) {
  if constexpr (Q < NQ) {
    constexpr int L = (NN - 16 * Q < 16) ? NN - 16 * Q : 16;
    double m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = Mrow[(16 * Q + i < NN) ? 16 * Q + i : NN - 1];
    dpp_block<L>(acc, Wq[Q], m);
    dpp_matvec<Q + 1, NN, NQ>(acc, Wq, Mrow);
  }
}

// The same mat-vec with TWO 16-column blocks per asm statement (the one-wave kernel's ADMM loop): between two asm
// statements that touch the same accumulator the backend puts an `s_nop 0`, which costs this lone wave a whole issue slot
// like any other instruction -- four statements per iteration cost three of them, two statements cost one (the
// `s_waitcnt` for the second pair of w copies sits between the two).  Operands by name: m<B>_<K> = entry K of block B.
#define FCCQP_MVF(B, K) "v_fmac_f64_dpp %[acc], %[w" #B "], %[m" #B "_" #K "] row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
#define FCCQP_MV1(B) FCCQP_MVF(B, 0)
#define FCCQP_MV2(B) FCCQP_MV1(B) FCCQP_MVF(B, 1)
#define FCCQP_MV3(B) FCCQP_MV2(B) FCCQP_MVF(B, 2)
#define FCCQP_MV4(B) FCCQP_MV3(B) FCCQP_MVF(B, 3)
#define FCCQP_MV5(B) FCCQP_MV4(B) FCCQP_MVF(B, 4)
#define FCCQP_MV6(B) FCCQP_MV5(B) FCCQP_MVF(B, 5)
#define FCCQP_MV7(B) FCCQP_MV6(B) FCCQP_MVF(B, 6)
#define FCCQP_MV8(B) FCCQP_MV7(B) FCCQP_MVF(B, 7)
#define FCCQP_MV9(B) FCCQP_MV8(B) FCCQP_MVF(B, 8)
#define FCCQP_MV10(B) FCCQP_MV9(B) FCCQP_MVF(B, 9)
#define FCCQP_MV11(B) FCCQP_MV10(B) FCCQP_MVF(B, 10)
#define FCCQP_MV12(B) FCCQP_MV11(B) FCCQP_MVF(B, 11)
#define FCCQP_MV13(B) FCCQP_MV12(B) FCCQP_MVF(B, 12)
#define FCCQP_MV14(B) FCCQP_MV13(B) FCCQP_MVF(B, 13)
#define FCCQP_MV15(B) FCCQP_MV14(B) FCCQP_MVF(B, 14)
#define FCCQP_MV16(B) FCCQP_MV15(B) FCCQP_MVF(B, 15)
#define FCCQP_MVOPS(B, A) [m##B##_0] "v"(A[0]), [m##B##_1] "v"(A[1]), [m##B##_2] "v"(A[2]), [m##B##_3] "v"(A[3]), [m##B##_4] "v"(A[4]), [m##B##_5] "v"(A[5]), [m##B##_6] "v"(A[6]), [m##B##_7] "v"(A[7]), [m##B##_8] "v"(A[8]), [m##B##_9] "v"(A[9]), [m##B##_10] "v"(A[10]), [m##B##_11] "v"(A[11]), [m##B##_12] "v"(A[12]), [m##B##_13] "v"(A[13]), [m##B##_14] "v"(A[14]), [m##B##_15] "v"(A[15])
#define FCCQP_MV_SINGLE(L) asm(FCCQP_MV##L(0) : [acc] "+v"(acc) : [w0] "v"(w0), FCCQP_MVOPS(0, m0))
#define FCCQP_MV_PAIR(L) asm(FCCQP_MV16(0) FCCQP_MV##L(1) : [acc] "+v"(acc) : [w0] "v"(w0), [w1] "v"(w1), FCCQP_MVOPS(0, m0), FCCQP_MVOPS(1, m1))
#define FCCQP_MV_DISPATCH(STMT) \
  if constexpr (L == 16) STMT(16); else if constexpr (L == 15) STMT(15); else if constexpr (L == 14) STMT(14); \
  else if constexpr (L == 13) STMT(13); else if constexpr (L == 12) STMT(12); else if constexpr (L == 11) STMT(11); \
  else if constexpr (L == 10) STMT(10); else if constexpr (L == 9) STMT(9); else if constexpr (L == 8) STMT(8); \
  else if constexpr (L == 7) STMT(7); else if constexpr (L == 6) STMT(6); else if constexpr (L == 5) STMT(5); \
  else if constexpr (L == 4) STMT(4); else if constexpr (L == 3) STMT(3); else if constexpr (L == 2) STMT(2); \
  else STMT(1)
// blocks Q and Q + 1 (block Q whole, block Q + 1 with L columns), or -- the last of an odd number -- block Q alone
template <int Q, int NN, int NQ>
__device__ __forceinline__ void dpp_matvec_pairs(double &acc, const double (&Wq)[NQ], const double (&Mrow)[NN]) {
  if constexpr (Q + 1 < NQ) {
    constexpr int L = (NN - 16 * (Q + 1) < 16) ? NN - 16 * (Q + 1) : 16;
    double m0[16], m1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      m0[i] = Mrow[16 * Q + i];
      m1[i] = Mrow[(16 * (Q + 1) + i < NN) ? 16 * (Q + 1) + i : NN - 1];      // (past L: never named by an instruction)
    }
    const double w0 = Wq[Q], w1 = Wq[Q + 1];
    FCCQP_MV_DISPATCH(FCCQP_MV_PAIR);
    dpp_matvec_pairs<Q + 2, NN, NQ>(acc, Wq, Mrow);
  } else if constexpr (Q < NQ) {
    constexpr int L = (NN - 16 * Q < 16) ? NN - 16 * Q : 16;
    double m0[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m0[i] = Mrow[(16 * Q + i < NN) ? 16 * Q + i : NN - 1];
    const double w0 = Wq[Q];
    FCCQP_MV_DISPATCH(FCCQP_MV_SINGLE);
  }
}

// All four blocks in one statement (shapes with 49 <= n <= 63): one `s_waitcnt` for both pairs of w copies in front
// of it instead of two.
#define FCCQP_MV_QUAD(L)                                                                                      \
  asm(FCCQP_MV16(0) FCCQP_MV16(1) FCCQP_MV16(2) FCCQP_MV##L(3)                                                \
      : [acc] "+v"(acc) : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), FCCQP_MVOPS(0, m0), FCCQP_MVOPS(1, m1), \
        FCCQP_MVOPS(2, m2), FCCQP_MVOPS(3, m3))
template <int NN>
__device__ __forceinline__ void dpp_matvec_quad(double &acc, const double (&Wq)[4], const double (&Mrow)[NN]) {
  static_assert(NN > 48 && NN <= 64, "four blocks");
  constexpr int L = NN - 48;
  double m0[16], m1[16], m2[16], m3[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    m0[i] = Mrow[i]; m1[i] = Mrow[16 + i]; m2[i] = Mrow[32 + i];
    m3[i] = Mrow[(48 + i < NN) ? 48 + i : NN - 1];
  }
  const double w0 = Wq[0], w1 = Wq[1], w2 = Wq[2], w3 = Wq[3];
  FCCQP_MV_DISPATCH(FCCQP_MV_QUAD);
}

// forces v[J .. NN) into (arch) VGPRs at this point of the program, ten at a time
template <int J, int NN>
__device__ __forceinline__ void pin_vgprs(double (&v)[NN]) {
  if constexpr (J + 10 <= NN) {
    asm volatile("" : "+v"(v[J]), "+v"(v[J + 1]), "+v"(v[J + 2]), "+v"(v[J + 3]), "+v"(v[J + 4]), "+v"(v[J + 5]), "+v"(v[J + 6]),
                 "+v"(v[J + 7]), "+v"(v[J + 8]), "+v"(v[J + 9]));
    pin_vgprs<J + 10>(v);
  } else if constexpr (J < NN) {
    asm volatile("" : "+v"(v[J]));
    pin_vgprs<J + 1>(v);
  }
}

// ---- the per-lane roles of the ADMM layout (Shape::lane_variable) and the per-instance small vectors ----------
template <class SH>
struct AdmmLane {
  int var, vi, cidx, ccomp;
  bool isvar, is_cone;
  const double *lbp, *ubp, *mup;
  double *xs, *mux_s, *mulc_s;
  __device__ __forceinline__ AdmmLane(const KernelArgs &a, const int inst, const int lane) {
    constexpr int n = SH::n, nc = SH::nc;
    var = SH::lane_variable(lane);
    isvar = var >= 0;
    vi = isvar ? var : 0;
    is_cone = SH::lane_is_cone(lane);
    cidx = is_cone ? SH::lane_contact(lane) : 0;
    ccomp = is_cone ? SH::lane_component(lane) : 0;
    mup = a.mu + (size_t)inst * (nc / 3);
    lbp = a.lb + (size_t)inst * n;
    ubp = a.ub + (size_t)inst * n;
    xs = a.x + (size_t)inst * n;
    mux_s = a.mu_x + (size_t)inst * n;
    mulc_s = a.mu_lc + (size_t)inst * nc;
  }
};
// `fresh()` hides a pointer from CSE so that a load is redone at the point of use instead of its value being
// kept in a register (or in scratch) across the factorisation
__device__ __forceinline__ const double *fresh(const double *p) { asm volatile("" : "+s"(p)); return p; }
// The ADMM lane's own small inputs -- bounds, friction coefficient and, on a warm start, its duals -- ISSUED EARLY: right
// after the Gram product, a presolve ahead of their first use.  (Through round 3 they were loaded where the loop begins
// and again in the epilogue, and the wave sat through a memory round trip each time -- 1-2 us with the chip under
// load; before the Gram product there is no register to spare for them, after it there is.)  Unconditional loads
// from clamped indices: no branch; the lane kind is applied where the values are used.
struct LaneInputs { double lb, ub, mu, mux, mulc; };
template <class SH>
__device__ __forceinline__ LaneInputs load_lane_inputs(const KernelArgs &a, const AdmmLane<SH> &L) {
  LaneInputs v;
  v.lb = fresh(L.lbp)[L.vi];
  v.ub = fresh(L.ubp)[L.vi];
  v.mu = fresh(L.mup)[L.cidx];
  v.mux = 0.0;
  v.mulc = 0.0;
  if (a.warm_start) {  // fcc_qp.cpp:136-139
    v.mux = L.mux_s[L.vi];
    v.mulc = L.mulc_s[L.is_cone ? L.vi - SH::s : 0];
  }
  return v;
}

// ---- presolve by conjugate gradients (round 4) ------------------------------------------------------------------
// x0 = W^-1 x_c with W = I - rho G (the Woodbury form of the reference's presolve, fcc_qp.cpp:159-178; see the kernel
// body).  W is symmetric positive definite exactly when K is non-singular, its eigenvalues are 1 on range(A_eq^T) and
// lambda / (lambda + rho) for the eigenvalues lambda of the reduced Hessian -- at most n - m + 1 DISTINCT eigenvalues, so
// conjugate gradients end in at most n - m + 1 steps in exact arithmetic whatever the condition number, and the rows
// of rho G are in registers already (Mrow, the ADMM layout): an iteration is the ADMM loop's mat-vec, two wave
// reductions and three FMAs, ~0.3 us.  Emulated in fp64 on the CPU (tools/cg_presolve_experiment.py) and measured on
// the device: 11-13 iterations to a relative residual of 1e-13 on W-sat / W-psd (cond(W) 24 .. 450), 4-12 on W-mix,
// x0 within 7e-14 of the direct solve -- 3.5 us against the 11.9 us of factorising W by tiles and sweeping twice.
// Returns false -- and the caller takes that direct path, with its own singularity rule and the COD hand-off -- when
//   * the diagonal of W has an entry <= kPresolveMinPivot (every pivot of W's LDL^T is <= its diagonal entry: the direct
//     path would declare K singular; a variable without cost or constraint -- the tested case -- shows exactly so),
//   * a search direction has a Rayleigh quotient p.Wp / p.p <= kCgMinRayleigh = 1e-6 (W has an eigenvalue at least that
//     small: with the stop rule on the residual alone the component of x0 along it could be off by tolerance / 1e-6 and
//     more -- ADVICE r4; the direct solve resolves it.  p.p costs no reduction: the residuals of the method are mutually
//     orthogonal, so p.p = r.r + beta^2 p'.p' from the step before.  Through round 4: p.Wp <= 1e-10 r.r, which this
//     implies since p.p >= r.r), or
//   * the residual has not reached 1e-13 |x_c| after kCap iterations (ill-conditioned W).
// On success x is x0 in the ADMM layout (this lane's variable).  vbuf is used for the exchange of p.
constexpr double kCgTolSquared = 1e-26;
constexpr double kCgMinRayleigh = 1e-6;
template <class SH>
__device__ __forceinline__ bool cg_presolve(lds_f64 *vbuf, const int lane, const bool isvar, const int var,
                                            const double (&Mrow)[SH::n], const double xc, const double gdiag, double &x) {
  constexpr int n = SH::n, NQ = (n + 15) / 16;
  constexpr int kCap = (n - SH::m > 0 ? n - SH::m : n) + 10;
  if (__builtin_amdgcn_ballot_w64(isvar && !(1.0 - gdiag > kPresolveMinPivot)) != 0) return false;
  lds_f64 *const pslot = vbuf + (isvar ? var : kWave - 1);
  double r = xc, p = xc, xs = 0.0;                           // x = 0: r = b = x_c (zero in a lane without a variable)
  double rs = wave_sum(r * r);
  const double bb = rs;
  if (!(bb > 0.0)) { x = 0.0; return bb == 0.0; }            // x_c == 0: x0 = 0 (NaN: leave it to the direct path)
  const double stop = kCgTolSquared * bb;
  double pp = rs;                                            // p.p (p = r)
  bool done = false;
#pragma nounroll
  for (int it = 0; it < kCap; ++it) {
    // Wp = p - (rho G) p: the ADMM loop's mat-vec on p
    *pslot = p;
    lds_fence();
    double Wq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) Wq[q] = vbuf[16 * q + (lane & 15)];
    lds_fence();
    double acc = 0.0;
    if constexpr (NQ == 4) dpp_matvec_quad<n>(acc, Wq, Mrow);
    else dpp_matvec_pairs<0, n, NQ>(acc, Wq, Mrow);
    const double wp = p - acc;
    const double pwp = wave_sum(p * wp);
    if (!(pwp > kCgMinRayleigh * pp)) break;                 // (also NaN)
    const double alpha = rs * fast_rcp(pwp);
    xs = fma(alpha, p, xs);
    r = fma(-alpha, wp, r);
    const double rs_new = wave_sum(r * r);
    if (rs_new <= stop) { done = true; break; }
    const double beta = rs_new * fast_rcp(rs);
    p = fma(beta, p, r);
    pp = fma(beta * beta, pp, rs_new);
    rs = rs_new;
  }
  lds_fence();
  x = xs;
  return done;
}

// P4: ADMM iterations (fcc_qp.cpp:74-110) from x (the presolve's x0, or the warm-start state) with this lane's
// row Mrow of rho G and entry xc of x_c; leaves the last iterate in x, the duals in the warm-start state.
template <class SH>
__device__ __forceinline__ void admm_iterations(const KernelArgs &a, const AdmmLane<SH> &L, const int lane, lds_f64 *vbuf,
                                                double (&Mrow)[SH::n], const double xc, double &x, int &n_iter,
                                                double &x_res_norm, double &lc_res_norm, const LaneInputs &in) {
  constexpr int n = SH::n, s0 = SH::s;
  const bool isvar = L.isvar, is_cone = L.is_cone;
  const int var = L.var, vi = L.vi, cidx = L.cidx, ccomp = L.ccomp;
  double *mux_s = L.mux_s, *mulc_s = L.mulc_s;
  const double vlb = isvar ? in.lb : -HUGE_VAL;
  const double vub = isvar ? in.ub : HUGE_VAL;
  const double vmu = is_cone ? in.mu : 0.0;
  double mux = isvar ? in.mux : 0.0, mulc = is_cone ? in.mulc : 0.0;    // (zero on a cold start, fcc_qp.cpp:136-139)
  double xbar = x, lcbar = x;  // :74-75
  double xr = 0.0, lr = 0.0;
  // The loop's wave-uniform operands (lane masks, the two eps) are re-made here as per-lane
  // VGPR values: as SGPRs their live ranges span P3, the allocator spills them to VGPR lanes and
  // the loop reloads each one with v_readlane pairs + wait states, several times per iteration;
  // a VGPR copy costs one v_cmp per use at most (and is parked in an AGPR across P3 for free).
  int var_i = isvar ? 1 : 0, cone_i = is_cone ? 1 : 0, z_i = (ccomp == 2) ? 1 : 0;
  double eps_b = a.eps_bound, eps_f = a.eps_fcone;
  const double vkc = cone_lane_constant(vmu, ccomp == 2);
  double dir_xy = (ccomp == 2) ? 0.0 : 1.0, dir_z = (ccomp == 2) ? 1.0 : 0.0;     // project_cone_component_lean
  asm volatile("" : "+v"(var_i), "+v"(cone_i), "+v"(z_i), "+v"(eps_b), "+v"(eps_f), "+v"(dir_xy), "+v"(dir_z));
  // w goes to vbuf[variable]; lanes without a variable write entry 63, which is no variable's (n < 64)
  // and is never multiplied (the mat-vec stops at column n-1)
  lds_f64 *const wslot = vbuf + (isvar ? var : kWave - 1);
  // the trip count as a fresh SGPR: the kernarg copy's live range spans P3 and it is spilled to a VGPR
  // lane, read back by v_readlane on every iteration
  int max_iter = a.max_iter;
  asm volatile("" : "+s"(max_iter));
  // x = x_c + (rho G) w   == the KKT solve of :83-87, this lane = row `var` of rho G.  Every lane needs all
  // of w.  Reading it back as LDS broadcasts (ds_read_b128 of two doubles into all 64 lanes: a
  // 1-KiB return per instruction) made the four waves of a CU wait on the LDS return path: 409 ns
  // per mat-vec.  Instead each 16-lane row takes a copy of w, 16 entries per register
  // (Wq[lane] = w[16 q + (lane & 15)]: NQ plain ds_read_b64), and the FMA itself fetches its
  // operand from lane l of the row (v_fmac_f64 with the DPP control row_newbcast:l, the one DPP
  // mode the fp64 ALU has): no LDS traffic, no extra instruction (tools/dpp_bench.hip).
  // The write of w and the reads of its copies are issued at the END of an iteration, before the exit
  // test, which then runs in the shadow of the LDS round trip (the wave's only one per iteration).
  constexpr int NQ = (n + 15) / 16;
  double Wq[NQ];
  // (lane masks of the loop as 64-bit scalars: the selects and the exit test below take them as SGPR pairs)
  const unsigned long long var_lanes = __builtin_amdgcn_ballot_w64(var_i != 0), cone_lanes = __builtin_amdgcn_ballot_w64(cone_i != 0),
                           z_lanes = __builtin_amdgcn_ballot_w64(z_i != 0);
  auto exchange_w = [&]() {
    // w = x_bar - mu_x, contact segment overwritten by lambda_c_bar - mu_lambda_c (:81-82); both differences, then
    // one select (the compiler's own choice was to select both operands first: one instruction more)
    const double w = select_lanes(cone_lanes, lcbar - mulc, xbar - mux);
    *wslot = w;
    lds_fence();     // (one wave: its LDS operations complete in order; this is a compiler fence)
#pragma unroll
    for (int q = 0; q < NQ; ++q) Wq[q] = vbuf[16 * q + (lane & 15)];
    lds_fence();
  };
  // rho G's row in ArchVGPRs from here on: its live range spans the presolve, where the allocator may park
  // parts of it in AGPRs -- and then reads them back with v_accvgpr_read on EVERY iteration of the loop
  // (seen once: 38.5 -> 44.7 us for the loop after an unrelated change upstream)
  pin_vgprs<0>(Mrow);
  double xc_v = xc;            // likewise x_c, the start of the mat-vec's accumulator
  asm volatile("" : "+v"(xc_v));
  n_iter = max_iter;           // :78
  if (max_iter > 0) {
    exchange_w();
    // (the build aligns loop headers to 256 bytes, -falign-loops: the time per iteration moved by 5 % with
    // nothing but the placement of this -- otherwise identical -- loop in the code)
    int iter = 0, bound;
#pragma nounroll
    do {
      // (the DPP read of a VGPR needs two wait states after a VALU write of it.  Wq comes straight from the LDS
      // reads and the s_waitcnt sits in between; tests/test_code_objects.py fails on a build where the
      // compiler puts a copy there.  An `s_nop 1` pinned here used to cost 16 cycles per iteration.)
      double acc = xc_v;
      // (all four blocks in ONE asm statement where there are four: a second statement means a second s_waitcnt and an
      // s_nop between them, and that costs this loop 2 % -- measured; -DFCCQP_MV_TWO_STATEMENTS builds the pairs)
#ifndef FCCQP_MV_TWO_STATEMENTS
      if constexpr (NQ == 4) dpp_matvec_quad<n>(acc, Wq, Mrow);
      else dpp_matvec_pairs<0, n, NQ>(acc, Wq, Mrow);
#else
      dpp_matvec_pairs<0, n, NQ>(acc, Wq, Mrow);
#endif
      x = acc;
      // slack update :90-92.  The contact force (x + mu_lambda_c)[3c .. 3c+2] sits in the first three lanes of this
      // lane's 16-lane row (or quad, Shape::kRowContacts): three DPP broadcasts, no LDS round trip
      double xm = x + mulc;             // lanes outside the contacts: mulc == 0, value unused
      constexpr bool kRow = SH::kRowContacts;
      // Every lane runs both updates, unmasked (a select per state variable per iteration otherwise).  A
      // lane without a variable has x = mu_x = 0 and infinite bounds: it stays at zero.  Outside the
      // contacts lcbar / lr / mulc are garbage that nothing reads: w, the exit test, the reported norms
      // and the write-back all select on the lane's kind.
      {
        const double v = x + mux;
        // std::max(std::min(x(i), ub(i)), lb(i)), constraint_utils.cpp:43, as v_min_f64 / v_max_f64: the same value
        // for every finite x (also for lb > ub, and a NaN bound is ignored by both), one instruction each instead
        // of a compare and two selects; only a NaN x -- which the reference carries into x_bar -- is clamped here
        const double tt = __builtin_fmin(v, vub);
        xbar = __builtin_fmax(tt, vlb);
        xr = x - xbar;                    // :95
        mux += xr;                        // :101
      }
      {
        const double cfx = contact_bcast<0, kRow>(xm), cfy = contact_bcast<1, kRow>(xm), cfz = contact_bcast<2, kRow>(xm);
        lcbar = project_cone_component_lean(cfx, cfy, cfz, vmu, vkc, xm, dir_xy, dir_z);
        lr = x - lcbar;                   // :96
        mulc += lr;                       // :102
      }
#ifdef FCCQP_PROBE_NO_EXCHANGE
      // timing probe (WRONG results): the loop without its LDS exchange -- what a free exchange would buy
      { double w = select_lanes(cone_lanes, lcbar - mulc, xbar - mux); asm volatile("" :: "v"(w)); }
#else
      exchange_w();                       // for the next iteration (harmless after the last)
#endif
      // exit test :105-109.  norm_inf(r) < eps  <=>  every |r_i| < eps, so the per-iteration test is two compares
      // whose lane masks meet in scalar registers (s_and / s_or set SCC for the branch: no VALU instruction makes
      // a boolean out of them) instead of two wave reductions; the norms themselves (:97-98) are only reported,
      // and are reduced once after the loop.  An empty contact set has norm 0.  n_iter: the index of the
      // iteration that passed the test, or max_iter.
      // (no lane mask on the first compare: a lane without a variable has x_res == 0 exactly, which passes whenever a
      // real lane can pass at all -- with eps_bound <= 0 or NaN every lane fails, variable or not, and the loop runs
      // to max_iter either way)
      const unsigned long long fail_b = __builtin_amdgcn_ballot_w64(!(fabs(xr) < eps_b)),
                               fail_f = __builtin_amdgcn_ballot_w64(!(fabs(lr) < eps_f));
      // one exit: a converged iteration is not counted and drops the bound to zero.  Written out, because the
      // backend turns every formulation of "break, or count and compare" into the same twelve SALU instructions
      // (each one an issue slot of this lone wave like any other instruction): s_and / s_or of the masks -- the s_or
      // leaves SCC = "some lane fails" --, then bound = SCC ? max_iter : 0 and iter += SCC
      unsigned long long fail;
      asm volatile("s_and_b64 %2, %4, %5\n\ts_or_b64 %2, %2, %3\n\ts_cselect_b32 %1, %6, 0\n\ts_addc_u32 %0, %0, 0"
                   : "+s"(iter), "=&s"(bound), "=&s"(fail) : "s"(fail_b), "s"(fail_f), "s"(cone_lanes), "s"(max_iter) : "scc");
    } while (iter < bound);
    n_iter = iter;
  }
  if (a.max_iter > 0) {
    // :97-98  abs(max(maxCoeff, -minCoeff)) == max_i |r_i| of the last iteration executed
    x_res_norm = wave_max(isvar ? fabs(xr) : 0.0);
    lc_res_norm = wave_max(is_cone ? fabs(lr) : 0.0);
  }
  // duals are part of the warm-start state (fcc_qp.hpp:149-150)
  if (isvar) mux_s[vi] = mux;
  if (is_cone) mulc_s[vi - s0] = mulc;
}

// P5: epilogue metrics (fcc_qp.cpp:184-186) and write-back of z and the per-instance details
template <class SH>
__device__ __forceinline__ void epilogue_writeback(const KernelArgs &a, const AdmmLane<SH> &L, const int inst, const int lane,
                                                   const double x, const int n_iter, const int flags,
                                                   const double x_res_norm, const double lc_res_norm, const LaneInputs &in) {
  const bool isvar = L.isvar, is_cone = L.is_cone;
  const int vi = L.vi, cidx = L.cidx, ccomp = L.ccomp;
  double *xs = L.xs;
  lds_fence();
  const double ffy = contact_bcast<1, SH::kRowContacts>(x), ffz = contact_bcast<2, SH::kRowContacts>(x);   // this lane's contact force
  double bv = 0.0, cv = 0.0;
  if (isvar) {
    const double tt = min2(x, in.ub);
    const double p = max2(tt, in.lb);
    const double dd = x - p;
    bv = dd * dd;                                        // constraint_utils.cpp:61-65
  }
  if (is_cone && ccomp == 0) {
    const double fx = x, fy = ffy, fz = ffz;
    const double v = sqrt(fx * fx + fy * fy) - in.mu * fz;
    cv = (0.0 < v) ? v : 0.0;                            // constraint_utils.cpp:48-59
  }
  bv = sqrt(wave_sum(bv));
  cv = wave_sum(cv);

  if (isvar) xs[vi] = x;
  if (lane == 0) {
    a.n_iter[inst] = n_iter;
    a.flags[inst] = flags;
    a.metrics[0 * (size_t)a.batch + inst] = x_res_norm;
    a.metrics[1 * (size_t)a.batch + inst] = lc_res_norm;
    a.metrics[2 * (size_t)a.batch + inst] = bv;
    a.metrics[3 * (size_t)a.batch + inst] = cv;
  }
}

// TIN: storage type of Q and A_eq (double, or float for the fp32-storage entry points: widened on
// load).  A template parameter rather than a run-time flag: an extra code path in the assembly
// phase perturbed the register allocation of the whole kernel (36 VGPR spills, +30% on the ADMM
// loop), and even an unused extra kernarg field cost 3%.
//
// SPLIT = true is the first half of the two-launch form (FCCQP_KERNEL_FUSED_SPLIT): the kernel ends after the
// presolve and leaves this lane's row of rho G, x_c (workspace, lane-major: 512-byte stores) and x0 (the
// warm-start state) for admm_kernel below, which runs P4 + P5 at two or three waves per SIMD.
// ORDERED = true: block b solves instance a.inst_list[b] (the longest-first order of a chain's next solve,
// launch_order_by_iterations) instead of the fixed permutation.  An instantiation of its own: the test of a.inst_list in
// front of the whole-solve kernel cost the headline 0.6 % (same-box A/B of builds), for a feature it does not use.
template <class SH, class TIN, bool SPLIT = false, bool ORDERED = false>
__global__ __launch_bounds__(kWave, 1) void fused_kernel(const KernelArgs a) {
  constexpr int n = SH::n, m = SH::m, nc = SH::nc, s0 = SH::s, N = SH::N;
  __shared__ __attribute__((aligned(16))) double lds_raw[SH::kLdsDoubles];
  lds_f64 *const M = (lds_f64 *)lds_raw;
  lds_f64 *const vbuf = M + SH::kMat;

  const int lane = threadIdx.x;
  // block -> instance map: the fixed permutation of fccqp_device.h, or -- the next solve of a chain -- the caller's
  // longest-first order (launch_order_by_iterations)
  int inst;
  if constexpr (ORDERED) {
    if (blockIdx.x >= (unsigned)a.batch) return;
    inst = __builtin_amdgcn_readfirstlane(a.inst_list[blockIdx.x]);
  } else {
    inst = block_instance(a);
    if (inst < 0) return;
  }
  // Staggered start.  A launch of several wave-rounds keeps the resident waves (1024 on MI355X) in lock step from the first load to
  // the last store: the four waves of a CU reach every LDS-heavy piece (tile exchanges, the rho G image) and every
  // burst of global loads together, round after round.  Same-box A/B at batch 4096: starting the waves of SIMD k
  // k x 1.9 us late takes 2.7 % off the launch (W-sat; the phases then interleave, and the price is at most 5.6 us at
  // the end); 0.9 us steps gain 2.2 %, 3.7 us steps 2.7 %, 5.6 us steps 1.4 %.  Only the first round waits, and only
  // in launches of at least two rounds: a one-round launch would just end later.  (HW_REG_HW_ID bits 5:4: the SIMD.)
  // (a.m: the number of workgroups resident at once, put there by the launcher -- see KernelArgs)
  const unsigned resident = (unsigned)a.m;
  if (gridDim.x >= 2u * resident && blockIdx.x < resident) {
    const int simd = (int)((__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11))) & 3);
    for (int k = 0; k < 4 * simd; ++k) __builtin_amdgcn_s_sleep(16);       // 16 x 64 cycles ~ 0.47 us
  }

  const TIN *__restrict__ Q = reinterpret_cast<const TIN *>(a.Q) + (size_t)inst * n * n;
  const TIN *__restrict__ A = reinterpret_cast<const TIN *>(a.A) + (size_t)inst * m * n;
  const double *__restrict__ bq = a.b + (size_t)inst * n;
  const double *__restrict__ beq = a.b_eq + (size_t)inst * m;
  double *xs = a.x + (size_t)inst * n;

  // ---- vectors.  Two lane <-> variable maps: the natural one (lane i <-> variable i, i < n; lane r <->
  // equality row r) for the right-hand side and the presolve solve, and
  // Shape::lane_variable (each contact in a quad of lanes) for rho G, the ADMM loop and the epilogue ----
  // The small vectors (b, b_eq, lb, ub, mu, warm-start state) are fetched from global memory at
  // the point of use instead of being held in registers across the phases: the factorisation needs
  // nearly the whole register file, and anything live across it is spilled to scratch and reloaded inside
  // the ADMM loop (AdmmLane, fresh()).
  const bool isnat = lane < n;
  const int var = SH::lane_variable(lane);
  const bool isvar = var >= 0;
  const int vi = isvar ? var : 0;
  double x = 0.0;
  if (a.warm_start) x = isvar ? xs[vi] : 0.0;  // fcc_qp.cpp:136-139 (duals are fetched at P4)
  // nc > 0: never the reference's equality-only case (fcc_qp.cpp:132-133)
  int flags = 0;
  long long ticks = 0;

  // [-b; b_eq] (fcc_qp.cpp:149-150) into vbuf[0..N), and the border columns K_rho[0:N1, N1 + b] =
  // A_eq[N1 + b - n][0:n] (then zeros; Shape::kBorder) into their LDS staging area.  The global loads are
  // issued between the tile loads of row panel 0 and the rest (rhs_fetch) and stored after all of them
  // (rhs_store): memory returns in order, so a wait for a load issued behind the 88 tile loads would wait
  // for all of them, and the first diagonal tile could not be factorised while the matrix is in flight.
  double rhs_b = 0.0, rhs_e = 0.0;
  TIN rhs_border[SH::kBorder > 0 ? SH::kBorder : 1];
  auto rhs_fetch = [&]() {   // unconditional (clamped) loads: no branch, no wait in front of the tile loads
    rhs_b = bq[isnat ? lane : 0];
    rhs_e = beq[(lane < m) ? lane : 0];
    if constexpr (SH::kBorder > 0) {
#pragma unroll
      for (int bb = 0; bb < SH::kBorder; ++bb) rhs_border[bb] = A[(SH::N1 + bb - n) * n + (isnat ? lane : 0)];
    }
  };
  auto rhs_store = [&]() {
    vbuf[lane] = isnat ? -rhs_b : 0.0;
    lds_fence();
    if (lane < m) vbuf[n + lane] = rhs_e;        // n + lane < N <= kVbuf
    if constexpr (SH::kBorder > 0) {
#pragma unroll
      for (int bb = 0; bb < SH::kBorder; ++bb) {
        lds_f64 *bp_ = M + SH::kBorderOffset + bb * SH::kTileRows;
        if (lane < SH::kTileRows) bp_[lane] = isnat ? (double)rhs_border[bb] : 0.0;
        if (kWave + lane < SH::kTileRows) bp_[kWave + lane] = 0.0;
      }
    }
    lds_fence();
  };

  int n_iter = 0;
  double x_res_norm = 0.0, lc_res_norm = 0.0;
  LaneInputs in{};
  // phase stamps (100 MHz): 0 load issue, 1 factorisations, 2 presolve (W), 3 Gram product, 4 ADMM loop, 5 total
  long long ph_asm = 0, ph_solve = 0, ph_p3 = 0, ph_p4 = 0;
  const long long t_begin = wall_clock64();

  // ---- P0 + P1: K_rho = K + rho [I 0; 0 0], loaded and factorised ONCE (fcc_qp.cpp:62-68) ------------
  // The reference factorises twice on a cold start: K for the presolve x0 = (K^-1 [-b; b_eq]).head(n)
  // (fcc_qp.cpp:159-178) and K_rho for the iterations.  Here x0 comes out of the K_rho factorisation
  // (see "presolve" below), so Q and A_eq are read from HBM exactly once and the second N^3/3 is
  // replaced by an n^3/3 one.
#ifdef FCCQP_FINE_STAMPS
  long long fine_k[3] = {0, 0, 0}, fine_w[3] = {0, 0, 0};   // K_rho: diagonal tiles, -, Gram product; W: diagonal tiles
#else
  long long *const fine_k = nullptr, *const fine_w = nullptr;
#endif
  {
    // ---- P1 + P3: factorisation of K_rho with the forward solve of [E | -b; b_eq] fused in, then
    // rho G = rho (K_rho^-1)[0:n,0:n] and x_c = (K_rho^-1 [-b; b_eq])[0:n] as a Gram product ----------
    double Mrow[n];
    double xc, gdiag;
    long long t4;
    constexpr int NW = (n + 15) / 16;
    {
      constexpr int NRB = (SH::N1 + 15) / 16;
      d4 C[NRB][NRB], X[NRB][kWave / 16];
      // row panel 0 first, then the small vectors, then the rest: what the first panel needs arrives first,
      // and the wait for the vectors does not cover the bulk of the matrix
      TIN raw[NRB][NRB][4];
      issue_kkt_loads<SH, TIN, 0, 1>(Q, A, lane, raw);
      rhs_fetch();
      issue_kkt_loads<SH, TIN, 1, NRB>(Q, A, lane, raw);
      rhs_store();
      convert_kkt_tiles<SH, TIN>(raw, a.rho, lane, C);
      // (the loads are still in flight here: the phase boundary is nominal, the wait falls into the
      // factorisation's first use of each tile)
      const long long t0 = wall_clock64();
      ph_asm = t0 - t_begin;
      // info() is not checked by the reference (cpp:68)
      if (tile_factor_forward<SH>(M, vbuf, lane, C, X, fine_k)) flags |= kFlagAdmmNumerical;
      const long long t3 = wall_clock64();
      ticks = t3 - t0;
      if (gram_output<SH>(M, vbuf, lane, a.rho, var, X, Mrow, xc, gdiag, fine_k)) flags |= kFlagAdmmNumerical;
      t4 = wall_clock64();
      ph_p3 = t4 - t3;
    }
    // the ADMM lane's bounds / friction coefficient / warm duals: on their way from here (load_lane_inputs)
    if constexpr (!SPLIT) in = load_lane_inputs<SH>(a, AdmmLane<SH>(a, inst, lane));

    // ---- presolve (fcc_qp.cpp:159-178; skipped on warm start) ----------------------------------------
    // x0 = (K^-1 r).head(n), r = [-b; b_eq].  With E = [I_n; 0], K = K_rho - rho E E^T, and the
    // Woodbury identity gives
    //     x0 = (I - rho G)^-1 x_c,        G = E^T K_rho^-1 E,   x_c = E^T K_rho^-1 r
    // (equivalently: x0 is the fixed point of the unconstrained iteration x = x_c + rho G x).
    // W = I - rho G is symmetric, with eigenvalue 1 on range(A_eq^T) and lambda / (lambda + rho) for
    // every eigenvalue lambda of the reduced Hessian Z^T Q Z: positive definite exactly when K is
    // non-singular, and far better conditioned than K (cond <= 1 + rho / lambda_min).  It is
    // factorised by the same register tiles (n x n: n^3/3 instead of N^3/3 flops, no second assembly,
    // no second pass over Q / A_eq in HBM).  Against the exact K^-1 r this x0 is accurate to ~1e-14
    // relative on the walking workloads at the harness rho, ~1e-11 at rho = 0.1 (the direct
    // unpivoted LDL^T of K: ~1e-13; tools/woodbury_presolve.py), and it exists for semi-definite
    // costs with zero diagonal entries, where an unpivoted LDL^T of K breaks down and the
    // reference's own pivoted one fails over to its COD (fcc_qp.cpp:164-166).  Where W is singular
    // to working precision (a pivot <= kPresolveMinPivot: K is singular, the reference's COD gives
    // the minimum-norm solution) the instance goes to the device COD fallback, as before.
    // Since round 4 the system W x0 = x_c is solved by conjugate gradients on the rows of rho G this wave holds
    // anyway (cg_presolve: a dozen mat-vecs instead of a factorisation by tiles and two sweeps), and the direct
    // factorisation below is what runs when CG declines or fails: W's diagonal says "singular", a search direction
    // finds a vanishing eigenvalue, or the residual does not come down in n - m + 10 steps.
    bool presolved = false;
    if (!a.warm_start) {
      flags |= kFlagPresolveRan;
#ifndef FCCQP_PRESOLVE_DIRECT
      presolved = cg_presolve<SH>(vbuf, lane, isvar, var, Mrow, xc, gdiag, x);
      if (presolved) flags |= kFlagPresolveCg;
      if (!presolved) {          // the direct path reads x_c from vbuf[0..n), which carried CG's search directions
        lds_fence();
        if (isvar) vbuf[vi] = xc;
        lds_fence();
      }
#endif
    }
    if (!a.warm_start && !presolved) {
      using SW = typename SH::Reduced;
      // W arrives as register tiles from the Gram product; its factor goes to LDS (over the rho G image:
      // row `var` of rho G is in registers by now) for the two triangular sweeps
      const long long tw = wall_clock64();
      // the upper tiles of W = I - rho G (identity-padded to whole tiles), back out of the rho G image in LDS, which
      // conjugate gradients left alone (through round 3 gram_output produced them on every solve; now only this path
      // wants them)
      d4 Wt[NW][NW];
      w_tiles_from_image<SH>(M, lane, Wt);
      const bool wbad = tile_factor_store<SH>(M, lane, kPresolveMinPivot, Wt, fine_w);
#ifdef FCCQP_FINE_STAMPS
      fine_w[1] = wall_clock64() - tw;       // the whole factorisation of W
#endif
      ticks += wall_clock64() - tw;
      if (wbad) {
        flags |= kFlagPresolveNumerical;
        if (a.fb_list) {
          // hand the instance to the fallback pass (COD, fccqp_cod_impl.h + generic kernel in list
          // mode) and stop here
          if (lane == 0) {
            a.fb_list[atomicAdd(a.fb_count, 1)] = inst;
            a.flags[inst] = flags;
          }
          return;
        }
      }
      // the solve runs in the natural layout (x_c is still in vbuf[0..n) from P3); its result goes
      // through vbuf to the lanes of the ADMM loop
      double r0 = isnat ? vbuf[isnat ? lane : 0] : 0.0;
      solve_lds_one_slot<SW>(M, lane, r0);
      lds_fence();
      vbuf[lane] = r0;
      lds_fence();
      x = isvar ? vbuf[vi] : 0.0;
      lds_fence();
    }
    if (!a.warm_start) {
      const long long t5 = wall_clock64();
      ph_solve = t5 - t4;
      t4 = t5;
    }

    if constexpr (SPLIT) {
      // hand-over to admm_kernel: row `var` of rho G and x_c, lane-major (entry j of lane l at [j][l]); x0 goes
      // to the warm-start state, which is where admm_kernel (like a warm start) takes x from
      double *ws = a.workspace + (size_t)inst * (size_t)a.workspace_stride;
#pragma unroll
      for (int j = 0; j < n; ++j) ws[j * kWave + lane] = Mrow[j];
      ws[n * kWave + lane] = xc;
      if (!a.warm_start && isvar) xs[vi] = x;
      if (lane == 0) {
        a.flags[inst] = flags;
        a.factor_ticks[inst] = ticks;
        if (a.phase_ticks) {
          const size_t B = (size_t)a.batch;
          a.phase_ticks[0 * B + inst] = ph_asm;
          a.phase_ticks[1 * B + inst] = ticks;
          a.phase_ticks[2 * B + inst] = ph_solve;
          a.phase_ticks[3 * B + inst] = ph_p3;
          a.phase_ticks[5 * B + inst] = wall_clock64() - t_begin;
          a.phase_ticks[6 * B + inst] = ph_asm;
          a.phase_ticks[7 * B + inst] = t_begin;
        }
      }
      return;
    }
    // ---- P4: ADMM iterations (fcc_qp.cpp:74-110) ---------------------------------------------
    const AdmmLane<SH> L(a, inst, lane);
    admm_iterations<SH>(a, L, lane, vbuf, Mrow, xc, x, n_iter, x_res_norm, lc_res_norm, in);
    ph_p4 = wall_clock64() - t4;
  }

  // ---- P5: epilogue (fcc_qp.cpp:184-186) and write-back -----------------------------------------
  const AdmmLane<SH> L(a, inst, lane);
  epilogue_writeback<SH>(a, L, inst, lane, x, n_iter, flags, x_res_norm, lc_res_norm, in);
  if (lane == 0) {
    a.factor_ticks[inst] = ticks;
    if (a.phase_ticks) {
      const size_t B = (size_t)a.batch;
      a.phase_ticks[0 * B + inst] = ph_asm;
      a.phase_ticks[1 * B + inst] = ticks;
      a.phase_ticks[2 * B + inst] = ph_solve;
      a.phase_ticks[3 * B + inst] = ph_p3;
      a.phase_ticks[4 * B + inst] = ph_p4;
      a.phase_ticks[5 * B + inst] = wall_clock64() - t_begin;
#ifdef FCCQP_FINE_STAMPS
      // K_rho: bulk, load/store, in-panel; W: the same, packed as slots 6, 7 (K) and 0, 2, 3 (see tools)
      a.phase_ticks[0 * B + inst] = fine_k[0];
      a.phase_ticks[6 * B + inst] = fine_k[1];
      a.phase_ticks[7 * B + inst] = fine_k[2];
      a.phase_ticks[2 * B + inst] = fine_w[0];
      a.phase_ticks[3 * B + inst] = fine_w[1];
      a.phase_ticks[4 * B + inst] = fine_w[2];
#else
      a.phase_ticks[6 * B + inst] = ph_asm;
      a.phase_ticks[7 * B + inst] = t_begin;
#endif
    }
  }
}

// Second half of the two-launch form: P4 + P5 for one instance per wavefront from the rows of rho G that
// fused_kernel<SH, TIN, true> left in the workspace.  The loop needs ~170 registers and 512 bytes of LDS, so
// two or three of these waves share a SIMD (the whole-solve kernel holds its 512 registers from the first load
// to the last store and runs this loop alone on its SIMD, at about half the fp64 issue rate two waves reach).
// Instances that were handed to the COD pass (fb_list) are skipped; they are solved by the list launch.
template <class SH, int WAVES_PER_SIMD>
__global__ __launch_bounds__(kWave, WAVES_PER_SIMD) void admm_kernel(const KernelArgs a) {
  constexpr int n = SH::n;
  __shared__ __attribute__((aligned(16))) double vraw[kWave];
  lds_f64 *const vbuf = (lds_f64 *)vraw;
  const int lane = threadIdx.x;
  const int inst = block_instance(a);
  if (inst < 0) return;
  const long long t_begin = wall_clock64();
  const int flags = a.flags[inst];
  if (a.fb_list && (flags & kFlagPresolveNumerical)) return;
  const AdmmLane<SH> L(a, inst, lane);
  const double *ws = a.workspace + (size_t)inst * (size_t)a.workspace_stride;
  double Mrow[n];
#pragma unroll
  for (int j = 0; j < n; ++j) Mrow[j] = ws[j * kWave + lane];
  const double xc = ws[n * kWave + lane];
  double x = L.isvar ? L.xs[L.vi] : 0.0;
  int n_iter = 0;
  double x_res_norm = 0.0, lc_res_norm = 0.0;
  const LaneInputs in = load_lane_inputs<SH>(a, L);
  admm_iterations<SH>(a, L, lane, vbuf, Mrow, xc, x, n_iter, x_res_norm, lc_res_norm, in);
  epilogue_writeback<SH>(a, L, inst, lane, x, n_iter, flags, x_res_norm, lc_res_norm, in);
  if (lane == 0 && a.phase_ticks) {
    const size_t B = (size_t)a.batch;
    const long long dt = wall_clock64() - t_begin;
    a.phase_ticks[4 * B + inst] = dt;
    a.phase_ticks[5 * B + inst] += dt;
  }
}

}  // namespace fused
}  // namespace fccqp
