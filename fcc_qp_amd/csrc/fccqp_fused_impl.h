// fccqp_fused_impl.h -- shape-specialised gfx950 kernel: ONE WAVEFRONT PER QP, four one-wave
// workgroups per CU (one per SIMD), the whole solve fused in one launch.
//
// Hot path replaced (reference /root/reference):
//   FCCQP::Solve   src/fcc_qp.cpp:114-191   FCCQP::DoADMM  src/fcc_qp.cpp:57-112
//   projections    src/constraint_utils.cpp:5-65      Eigen LDLT call sites fcc_qp.cpp:68,86,161,174
//
// Why it looks the way it does (numbers from tools/microbench.hip on MI355X, gfx950):
//   * v_fma_f64 issues every ~4 cycles and one wave with one dependent chain already saturates
//     its SIMD: occupancy buys nothing for the arithmetic.
//   * only 256 of the 512 registers per lane are VALU-addressable (the rest are AGPRs), so the
//     98x98 KKT cannot live in registers: it lives in LDS (packed lower triangle, 39 KB) and is
//     factorised in place; registers hold one 16-column panel at a time.
//   * an operand every lane needs (same-address ds_read) costs 2 LDS cycles per double per CU;
//     v_readlane costs ~9 cycles per double.  Broadcast operands therefore come from LDS as
//     16-byte pairs, and the left-looking panel form reuses each lane-private operand 16 times.
//   * fp64 MFMA has the same 78.6 TFLOP/s peak as the fp64 VALU on gfx950 and the per-QP blocks
//     are triangular 98x98 updates and 60x60 mat-vecs, not contractions: no MFMA.
//
// Phases, N = n + m (walking shape: n = 60, m = 38, N = 98):
//   P0  assemble the lower triangle of K = [[Q (+rho I), A^T],[A, 0]] in LDS (coalesced 16-byte
//       reads of Q and A_eq; fcc_qp.cpp:146-150).
//   P1  presolve, skipped on warm start (fcc_qp.cpp:159): LDL^T of K, solve for [-b; b_eq] -> x0.
//   P2  K_rho assembled again and factorised (fcc_qp.cpp:62-68).
//       Factorisation = unpivoted LDL^T in natural order, left-looking by panels of 16 columns:
//       lane l owns rows l and 64+l; a panel's columns sit in registers, every earlier column q
//       contributes  T[i][j] -= (L[i][q] d_q) * L[j][q]  with L[i][q] a lane-private LDS read and
//       L[j][q], j in the panel, eight broadcast 16-byte reads.  L overwrites K; 1/d_j (with
//       Eigen's pseudo-inverse rule) is kept on the diagonal.
//   P3  61 right-hand sides at once, one per lane (e_0..e_{n-1} and [-b; b_eq]): forward / D /
//       backward substitution with L streamed out of LDS as broadcast operands.  Lane l ends up
//       holding row l of G = (K_rho^-1)[0:n,0:n]; lane n holds x_c = (K_rho^-1 [-b; b_eq])[0:n].
//   P4  ADMM iterations (fcc_qp.cpp:79-110): the KKT solve of an iteration is
//       x = x_c + (rho G) w,  w = x_bar - mu_x (lambda_c segment overwritten, fcc_qp.cpp:82):
//       a 60x60 mat-vec with w broadcast through LDS; projections, residuals, dual update and the
//       exit test are per-lane work plus two DPP wave reductions.
//   P5  epilogue metrics (fcc_qp.cpp:184-186) and write-back of z and the warm-start state.
//
// Pivot order: natural (no pivoting).  K_rho is symmetric quasi-definite, so the factorisation
// exists in every symmetric order; the reference's Eigen LDLT orders pivots by the largest
// remaining diagonal entry, which only reorders the same block elimination.  Parity with the
// reference-order oracle is ~1e-13 relative on the walking-shape workloads.
#pragma once
#include <cfloat>
#include <type_traits>

#include "fccqp_device.h"

namespace fccqp {
namespace fused {

constexpr int kWave = 64;
constexpr int kPanel = 16;
using d2 = double __attribute__((ext_vector_type(2)));
// explicit LDS address space: keeps every access a ds_* instruction (a generic pointer decays to
// flat_load / flat_store as soon as it goes through a select or a volatile cast)
using lds_f64 = __attribute__((address_space(3))) double;
using lds_d2 = __attribute__((address_space(3))) d2;

template <int I, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, E>(f);
  }
}

__device__ __forceinline__ void lds_fence() {
  // one-wave workgroup: LDS ops of a wave complete in order; this only pins the compiler's order
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double bcast_lane(double v, int src_lane) {  // src_lane wave-uniform
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double max2(double a, double b) { return (a < b) ? b : a; }  // std::max
__device__ __forceinline__ double min2(double a, double b) { return (b < a) ? b : a; }  // std::min

// ---- DPP wave reductions (no LDS traffic).  gfx9 DPP controls: quad_perm 0x00..0xff,
// row_half_mirror 0x141, row_mirror 0x140, row_bcast15 0x142, row_bcast31 0x143 ----------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get(double v, double old) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
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

// reference: project_to_friction_cone(Vector3d, mu), constraint_utils.cpp:5-25
__device__ __forceinline__ void project_cone3(double fx, double fy, double fz, double mu,
                                              double &ox, double &oy, double &oz) {
  const double nxy = sqrt(fx * fx + fy * fy);
  if (mu * fz >= nxy) { ox = fx; oy = fy; oz = fz; return; }      // :10-12
  if (fz < -mu * nxy) { ox = 0.0; oy = 0.0; oz = 0.0; return; }   // :15-17
  const double ratio = mu * fz / nxy;                              // :20
  double rx = ratio * fx, ry = ratio * fy, rz = fz;
  const double z2 = rx * rx + ry * ry + rz * rz;
  if (z2 > 0.0) {  // Eigen normalize(): zero vector untouched (the f_z == 0 quirk)
    const double nr = sqrt(z2);
    rx /= nr; ry /= nr; rz /= nr;
  }
  const double dot = rx * fx + ry * fy + rz * fz;                  // :23
  ox = dot * rx; oy = dot * ry; oz = dot * rz;
}

// ---- compile-time shape description ------------------------------------------------------------
template <int N_VARS, int N_EQ, int NC, int LC_START>
struct Shape {
  static constexpr int n = N_VARS, m = N_EQ, nc = NC, s = LC_START, N = N_VARS + N_EQ;
  static_assert(n >= 1 && n < kWave, "one lane per variable plus one lane for the rhs column");
  static_assert(N <= 2 * kWave, "two KKT rows per lane");
  static_assert(nc % 3 == 0 && s + nc <= n, "fcc_qp.cpp:32-33");
  // Packed lower triangle, column-major: element (i,j), i >= j, lives at cbase(j) + i.  Column
  // starts are padded so that cbase(j) is even: an element's address parity is its row parity and
  // ds_read_b128 of rows (2r, 2r+1) is always 16-byte aligned.
  static constexpr int cstart(int j) {
    return j * N - (j * (j - 1)) / 2 + ((N % 2 == 0) ? (j + 1) / 2 : j / 2);
  }
  static constexpr int cbase(int j) { return cstart(j) - j; }
  static constexpr int kMat = (cstart(N) + 1) & ~1;
  static constexpr int kNp = (N + 1) & ~1;
  // LDS carve (doubles): matrix | dvec[kNp] | vbuf[max(kNp, 64)]
  static constexpr int kVbuf = (kNp > kWave) ? kNp : kWave;
  static constexpr int kLdsDoubles = kMat + kNp + kVbuf;
  static_assert(kLdsDoubles * 8 <= 160 * 1024 / 4, "four one-wave workgroups per CU");

};

// Phase P3 is generated per shape by gen_p3.py (straight-line code; see there for why) and
// plugged in through this trait (specialised in fccqp_fused.hip): run() turns the factorisation in
// M into this lane's row of rho * G (Mrow) and its entry of x_c (xc).
//   static void run(const lds_f64 *M, lds_f64 *vbuf, int lane, bool isvar, const KernelArgs &a,
//                   double (&Mrow)[n], double &xc);
template <class SH>
struct P3;

// Left-looking panel LDL^T, in place in LDS.  On return M holds L below the diagonal, the
// diagonal slots hold 1/d (0 where |d| <= DBL_MIN: Eigen's LDLT::solve rule) and dvec holds d.
// Returns true if a pivot was exactly zero or not finite (where Eigen's info() can fail).
template <class SH>
__device__ __forceinline__ bool factor_lds(lds_f64 *M, lds_f64 *dvec, const int lane) {
  constexpr int N = SH::N, P = kPanel;
  bool bad_any = false;
  const int row0 = lane, row1 = kWave + lane;
  const bool has1 = row1 < N;
#pragma nounroll
  for (int c0 = 0; c0 < N; c0 += P) {
    // rows above the panel take no part: slot 0 is idle once c0 >= 64
    const bool use0 = c0 < kWave;
    double T0[P], T1[P];
    // (a) panel columns into registers (rows above the diagonal of a column read as 0)
#pragma unroll
    for (int jj = 0; jj < P; ++jj) {
      const int j = c0 + jj;
      const int cb = SH::cbase(j < N ? j : N - 1);
      T0[jj] = (use0 && j < N && row0 >= j) ? M[cb + row0] : 0.0;
      T1[jj] = (has1 && j < N && row1 >= j) ? M[cb + row1] : 0.0;
    }
    // (b) contribution of every earlier column q:  T[i][j] -= (L[i][q] d_q) L[j][q]
#pragma nounroll
    for (int q = 0; q < c0; ++q) {
      const int cb = SH::cbase(q);
      const double dq = dvec[q];
      const lds_d2 *bc = reinterpret_cast<const lds_d2 *>(M + cb + c0);  // rows c0.. of column q
      d2 w[P / 2];
#pragma unroll
      for (int t = 0; t < P / 2; ++t) w[t] = bc[t];  // may run past row N-1 in the last panel: unused
      const double u1 = has1 ? M[cb + row1] * dq : 0.0;
      if (use0) {
        const double u0 = M[cb + row0] * dq;  // rows < c0 compute garbage that is never stored
#pragma unroll
        for (int t = 0; t < P / 2; ++t) {
          T0[2 * t] = fma(-u0, w[t].x, T0[2 * t]);
          T0[2 * t + 1] = fma(-u0, w[t].y, T0[2 * t + 1]);
        }
      }
#pragma unroll
      for (int t = 0; t < P / 2; ++t) {
        T1[2 * t] = fma(-u1, w[t].x, T1[2 * t]);
        T1[2 * t + 1] = fma(-u1, w[t].y, T1[2 * t + 1]);
      }
    }
    // (c) factorise the panel itself, column by column
#pragma unroll
    for (int k = 0; k < P; ++k) {
      const int j = c0 + k;
      if (j < N) {
        const int src = j & (kWave - 1);
        const double d = (j < kWave) ? bcast_lane(T0[k], src) : bcast_lane(T1[k], src);
        const bool bad = !(fabs(d) > 0.0) || !(fabs(d) <= DBL_MAX);
        bad_any |= bad;
        const double rinv = 1.0 / d;
        const double lsc = bad ? 1.0 : rinv;  // Eigen scales the column only by a non-zero pivot
        const int cb = SH::cbase(j);
        const double a0 = T0[k], a1 = T1[k];
        const double l0 = a0 * lsc, l1 = a1 * lsc;
        if (use0 && row0 > j && row0 < N) M[cb + row0] = l0;  // row0 < N matters only when N < 64
        if (has1 && row1 > j) M[cb + row1] = l1;
        if (lane == 0) {
          dvec[j] = d;
          M[cb + j] = (fabs(d) > DBL_MIN) ? rinv : 0.0;
        }
        lds_fence();
        // rank-1 update of the panel columns to the right:  T[i][jj] -= a_i * l_{c0+jj, j}
        const lds_d2 *bc = reinterpret_cast<const lds_d2 *>(M + cb + c0);
#pragma unroll
        for (int t = (k + 1) / 2; t < P / 2; ++t) {
          const d2 w = bc[t];
          if (2 * t > k) {
            if (use0) T0[2 * t] = fma(-a0, w.x, T0[2 * t]);
            T1[2 * t] = fma(-a1, w.x, T1[2 * t]);
          }
          if (use0) T0[2 * t + 1] = fma(-a0, w.y, T0[2 * t + 1]);
          T1[2 * t + 1] = fma(-a1, w.y, T1[2 * t + 1]);
        }
      }
    }
    lds_fence();
  }
  return bad_any;
}

// x := (L D L^T)^-1 x with M as left by factor_lds; x0/x1 are this lane's rows lane / 64+lane.
template <class SH>
__device__ __forceinline__ void solve_lds(const lds_f64 *M, const int lane, double &x0, double &x1) {
  constexpr int N = SH::N;
  const int row0 = lane, row1 = kWave + lane;
  const bool ok0 = row0 < N, ok1 = row1 < N;
  const int cb0 = SH::cbase(ok0 ? row0 : 0), cb1 = SH::cbase(ok1 ? row1 : 0);
  // forward, unit lower, column oriented
#pragma nounroll
  for (int k = 0; k < N; ++k) {
    const int src = k & (kWave - 1);
    const double xk = (k < kWave) ? bcast_lane(x0, src) : bcast_lane(x1, src);
    const int cb = SH::cbase(k);
    if (ok0 && row0 > k) x0 = fma(-M[cb + row0], xk, x0);
    if (ok1 && row1 > k) x1 = fma(-M[cb + row1], xk, x1);
  }
  // the diagonal slot holds 1/d with Eigen's pseudo-inverse rule folded in
  x0 = ok0 ? x0 * M[cb0 + row0] : 0.0;
  x1 = ok1 ? x1 * M[cb1 + row1] : 0.0;
  // backward with L^T: x_i -= L[k][i] x_k, i < k
#pragma nounroll
  for (int k = N - 1; k > 0; --k) {
    const int src = k & (kWave - 1);
    const double xk = (k < kWave) ? bcast_lane(x0, src) : bcast_lane(x1, src);
    if (ok0 && row0 < k) x0 = fma(-M[cb0 + k], xk, x0);
    if (ok1 && row1 < k) x1 = fma(-M[cb1 + k], xk, x1);
  }
}

template <class SH>
__global__ __launch_bounds__(kWave, 1) void fused_kernel(const KernelArgs a) {
  constexpr int n = SH::n, m = SH::m, nc = SH::nc, s0 = SH::s, N = SH::N;
  __shared__ __attribute__((aligned(16))) double lds_raw[SH::kLdsDoubles];
  lds_f64 *const M = (lds_f64 *)lds_raw;
  lds_f64 *const dvec = M + SH::kMat;
  lds_f64 *const vbuf = dvec + SH::kNp;

  const int lane = threadIdx.x;
  const int inst = blockIdx.x;
  if (inst >= a.batch) return;

  const double *__restrict__ Q = a.Q + (size_t)inst * n * n;
  const double *__restrict__ A = a.A + (size_t)inst * m * n;
  const double *__restrict__ bq = a.b + (size_t)inst * n;
  const double *__restrict__ beq = a.b_eq + (size_t)inst * m;
  const double *__restrict__ mup = a.mu + (size_t)inst * (nc / 3);
  const double *__restrict__ lbp = a.lb + (size_t)inst * n;
  const double *__restrict__ ubp = a.ub + (size_t)inst * n;
  double *xs = a.x + (size_t)inst * n;
  double *mux_s = a.mu_x + (size_t)inst * n;
  double *mulc_s = a.mu_lc + (size_t)inst * nc;

  // ---- vectors, variable layout: lane i <-> variable i (i < n); lane r <-> equality row r ----
  const bool isvar = lane < n;
  const bool is_cone = (lane >= s0) && (lane < s0 + nc);
  const double vb = isvar ? bq[lane] : 0.0;
  const double vlb = isvar ? lbp[lane] : -HUGE_VAL;
  const double vub = isvar ? ubp[lane] : HUGE_VAL;
  const double vbe = (lane < m) ? beq[lane] : 0.0;
  const int cidx = is_cone ? (lane - s0) / 3 : 0;
  const int ccomp = is_cone ? (lane - s0) - 3 * cidx : 0;
  const double vmu = is_cone ? mup[cidx] : 0.0;
  double x = 0.0, mux = 0.0, mulc = 0.0;
  if (a.warm_start) {  // fcc_qp.cpp:136-139
    x = isvar ? xs[lane] : 0.0;
    mux = isvar ? mux_s[lane] : 0.0;
    mulc = is_cone ? mulc_s[lane - s0] : 0.0;
  }
  const bool all_inf = !isvar || (isinf(vlb) && isinf(vub));
  const bool equality_constrained = (nc == 0) && __all(all_inf);  // fcc_qp.cpp:132-133
  int flags = equality_constrained ? kFlagEqualityOnly : 0;
  long long ticks = 0;
  const bool cold = equality_constrained || !a.warm_start;  // presolve runs (fcc_qp.cpp:159)

  // P0: lower triangle of [[Q + rho I, A^T],[A, 0]] into LDS; Q and A_eq are read front to back,
  // 16 bytes per lane (fcc_qp.cpp:146-150; only Q[i][j], i >= j, is used, like Eigen's Lower LDLT)
  auto assemble = [&](double rho) {
    auto put = [&](int i, int j, double v) {  // KKT row i, column j, both < n
      if (i >= j) M[SH::cbase(j) + i] = v + ((i == j) ? rho : 0.0);
    };
    if constexpr (n % 2 == 0) {
      const d2 *Q2 = reinterpret_cast<const d2 *>(Q);
#pragma unroll 4
      for (int p = lane; p < n * n / 2; p += kWave) {
        const d2 v = Q2[p];
        const int e = 2 * p, i = e / n, j = e - i * n;  // n even: a pair never straddles two rows
        put(i, j, v.x);
        put(i, j + 1, v.y);
      }
      const d2 *A2 = reinterpret_cast<const d2 *>(A);
#pragma unroll 4
      for (int p = lane; p < m * n / 2; p += kWave) {
        const d2 v = A2[p];
        const int e = 2 * p, r = e / n, j = e - r * n;
        M[SH::cbase(j) + n + r] = v.x;
        M[SH::cbase(j + 1) + n + r] = v.y;
      }
    } else {
      for (int e = lane; e < n * n; e += kWave) { const int i = e / n; put(i, e - i * n, Q[e]); }
      for (int e = lane; e < m * n; e += kWave) { const int r = e / n; M[SH::cbase(e - r * n) + n + r] = A[e]; }
    }
    // zero block: rows >= n of columns >= n
    for (int j = n; j < N; ++j)
      for (int i = j + lane; i < N; i += kWave) M[SH::cbase(j) + i] = 0.0;
    lds_fence();
  };

  int n_iter = 0;
  double x_res_norm = 0.0, lc_res_norm = 0.0;
  // phase stamps (100 MHz): 0 assemble, 1 factor, 2 presolve solve, 3 P3, 4 P4, 5 total
  long long ph_asm = 0, ph_solve = 0, ph_p3 = 0, ph_p4 = 0;
  const long long t_begin = wall_clock64();

#pragma nounroll
  for (int pass = cold ? 0 : 1; pass < 2; ++pass) {
    if (pass == 1 && equality_constrained) break;
    const long long ta = wall_clock64();
    assemble(pass == 0 ? 0.0 : a.rho);  // P1: K (fcc_qp.cpp:161);  P2: K_rho (fcc_qp.cpp:62-68)
    const long long t0 = wall_clock64();
    ph_asm += t0 - ta;
    const bool bad = factor_lds<SH>(M, dvec, lane);
    ticks += wall_clock64() - t0;
    if (pass == 0) {
      flags |= kFlagPresolveRan | (bad ? kFlagPresolveNumerical : 0);
      // x_ = (K^-1 [-b; b_eq]).head(n)   (fcc_qp.cpp:149-150,174)
      vbuf[lane] = isvar ? -vb : 0.0;
      lds_fence();
      if (lane < m) vbuf[n + lane] = vbe;  // n + lane < N <= kVbuf
      lds_fence();
      double r0 = (lane < N) ? vbuf[lane < N ? lane : 0] : 0.0;
      double r1 = (kWave + lane < N) ? vbuf[kWave + lane < N ? kWave + lane : 0] : 0.0;
      lds_fence();
      const long long ts = wall_clock64();
      solve_lds<SH>(M, lane, r0, r1);
      ph_solve += wall_clock64() - ts;
      x = isvar ? r0 : 0.0;  // n < 64: the variables are rows 0..n-1 = lanes 0..n-1 of slot 0
    } else if (bad) {
      flags |= kFlagAdmmNumerical;  // info() is not checked by the reference here (cpp:68)
    }
  }

  if (!equality_constrained) {
    // ---- P3: G = (K_rho^-1)[0:n,0:n] and x_c, one right-hand side per lane -------------------
    double Mrow[n];
    double xc;
    // right-hand sides: lanes < n take unit vectors, lane n takes [-b; b_eq] out of vbuf
    vbuf[lane] = isvar ? -vb : 0.0;
    lds_fence();
    if (lane < m) vbuf[n + lane] = vbe;
    lds_fence();
    const long long t3 = wall_clock64();
    P3<SH>::run(M, vbuf, lane, isvar, a, Mrow, xc);
    const long long t4 = wall_clock64();
    ph_p3 = t4 - t3;

    // ---- P4: ADMM iterations (fcc_qp.cpp:74-110) ---------------------------------------------
    double xbar = x, lcbar = x;  // :74-75
    n_iter = a.max_iter;         // :78
#pragma nounroll
    for (int iter = 0; iter < a.max_iter; ++iter) {
      // w = x_bar - mu_x, contact segment overwritten by lambda_c_bar - mu_lambda_c (:81-82)
      const double w = is_cone ? (lcbar - mulc) : (xbar - mux);
      vbuf[lane] = isvar ? w : 0.0;
      lds_fence();
      // x = x_c + (rho G) w   == the KKT solve of :83-87
      double acc0 = xc, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
      const lds_d2 *w2 = reinterpret_cast<const lds_d2 *>(vbuf);
      static_for<0, (n + 3) / 4>([&](auto qc) {
        constexpr int j = 4 * decltype(qc)::value;
        const d2 wa = w2[j / 2];
        acc0 = fma(Mrow[j], wa.x, acc0);
        if constexpr (j + 1 < n) acc1 = fma(Mrow[j + 1], wa.y, acc1);
        if constexpr (j + 2 < n) {
          const d2 wb = w2[j / 2 + 1];
          acc2 = fma(Mrow[j + 2], wb.x, acc2);
          if constexpr (j + 3 < n) acc3 = fma(Mrow[j + 3], wb.y, acc3);
        }
      });
      x = (acc0 + acc1) + (acc2 + acc3);
      lds_fence();
      // slack update :90-92
      if (is_cone) vbuf[lane] = x + mulc;
      lds_fence();
      double xr = 0.0, lr = 0.0;
      if (isvar) {
        const double v = x + mux;
        const double tt = min2(v, vub);   // std::min(x(i), ub(i))   constraint_utils.cpp:43
        xbar = max2(tt, vlb);             // std::max(., lb(i))
        xr = x - xbar;                    // :95
        mux += xr;                        // :101
      }
      if (is_cone) {
        const int base = s0 + 3 * cidx;
        double ox, oy, oz;
        project_cone3(vbuf[base], vbuf[base + 1], vbuf[base + 2], vmu, ox, oy, oz);
        lcbar = (ccomp == 0) ? ox : ((ccomp == 1) ? oy : oz);
        lr = x - lcbar;                   // :96
        mulc += lr;                       // :102
      }
      lds_fence();
      // :97-98  abs(max(maxCoeff, -minCoeff)) == max_i |r_i|; empty contact set defined as 0
      x_res_norm = wave_max(isvar ? fabs(xr) : 0.0);
      if constexpr (nc > 0) lc_res_norm = wave_max(is_cone ? fabs(lr) : 0.0);
      if (lc_res_norm < a.eps_fcone && x_res_norm < a.eps_bound) {  // :105-109
        n_iter = iter;
        break;
      }
    }
    ph_p4 = wall_clock64() - t4;
  }

  // ---- P5: epilogue (fcc_qp.cpp:184-186) and write-back -----------------------------------------
  lds_fence();
  if (is_cone) vbuf[lane] = x;
  lds_fence();
  double bv = 0.0, cv = 0.0;
  if (isvar) {
    const double tt = min2(x, vub);
    const double p = max2(tt, vlb);
    const double dd = x - p;
    bv = dd * dd;                                        // constraint_utils.cpp:61-65
  }
  if (is_cone && ccomp == 0) {
    const double fx = vbuf[lane], fy = vbuf[lane + 1], fz = vbuf[lane + 2];
    const double v = sqrt(fx * fx + fy * fy) - vmu * fz;
    cv = (0.0 < v) ? v : 0.0;                            // constraint_utils.cpp:48-59
  }
  bv = sqrt(wave_sum(bv));
  cv = wave_sum(cv);

  if (isvar) { xs[lane] = x; mux_s[lane] = mux; }
  if (is_cone) mulc_s[lane - s0] = mulc;
  if (lane == 0) {
    a.n_iter[inst] = n_iter;
    a.flags[inst] = flags;
    a.metrics[0 * (size_t)a.batch + inst] = x_res_norm;
    a.metrics[1 * (size_t)a.batch + inst] = lc_res_norm;
    a.metrics[2 * (size_t)a.batch + inst] = bv;
    a.metrics[3 * (size_t)a.batch + inst] = cv;
    a.factor_ticks[inst] = ticks;
    if (a.phase_ticks) {
      const size_t B = (size_t)a.batch;
      a.phase_ticks[0 * B + inst] = ph_asm;
      a.phase_ticks[1 * B + inst] = ticks;
      a.phase_ticks[2 * B + inst] = ph_solve;
      a.phase_ticks[3 * B + inst] = ph_p3;
      a.phase_ticks[4 * B + inst] = ph_p4;
      a.phase_ticks[5 * B + inst] = wall_clock64() - t_begin;
    }
  }
}

}  // namespace fused
}  // namespace fccqp
