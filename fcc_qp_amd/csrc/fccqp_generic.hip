// fccqp_generic.hip -- shape-generic gfx950 kernel: one wavefront runs one QP instance (plus three
// helper wavefronts for shapes that own a whole CU's LDS).
//
// Hot path replaced (reference /root/reference):
//   FCCQP::Solve   src/fcc_qp.cpp:114-191   (assemble KKT, presolve, epilogue metrics)
//   FCCQP::DoADMM  src/fcc_qp.cpp:57-112    (rho-shifted factorisation + ADMM loop)
//   projections    src/constraint_utils.cpp:5-65
//   Eigen::LDLT compute/solve at the call sites fcc_qp.cpp:68,86,161,174
//
// Design (MI355X).  The KKT matrix [[Q(+rho I), A^T],[A, 0]] is assembled into LDS as a packed
// lower triangle (column-major, column j holds rows j..N-1 contiguously, so a lane-per-row sweep
// of a column is bank-conflict free) and factorised in place by an unpivoted LDL^T in natural
// order (16-column left-looking panels, ldlt_panel).  The matrix is symmetric quasi-definite for
// rho > 0 / Q > 0, so every symmetric pivot order exists; the reference's Eigen LDLT pivots by the
// largest remaining diagonal entry, which only reorders the same block elimination -- results
// agree to ~1e-13 relative (tests/test_numerics_choice.py).  All vectors (rhs, x, duals, bounds)
// live in registers, lane i owning entries i, i+64, ... (SLOTS of them).  ADMM iterations start as
// two triangular sweeps per iteration, like the reference; an instance that keeps iterating
// switches to the explicit operator x = x_c + (rho G) w (solve_instance).  Shapes with N <= 128
// run one wave per workgroup and no barrier; shapes with 128 < N <= 198 own a CU's LDS and get
// four waves, wave 0 the solver and waves 1-3 helpers for assembly, factorisation and the G build
// (helper_wave).  Shapes whose packed KKT exceeds the LDS budget (N > 198) keep the matrix in a
// per-block global workspace and use the unblocked factorisation and sweeps only.
//
// This is the any-shape path and the first correct HIP path; the shape-specialised kernel in
// fccqp_fused.hip is the fast one for the walking shape.
#include "fccqp_device.h"
#include "fccqp_cod_impl.h"

#include <cfloat>
#include <type_traits>

namespace fccqp {
namespace {

constexpr int kWave = 64;
// explicit LDS address space: a generic `double *` that may point to LDS or global memory compiles
// to flat_load / flat_store, whose latency is several times that of ds_read / ds_write
using lds_f64 = __attribute__((address_space(3))) double;

__device__ __forceinline__ void wave_sync() {
  // Single-wave workgroup: LDS/global accesses of one wave execute in program order; the fence
  // keeps the compiler from reordering cross-lane producer/consumer accesses.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Matrix in the per-block GLOBAL workspace (N > 198): the lanes of the wave hand matrix entries to each
// other through memory, and the owner of an entry changes from phase to phase (assembly: lane =
// column; factorisation: lane = row - k - 1; solves: lane = row).  Between phases the wave's stores must be
// complete and its L1 lines dropped: agent_release_acquire() (fccqp_device.h), with the waits written out --
// the plain agent-scope fence lost its `s_waitcnt vmcnt(0)` on one control path of the 8-slot kernel, which is what
// produced the sporadically wrong iterates of round 2's N > 256 tests.
__device__ __forceinline__ void global_phase_sync() {
  agent_release_acquire();
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double lane_bcast(double v, int src_lane) {
  // src_lane is wave-uniform
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double o = __shfl_xor(v, off, kWave);
    v = (v < o) ? o : v;  // std::max(v, o)
  }
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double o = __shfl_xor(v, off, kWave);
    v = (o < v) ? o : v;
  }
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// Packed lower triangle, column-major: element (i,j), i >= j, at col_base(j) + i.  Column starts
// are padded so that col_base(j) is even: an element's address parity is its row parity and a
// ds_read_b128 of rows (2r, 2r+1) of any column is 16-byte aligned.
__host__ __device__ __forceinline__ int col_start(int j, int N) {
  return j * N - (j * (j - 1)) / 2 + ((N & 1) ? j / 2 : (j + 1) / 2);
}
__host__ __device__ __forceinline__ int col_base(int j, int N) { return col_start(j, N) - j; }
// col_base(j + 1) - col_base(j)
__device__ __forceinline__ int col_step(int j, int N) { return N - j - 1 + (((j ^ N) & 1) ? 0 : 1); }
__host__ __device__ __forceinline__ int packed_doubles(int N) { return (col_start(N, N) + 1) & ~1; }

__device__ __forceinline__ double fast_rcp(double d) {   // <= 1 ulp from the rounded quotient
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

__device__ __forceinline__ double fast_rsqrt(double a) {  // a > 0; Newton twice on v_rsq_f64: <= 1-2 ulp
  double r = __builtin_amdgcn_rsq(a);
  double e = fma(-a * r, r, 1.0);
  r = fma(0.5 * r, e, r);
  e = fma(-a * r, r, 1.0);
  r = fma(0.5 * r, e, r);
  return r;
}
// reference: project_to_friction_cone(Vector3d, mu), constraint_utils.cpp:5-25, for ONE component
// v = f[comp] of the contact force (every lane of a contact triple calls it with the same
// (fx, fy, fz)).  Same three branches, same f_z == 0 quirk; the side-of-cone branch in the form
// t (mu f_xy / |f_xy|, 1), t = (mu |f_xy| + f_z) / (1 + mu^2), with the per-lane constant
// kc = 1 / (1 + mu^2) (z) or mu / (1 + mu^2) (x, y) -- same formulation as the fused kernel, derivation
// in fccqp_fused_impl.h.
__device__ __forceinline__ double cone_lane_constant(double mu, bool is_z) {
  const double c = 1.0 / fma(mu, mu, 1.0);
  return is_z ? c : mu * c;
}
__device__ __forceinline__ double project_cone_component(double fx, double fy, double fz, double mu, double kc,
                                                         double v, bool is_z) {
  const double nxy2 = fx * fx + fy * fy;
  const bool flat = !(nxy2 > 0.0);
  const double rs = fast_rsqrt(flat ? 1.0 : nxy2);
  const double nxy = flat ? 0.0 : nxy2 * rs;
  const bool inside = mu * fz >= nxy;                   // :10-12
  const bool polar = fz < -mu * nxy;                    // :15-17
  const double t = fma(mu, nxy, fz) * kc;               // :20-23
  const double dir = is_z ? 1.0 : rs * v;
  const double side = (fz == 0.0) ? 0.0 : t * dir;      // zero ray: Eigen's normalize() no-op -> 0
  return inside ? v : (polar ? 0.0 : side);
}

constexpr int kPanel = 16;
constexpr int kGSwitchIter = 16;       // ADMM iterations by triangular sweeps before G is built ...
constexpr int kGSwitchIterMulti = 4;   // ... when four waves share the build on the matrix cores
// G is only built if at least this many iterations can still follow (the build costs about that
// many sweep iterations: it cannot pay off under a small max_iter, e.g. the paper's 15)
constexpr int kGMinRemaining = 32, kGMinRemainingMulti = 12;
using d2 = double __attribute__((ext_vector_type(2)));
using lds_d2 = __attribute__((address_space(3))) d2;
using d4 = double __attribute__((ext_vector_type(4)));

// Bulk update of a panel on the matrix cores, for one-wave workgroups (the scheme of mfma_bulk in
// fccqp_fused_impl.h with runtime N): K[i][j] -= sum_{q < c0} (L[i][q] d_q) L[j][q] for the 16 panel
// columns and every 16-row block at or below the panel, operands one element per lane, tiles read
// from and written back to the panel's own columns.  RBMAX bounds the row blocks (4 per slot).
template <int RBMAX>
__device__ __forceinline__ void mfma_bulk_rt(lds_f64 *M, const int N, const int lane, const int c0) {
  const int col = lane & 15, grp = lane >> 4;
  const int rb0 = c0 >> 4, nrb = (N + 15) >> 4;
  const int j = (c0 + col < N) ? c0 + col : N - 1;
  const int cbj = col_base(j, N);
  d4 acc[RBMAX];
#pragma unroll
  for (int rb = 0; rb < RBMAX; ++rb) {
    if (rb >= rb0 && rb < nrb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[rb][r] = M[cbj + 16 * rb + grp + 4 * r];   // rows past N-1: garbage rows
    }
  }
  struct Ops { double b, nd, av[RBMAX]; };
  auto fetch = [&](const int q0, Ops &o) {
    const int q = q0 + grp;
    const int cq = col_base(q, N);
    o.b = M[cq + c0 + col];                              // B[k][n] = L[c0 + n][q0 + k]
    o.nd = -M[cq + q];                                   // the diagonal slot holds d_q
#pragma unroll
    for (int rb = 0; rb < RBMAX; ++rb)
      if (rb >= rb0 && rb < nrb) o.av[rb] = M[cq + 16 * rb + col];
  };
  auto apply = [&](const Ops &o) {
#pragma unroll
    for (int rb = 0; rb < RBMAX; ++rb)
      if (rb >= rb0 && rb < nrb)
        acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.av[rb] * o.nd, o.b, acc[rb], 0, 0, 0);
  };
  {
    Ops oa, ob;
    fetch(0, oa);
#pragma nounroll
    for (int q0 = 0; q0 < c0; q0 += 8) {                 // c0 is a multiple of 16: steps pair up
      fetch(q0 + 4, ob);
      apply(oa);
      fetch((q0 + 8 < c0) ? q0 + 8 : q0 + 4, oa);        // last trip: a harmless re-read
      apply(ob);
    }
  }
  wave_sync();
#pragma unroll
  for (int rb = 0; rb < RBMAX; ++rb) {
    if (rb >= rb0 && rb < nrb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * rb + grp + 4 * r;
        if (row < N && c0 + col < N && row >= c0 + col) M[cbj + row] = acc[rb][r];
      }
    }
  }
  wave_sync();
}

// One 16-column panel of the left-looking LDL^T (matrix in LDS), columns c0..c0+15, for a wave
// whose lanes own rows lane + 64 t.  SD = c0 / 64 is the register slot the panel's diagonal block
// lives in; slots above it hold rows above the panel and are not touched.  Same scheme as the
// shape-specialised kernel (fccqp_fused_impl.h factor_panel): (a) panel columns into registers,
// (b) subtract the contribution of every earlier column -- per column 8 broadcast ds_read_b128
// of L[c0..c0+15][q] + one private read per slot, software-pipelined two columns per trip,
// (c) factorise the 16x16 diagonal block and the rows below it with the multipliers read out of
// the owner lanes (v_readlane), (d) one write-back.  The diagonal slot keeps d (the solve applies
// Eigen's pseudo-inverse rule).  Returns true on a zero / non-finite pivot.
//
// COOP: the panels of one factorisation are dealt round-robin to the waves of a multi-wave
// workgroup (ldlt_factor_share).  Panel p needs the finished columns of panels 0..p-1: `done`
// (an LDS word) counts the finished panels, the bulk update is applied one earlier panel at a time
// as soon as that panel is published, and the panel publishes itself at the end.  Owners of lower
// panels never wait on higher ones, so the waits always terminate.
//
// tiny: a pivot with |d| <= tiny counts as a breakdown (returns true).  0 for the ADMM factorisation
// (only an exactly zero / non-finite pivot is reported, as in Eigen's info()); for the unpivoted
// presolve of the indefinite K it is N eps max|Q_ii|: in floating point a singular leading block
// leaves a pivot of rounding-error size, not 0, and the columns divided by it are garbage -- the
// reference's pivoted LDLT either survives such a matrix or fails over to its COD, so the instance
// is handed to the fallback pass instead of being solved wrongly without a flag.
// A pivot with |d| <= DBL_MIN scales its column by 0: Eigen's LDLT leaves the column below a zero
// pivot alone (it is zero for a consistent matrix, e.g. a duplicated equality row) and solve()
// drops the pivot (pseudo-inverse rule).
template <int SLOTS, int SD, bool COOP>
__device__ __forceinline__ bool ldlt_panel(lds_f64 *M, const int N, const int lane, const int c0,
                                           volatile lds_f64 *done, const double tiny) {
  constexpr int P = kPanel, NT = SLOTS - SD;
  int rc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int row = lane + kWave * (SD + t);
    rc[t] = (row < N) ? row : N - 1;               // clamped: in-bounds reads for idle lanes
  }
  // one-wave workgroups with at most two slots: the bulk update runs on the matrix cores, in place
  constexpr bool MFMA_BULK = !COOP && SLOTS <= 2;
  if constexpr (MFMA_BULK) {
    if (c0 > 0) mfma_bulk_rt<4 * SLOTS>(M, N, lane, c0);
  }
  double T[NT][P];
#pragma unroll
  for (int jj = 0; jj < P; ++jj) {
    const int j = (c0 + jj < N) ? c0 + jj : N - 1;
    const int cb = col_base(j, N);
#pragma unroll
    for (int t = 0; t < NT; ++t) T[t][jj] = M[cb + rc[t]];   // rows above the diagonal: garbage, never stored
  }
  struct Ops { d2 w[P / 2]; double u[NT]; double dq; };
  auto fetch = [&](const int cb, const int q, Ops &o) {
    const lds_d2 *bc = reinterpret_cast<const lds_d2 *>(M + cb + c0);
#pragma unroll
    for (int t = 0; t < P / 2; ++t) o.w[t] = bc[t];          // may run past row N-1 in the last panel: unused
    o.dq = M[cb + q];
#pragma unroll
    for (int t = 0; t < NT; ++t) o.u[t] = M[cb + rc[t]];
  };
  auto apply = [&](const Ops &o) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const double u = o.u[t] * o.dq;
#pragma unroll
      for (int h = 0; h < P / 2; ++h) {
        T[t][2 * h] = fma(-u, o.w[h].x, T[t][2 * h]);
        T[t][2 * h + 1] = fma(-u, o.w[h].y, T[t][2 * h + 1]);
      }
    }
  };
  // columns [q0, q1) of earlier panels, q0 and q1 multiples of 16 (so q, q+1 are always both < q1)
  auto bulk = [&](const int q0, const int q1) {
    Ops oa, ob;
    int cb = col_base(q0, N);
    fetch(cb, q0, oa);
#pragma nounroll
    for (int q = q0; q < q1; q += 2) {
      cb += col_step(q, N);
      fetch(cb, q + 1, ob);
      apply(oa);
      cb += col_step(q + 1, N);
      fetch(cb, (q + 2 < N) ? q + 2 : q + 1, oa);            // one past the end on the last trip: unused
      apply(ob);
    }
  };
  if constexpr (COOP) {
#pragma nounroll
    for (int q0 = 0; q0 < c0; q0 += P) {
      while ((int)done[0] < q0 / P + 1) __builtin_amdgcn_s_sleep(1);   // panel q0/16 published?
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      bulk(q0, q0 + P);
    }
  } else if constexpr (!MFMA_BULK) {
    if (c0 > 0) bulk(0, c0);
  }
  const int lane0 = c0 & (kWave - 1);
  const int rel = lane - lane0;
  const int ncol = (N - c0 < P) ? N - c0 : P;
  bool bad_any = false;
  double dmine = 1.0;
  auto pinv = [](double dd) { const double r = fast_rcp(dd); return (fabs(dd) > DBL_MIN) ? r : 0.0; };
  double d = lane_bcast(T[0][0], lane0);
  double rinv = pinv(d);
#pragma unroll
  for (int k = 0; k < P; ++k) {
    // zero, tiny, infinite or NaN pivot; columns past N in a short last panel hold garbage that is
    // computed on (no branch) but neither judged nor stored
    bad_any |= (k < ncol) & (__builtin_amdgcn_class(d, 0x3 | 0x4 | 0x20 | 0x40 | 0x200) | (fabs(d) <= tiny));
    dmine = (rel == k) ? d : dmine;
    double av[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { av[t] = T[t][k]; T[t][k] = av[t] * rinv; }
    const double lsrc = T[0][k];
    double lcur = (k + 1 < P) ? lane_bcast(lsrc, lane0 + k + 1) : 0.0;
    double dn = d, rn = rinv;
#pragma unroll
    for (int jj = k + 1; jj < P; ++jj) {
      double lnext = (jj + 1 < P) ? lane_bcast(lsrc, lane0 + jj + 1) : 0.0;
#pragma unroll
      for (int t = 0; t < NT; ++t) T[t][jj] = fma(-av[t], lcur, T[t][jj]);
      // keeps each scalar next to its FMAs (otherwise they are spilled to VGPR lanes)
      asm volatile("" : "+v"(T[0][jj]), "+s"(lnext));
      if (jj == k + 1) {
        dn = lane_bcast(T[0][jj], lane0 + jj);
        rn = pinv(dn);
      }
      lcur = lnext;
    }
    d = dn; rinv = rn;
  }
  // write-back: rows below the diagonal block take all columns, the block's own rows what lies
  // strictly below the diagonal plus d
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int row = lane + kWave * (SD + t);
    if (row >= c0 + P && row < N) {
#pragma unroll
      for (int k = 0; k < P; ++k)
        if (k < ncol) M[col_base(c0 + k, N) + row] = T[t][k];
    }
  }
  if (rel >= 0 && rel < ncol) {
    const int rowd = lane + kWave * SD;
#pragma unroll
    for (int k = 0; k < P - 1; ++k)
      if (rel > k) M[col_base(c0 + k, N) + rowd] = T[0][k];
    M[col_base(rowd, N) + rowd] = dmine;
  }
  wave_sync();
  if constexpr (COOP) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the panel's columns before the count
    if (lane == 0) done[0] = (double)(c0 / P + 1);
  }
  return bad_any;
}

template <int SLOTS>
__device__ bool ldlt_factor(lds_f64 *M, int N, int lane, const double tiny) {
  static_assert(SLOTS <= 4, "a matrix that fits in LDS has N <= 256");
  bool bad = false;
#pragma nounroll
  for (int c0 = 0; c0 < N; c0 += kPanel) {
    const int sd = c0 >> 6;
    if (sd == 0) bad |= ldlt_panel<SLOTS, 0, false>(M, N, lane, c0, nullptr, tiny);
    if constexpr (SLOTS > 1) { if (sd == 1) bad |= ldlt_panel<SLOTS, 1, false>(M, N, lane, c0, nullptr, tiny); }
    if constexpr (SLOTS > 2) { if (sd == 2) bad |= ldlt_panel<SLOTS, 2, false>(M, N, lane, c0, nullptr, tiny); }
    if constexpr (SLOTS > 3) { if (sd == 3) bad |= ldlt_panel<SLOTS, 3, false>(M, N, lane, c0, nullptr, tiny); }
  }
  return bad;
}

// The panels part, part + parts, ... of a factorisation shared by `parts` waves (see ldlt_panel).
template <int SLOTS>
__device__ bool ldlt_factor_share(lds_f64 *M, int N, int lane, int part, int parts, volatile lds_f64 *done,
                                  const double tiny) {
  static_assert(SLOTS <= 4, "a matrix that fits in LDS has N <= 256");
  bool bad = false;
#pragma nounroll
  for (int c0 = part * kPanel; c0 < N; c0 += parts * kPanel) {
    const int sd = c0 >> 6;
    if (sd == 0) bad |= ldlt_panel<SLOTS, 0, true>(M, N, lane, c0, done, tiny);
    if constexpr (SLOTS > 1) { if (sd == 1) bad |= ldlt_panel<SLOTS, 1, true>(M, N, lane, c0, done, tiny); }
    if constexpr (SLOTS > 2) { if (sd == 2) bad |= ldlt_panel<SLOTS, 2, true>(M, N, lane, c0, done, tiny); }
    if constexpr (SLOTS > 3) { if (sd == 3) bad |= ldlt_panel<SLOTS, 3, true>(M, N, lane, c0, done, tiny); }
  }
  return bad;
}

// ---- the same factorisation by PANELS of kGP columns (matrix in global memory, N <= 256) ---------------------------------
// The routine below (ldlt_factor on a global pointer) eliminates one column at a time and reads, updates and writes back
// the whole trailing matrix for each: N^3 / 6 entries x 16 bytes = 29 MB of traffic per factorisation at N = 222, 60 GB for
// the 1024 QPs a chip holds at once -- 11 ms, HBM-bound, twice per cold solve (tools/phase_probe.py mid generic).  Here
// kGP = 8 columns are eliminated among themselves first (their trailing columns inside the panel updated column by column,
// the multipliers l_ik and the unscaled column entries a_ik kept in registers: 2 x 8 x SLOTS doubles), and the rest of the
// matrix then takes all eight updates in ONE pass: M[i][j] -= sum_k l_ik a_jk, with a_jk from lane j by v_readlane.  Same
// row ownership (every entry is read and written by one lane only), same pivots, same Eigen rules (a zero / non-finite
// pivot leaves its column unscaled; `tiny` as below); the sums associate differently (eight products, then one
// subtraction per panel instead of eight rounded subtractions).
constexpr int kGP = 8;
template <int SLOTS>
__device__ bool ldlt_factor_panels(double *M, const int N, const int lane, const double tiny) {
  static_assert(SLOTS <= 4, "2 x kGP x SLOTS doubles of panel state: four slots at most");
  constexpr int U = 4;
  bool bad_any = false;
  int row[SLOTS];
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) row[t] = lane + kWave * t;
  for (int k0 = 0; k0 < N; k0 += kGP) {
    const int kp = (N - k0 < kGP) ? N - k0 : kGP;         // columns in this panel
    double A[kGP][SLOTS], Lm[kGP][SLOTS];                  // unscaled column entries (rows >= k), multipliers (rows > k)
#pragma unroll
    for (int q = 0; q < kGP; ++q)
#pragma unroll
      for (int t = 0; t < SLOTS; ++t) { A[q][t] = 0.0; Lm[q][t] = 0.0; }
    // ---- the panel's own columns, one at a time; column k0 + q has taken the updates of the columns left of it --------
#pragma unroll
    for (int q = 0; q < kGP; ++q) {
      if (q < kp) {
        const int k = k0 + q;
        const int ck = col_base(k, N);
        const int tk = k >> 6, lk = k & (kWave - 1);
        double dsel = 0.0;
        double cp[kGP];                                    // a_kp of row k for the panel's earlier columns p < q
#pragma unroll
        for (int p = 0; p < q; ++p) {
          double c = 0.0;
#pragma unroll
          for (int t2 = 0; t2 < SLOTS; ++t2) {
            const double b = lane_bcast(A[p][t2], lk);
            c = (t2 == tk) ? b : c;
          }
          cp[p] = c;
        }
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
          double v = (row[t] >= k && row[t] < N) ? M[ck + row[t]] : 0.0;
#pragma unroll
          for (int p = 0; p < q; ++p) v = fma(-Lm[p][t], cp[p], v);
          A[q][t] = (row[t] >= k) ? v : 0.0;
          dsel = (t == tk) ? A[q][t] : dsel;
        }
        const double d = lane_bcast(dsel, lk);
        const bool bad = !(fabs(d) > 0.0) || !(fabs(d) <= DBL_MAX);
        bad_any |= bad || !(fabs(d) > tiny);
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
          const double v = (row[t] > k) ? A[q][t] : 0.0;
          Lm[q][t] = bad ? v : v / d;                      // Eigen: A21 /= pivot only when the pivot is non-zero
          if (row[t] > k && row[t] < N) M[ck + row[t]] = Lm[q][t];
          else if (row[t] == k) M[ck + row[t]] = A[q][t];  // the pivot itself (it has taken the panel's updates)
        }
      }
    }
    // ---- the columns right of the panel take its kp updates in one pass ------------------------------------------------
    const int kend = k0 + kp;
#pragma unroll
    for (int tj = 0; tj < SLOTS; ++tj) {
      const int jbase = kWave * tj;
      if (jbase + kWave > kend && jbase < N) {             // wave-uniform: this slot holds columns >= kend
        const int jl0 = (kend > jbase) ? kend - jbase : 0;
        const int jl1 = (N - jbase < kWave) ? N - jbase : kWave;
        for (int jl = jl0; jl < jl1; jl += U) {
          double v[U][SLOTS];
          int cj[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int jlu = (jl + u < jl1) ? jl + u : jl1 - 1;
            cj[u] = col_base(jbase + jlu, N);
#pragma unroll
            for (int t2 = tj; t2 < SLOTS; ++t2) {
              const int j = jbase + jlu;
              v[u][t2] = (row[t2] >= j && row[t2] < N) ? M[cj[u] + row[t2]] : 0.0;
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int jlu = (jl + u < jl1) ? jl + u : jl1 - 1;
#pragma unroll
            for (int q = 0; q < kGP; ++q) {
              const double c = (jl + u < jl1) ? lane_bcast(A[q][tj], jlu) : 0.0;      // a_jq of row j (zero beyond the panel's kp)
#pragma unroll
              for (int t2 = tj; t2 < SLOTS; ++t2) v[u][t2] = fma(-Lm[q][t2], c, v[u][t2]);
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int j = jbase + jl + u;
            if (jl + u < jl1) {
#pragma unroll
              for (int t2 = tj; t2 < SLOTS; ++t2)
                if (row[t2] >= j && row[t2] < N) M[cj[u] + row[t2]] = v[u][t2];
            }
          }
        }
      }
    }
  }
  return bad_any;
}


// Unpivoted LDL^T in place (right-looking), natural order: the variant for matrices that do not
// fit in LDS (global workspace, N > 198).  Returns true if a pivot was zero, not finite or, in
// magnitude, <= tiny (the situation in which Eigen's info() != Success can arise, fcc_qp.cpp:164;
// see ldlt_panel for `tiny`).
//
// Row ownership is static: lane l owns rows l, l + 64, ... of the matrix for the whole
// factorisation, so every entry M[i][j] is read, updated and written back by ONE lane only, and
// what the other lanes need of column k -- the unscaled a_jk, j > k -- travels through registers
// (v_readlane), not through memory.  (The first version moved the lane <-> row map with k, i = k + 1 +
// lane, handing every entry from lane to lane through global memory once per step under a
// wavefront-scope fence; on some MI355X boxes, not on others, that returned sporadically wrong
// factors for N > 256 -- found by tests/test_gpu_reference_quirks.py, cause not established.)
// The trailing update is issued as groups of loads, FMAs and stores (U columns x the live slots):
// left as read-modify-write statements it compiles to one load / wait / store round trip each.
template <int SLOTS>
__device__ bool ldlt_factor(double *M, int N, int lane, const double tiny) {
  if constexpr (SLOTS <= 4) return ldlt_factor_panels<SLOTS>(M, N, lane, tiny);     // (round 5: rank-8 updates, see there)
  constexpr int U = 4;
  bool bad_any = false;
  int row[SLOTS];
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) row[t] = lane + kWave * t;
  for (int k = 0; k < N; ++k) {
    const int ck = col_base(k, N);
    const int tk = k >> 6, lk = k & (kWave - 1);          // owner slot / lane of row k
    double a[SLOTS];                                       // column k, rows >= k, unscaled
    double dsel = 0.0;
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) {
      a[t] = (row[t] >= k && row[t] < N) ? M[ck + row[t]] : 0.0;
      dsel = (t == tk) ? a[t] : dsel;
    }
    const double d = lane_bcast(dsel, lk);
    const bool bad = !(fabs(d) > 0.0) || !(fabs(d) <= DBL_MAX);
    bad_any |= bad || !(fabs(d) > tiny);
    double l[SLOTS];
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) {
      const double v = (row[t] > k) ? a[t] : 0.0;
      l[t] = bad ? v : v / d;                              // Eigen: A21 /= pivot only when the pivot is non-zero
    }
    // trailing update: M[i][j] -= l_i a_jk for k < j <= i, column by column, U columns per group
#pragma unroll
    for (int tj = 0; tj < SLOTS; ++tj) {
      const int jbase = kWave * tj;
      if (jbase + kWave - 1 > k && jbase < N) {            // wave-uniform: this slot holds columns > k
        const int jl0 = (k + 1 > jbase) ? k + 1 - jbase : 0;
        const int jl1 = (N - jbase < kWave) ? N - jbase : kWave;
        for (int jl = jl0; jl < jl1; jl += U) {
          double c[U], v[U][SLOTS];
          int cj[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int jlu = (jl + u < jl1) ? jl + u : jl1 - 1;         // tail: repeat the last column ...
            c[u] = (jl + u < jl1) ? lane_bcast(a[tj], jlu) : 0.0;      // ... with a zero multiplier
            cj[u] = col_base(jbase + jlu, N);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int j = jbase + ((jl + u < jl1) ? jl + u : jl1 - 1);
#pragma unroll
            for (int t2 = tj; t2 < SLOTS; ++t2) v[u][t2] = (row[t2] >= j && row[t2] < N) ? M[cj[u] + row[t2]] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t2 = tj; t2 < SLOTS; ++t2) v[u][t2] -= l[t2] * c[u];
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int j = jbase + jl + u;
            if (jl + u < jl1) {
#pragma unroll
              for (int t2 = tj; t2 < SLOTS; ++t2)
                if (row[t2] >= j && row[t2] < N) M[cj[u] + row[t2]] = v[u][t2];
            }
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < SLOTS; ++t)
      if (row[t] > k && row[t] < N) M[ck + row[t]] = l[t];
  }
  return bad_any;
}

// x := (L D L^T)^-1 x with Eigen's pseudo-inverse rule on D (|d| > DBL_MIN else 0).
// x is lane-distributed: lane owns entries lane + 64 t.  Column-oriented substitution; the only
// dependent chain is v_readlane(x_k) -> fma: the L operands of four steps are fetched ahead of
// it, unconditionally (clamped rows, masked by a select) and through running column offsets.
template <int SLOTS, class MP>
__device__ void ldlt_solve(MP M, int N, int lane, double (&x)[SLOTS]) {
  constexpr int U = 4;
  // per-slot row indices: rlo = row or -1 for an idle lane (never "> k"), rhi = row or INT_MAX
  // (never "< k"); rowc = clamped row for in-bounds operand reads
  int rowc[SLOTS], cbl[SLOTS], rlo[SLOTS], rhi[SLOTS];
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) {
    const int i = lane + kWave * t;
    rowc[t] = (i < N) ? i : N - 1;
    rlo[t] = (i < N) ? i : -1;
    rhi[t] = (i < N) ? i : 0x7fffffff;
    cbl[t] = col_base(rowc[t], N);
  }
  // ---- forward, unit lower: x_i -= L[i][k] x_k, i > k; x_k lives in slot s, lane k - 64 s -------
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int kend = (N - s * kWave < kWave) ? N - s * kWave : kWave;   // columns in this slot
    if (kend > 0) {
      int cb = col_base(s * kWave, N);
      const int kfull = kend & ~(U - 1);
      for (int kk0 = 0; kk0 < kfull; kk0 += U) {   // groups of U columns, operands fetched ahead
        double a[U][SLOTS];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = s * kWave + kk0 + u;
#pragma unroll
          for (int t = s; t < SLOTS; ++t) {
            const double l = M[cb + rowc[t]];
            // only the diagonal slot has rows <= k; slots below it are unmasked (their idle lanes
            // accumulate garbage that the D step zeroes)
            a[u][t] = (t == s) ? ((rlo[t] > k) ? l : 0.0) : l;
          }
          cb += col_step(k, N);                    // col_base(k + 1) - col_base(k)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const double xk = lane_bcast(x[s], kk0 + u);
#pragma unroll
          for (int t = s; t < SLOTS; ++t) x[t] = fma(-a[u][t], xk, x[t]);
        }
      }
      for (int kk = kfull; kk < kend; ++kk) {      // remainder, one column at a time
        const int k = s * kWave + kk;
        const double xk = lane_bcast(x[s], kk);
#pragma unroll
        for (int t = s; t < SLOTS; ++t) {
          const double l = M[cb + rowc[t]];
          x[t] = fma(-((t == s) ? ((rlo[t] > k) ? l : 0.0) : l), xk, x[t]);
        }
        cb += col_step(k, N);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) {
    const double d = M[cbl[t] + rowc[t]];
    x[t] = (rlo[t] >= 0 && fabs(d) > DBL_MIN) ? x[t] / d : 0.0;
  }
  // ---- backward with L^T: x_i -= L[k][i] x_k, i < k (row k of L: a gather along lane columns) --
#pragma unroll
  for (int s = SLOTS - 1; s >= 0; --s) {
    const int kend = (N - s * kWave < kWave) ? N - s * kWave : kWave;
    if (kend > 0) {
      const int rem = kend & (U - 1);
      for (int kk = kend - 1; kk >= kend - rem; --kk) {   // remainder first (highest columns)
        const int k = s * kWave + kk;
        const double xk = lane_bcast(x[s], kk);
#pragma unroll
        for (int t = 0; t <= s; ++t) {
          const double l = M[cbl[t] + k];
          x[t] = fma(-((t == s) ? ((rhi[t] < k) ? l : 0.0) : l), xk, x[t]);   // slots above: all rows < k
        }
      }
      for (int kk0 = kend - rem - 1; kk0 >= 0; kk0 -= U) {
        double a[U][SLOTS];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = s * kWave + kk0 - u;
#pragma unroll
          for (int t = 0; t <= s; ++t) {
            const double l = M[cbl[t] + k];
            a[u][t] = (t == s) ? ((rhi[t] < k) ? l : 0.0) : l;
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const double xk = lane_bcast(x[s], kk0 - u);
#pragma unroll
          for (int t = 0; t <= s; ++t) x[t] = fma(-a[u][t], xk, x[t]);
        }
      }
    }
  }
}

// R right-hand sides at once through the same factorisation (matrix in LDS): X[r][t] is entry
// lane + 64 t of right-hand side r.  The L operands of a step are loaded once and applied to all R
// vectors (per vector: one v_readlane pair + one FMA per live slot), which is what makes building
// G = (K_rho^-1)[0:n,0:n] affordable (see the ADMM loop).  kstart: every right-hand side is zero
// above row kstart, so the forward sweep starts there.
template <int SLOTS, int R>
__device__ void ldlt_solve_multi(const lds_f64 *M, const int N, const int lane, double (&X)[R][SLOTS],
                                 const int kstart) {
  int rowc[SLOTS], cbl[SLOTS], rlo[SLOTS], rhi[SLOTS];
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) {
    const int i = lane + kWave * t;
    rowc[t] = (i < N) ? i : N - 1;
    rlo[t] = (i < N) ? i : -1;
    rhi[t] = (i < N) ? i : 0x7fffffff;
    cbl[t] = col_base(rowc[t], N);
  }
  // ---- forward ----------------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int kend = (N - s * kWave < kWave) ? N - s * kWave : kWave;
    int kk = kstart - s * kWave;
    kk = (kk < 0) ? 0 : kk;
    if (kk < kend) {
      int cb = col_base(s * kWave + kk, N);
      double an[SLOTS];
#pragma unroll
      for (int t = s; t < SLOTS; ++t) an[t] = M[cb + rowc[t]];
      for (; kk < kend; ++kk) {
        const int k = s * kWave + kk;
        double ac[SLOTS];
#pragma unroll
        for (int t = s; t < SLOTS; ++t) ac[t] = (t == s) ? ((rlo[t] > k) ? an[t] : 0.0) : an[t];
        cb += col_step(k, N);
        const int cbn = (kk + 1 < kend) ? cb : cb - col_step(k, N);   // next step's column (or a re-read)
#pragma unroll
        for (int t = s; t < SLOTS; ++t) an[t] = M[cbn + rowc[t]];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const double xk = lane_bcast(X[r][s], kk);
#pragma unroll
          for (int t = s; t < SLOTS; ++t) X[r][t] = fma(-ac[t], xk, X[r][t]);
        }
      }
    }
  }
  // ---- D^-1 with the pseudo-inverse rule --------------------------------------------------------
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) {
    const double d = M[cbl[t] + rowc[t]];
    const double rinv = (rlo[t] >= 0 && fabs(d) > DBL_MIN) ? 1.0 / d : 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) X[r][t] *= rinv;
  }
  // ---- backward ---------------------------------------------------------------------------------
#pragma unroll
  for (int s = SLOTS - 1; s >= 0; --s) {
    const int kend = (N - s * kWave < kWave) ? N - s * kWave : kWave;
    if (kend > 0) {
      double an[SLOTS];
#pragma unroll
      for (int t = 0; t <= s; ++t) an[t] = M[cbl[t] + s * kWave + kend - 1];
      for (int kk = kend - 1; kk >= 0; --kk) {
        const int k = s * kWave + kk;
        double ac[SLOTS];
#pragma unroll
        for (int t = 0; t <= s; ++t) ac[t] = (t == s) ? ((rhi[t] < k) ? an[t] : 0.0) : an[t];
        const int kn = (kk > 0) ? k - 1 : k;
#pragma unroll
        for (int t = 0; t <= s; ++t) an[t] = M[cbl[t] + kn];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const double xk = lane_bcast(X[r][s], kk);
#pragma unroll
          for (int t = 0; t <= s; ++t) X[r][t] = fma(-ac[t], xk, X[r][t]);
        }
      }
    }
  }
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
  if (z2 > 0.0) {  // Eigen normalize() leaves a zero vector untouched (f_z == 0 quirk)
    const double nr = sqrt(z2);
    rx /= nr; ry /= nr; rz /= nr;
  }
  const double dot = rx * fx + ry * fy + rz * fz;                  // :23
  ox = dot * rx; oy = dot * ry; oz = dot * rz;
}

// Assembly of the packed lower KKT [[Q + rho I, .],[A, 0]] (fcc_qp.cpp:146-150), the share of wave
// `part` of `parts`.  Row-wise: lanes run along a row of Q (then A), so global reads are coalesced,
// and the loads of eight rows are issued before the first store.  Only the lower triangle of Q is
// kept (what Eigen's LDLT reads).  Loads are unconditional on clamped addresses; the mask only
// gates stores.  MP: LDS or global pointer to the matrix.
template <class TIN, class MP>
__device__ __forceinline__ void assemble_share(MP M, const TIN *Q, const TIN *A, const int n, const int N,
                                               const double rho, const int lane, const int part,
                                               const int parts) {
  constexpr int R = 8;
  const int jc_hi = (n > 0) ? n - 1 : 0;
  if (n > 0) {
    for (int i0 = R * part; i0 < N; i0 += R * parts) {
      const int ilast = (i0 + R - 1 < N) ? i0 + R - 1 : N - 1;
      const int jend = (ilast < n) ? ilast + 1 : n;          // columns needed by this row group
      for (int j0 = 0; j0 < jend; j0 += kWave) {
        const int j = j0 + lane;
        const int jc = (j < jc_hi) ? j : jc_hi;
        double v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int i = (i0 + r < N) ? i0 + r : N - 1;
          const TIN *src = (i < n) ? Q + (size_t)i * n : A + (size_t)(i - n) * n;
          v[r] = (double)src[jc];
        }
        const int cj = col_base(jc, N);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int i = i0 + r;
          if (i < N && j < n && j <= i) M[cj + i] = v[r] + ((i == j) ? rho : 0.0);
        }
      }
    }
  }
  for (int j = n + part; j < N; j += parts) {                  // zero block (rows and columns >= n)
    const int cj = col_base(j, N);
    for (int i = j + lane; i < N; i += kWave) M[cj + i] = 0.0;
  }
}

// ---- the operator iteration's mat-vec: acc += (rho G)[rows of this lane][columns of chunks [cb, ce)] w
// rho*G sits column-major at M[0 .. n*n), w zero-padded at M[w_off ..) (g_w_off / g_w_len), both in
// LDS.  A chunk is CH = 3 column pairs.  Left alone the compiler serialises this loop (one load, one
// wait, one FMA, same register): the operands of a whole chunk are loaded into distinct registers
// first, and the next chunk's loads are issued before this chunk's FMAs.  Chunks up to ce + 1 are
// fetched (never applied): w is zero-filled two chunks past the last real one and column indices
// are clamped.
constexpr int kMvPairs = 3;
__device__ __forceinline__ int g_w_off(int n) { return (n * n + 1) & ~1; }
__device__ __forceinline__ int g_chunks(int n) { return ((n + 1) / 2 + kMvPairs - 1) / kMvPairs; }
__device__ __forceinline__ int g_w_len(int n) { return 2 * kMvPairs * (g_chunks(n) + 2); }
__device__ __forceinline__ int g_part_off(int n) { return g_w_off(n) + g_w_len(n); }   // [4][128] partial sums

template <int GS>
__device__ __forceinline__ void matvec_range(const lds_f64 *M, const int n, const int lane, const int cbeg,
                                             const int cend, double (&acc0)[GS], double (&acc1)[GS]) {
  constexpr int CH = kMvPairs;
  int rc2[GS];
#pragma unroll
  for (int t = 0; t < GS; ++t) {
    const int i = lane + kWave * t;
    rc2[t] = (i < n) ? i : n - 1;
  }
  const lds_d2 *w2 = reinterpret_cast<const lds_d2 *>(M + g_w_off(n));
  struct Chunk { d2 w[CH]; double ga[CH][GS], gb[CH][GS]; };
  auto fetch = [&](const int c, Chunk &o) {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int jp = c * CH + u;
      o.w[u] = w2[jp];
      const int j0 = (2 * jp < n) ? 2 * jp : n - 1, j1 = (2 * jp + 1 < n) ? 2 * jp + 1 : n - 1;
#pragma unroll
      for (int t = 0; t < GS; ++t) {
        o.ga[u][t] = M[j0 * n + rc2[t]];
        o.gb[u][t] = M[j1 * n + rc2[t]];
      }
    }
  };
  auto apply = [&](const Chunk &o) {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
#pragma unroll
      for (int t = 0; t < GS; ++t) {
        acc0[t] = fma(o.ga[u][t], o.w[u].x, acc0[t]);
        acc1[t] = fma(o.gb[u][t], o.w[u].y, acc1[t]);
      }
    }
  };
  if (cbeg >= cend) return;
  Chunk ca, cb;
  fetch(cbeg, ca);
#pragma nounroll
  for (int c = cbeg; c < cend; c += 2) {
    fetch(c + 1, cb);
    __builtin_amdgcn_sched_barrier(0);
    apply(ca);
    __builtin_amdgcn_sched_barrier(0);
    fetch(c + 2, ca);
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < cend) apply(cb);                     // the range may hold an odd number of chunks
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Shape tags: DynShape reads the sizes from the kernel arguments (any shape); FixShape bakes one
// shape in, so that column bases, loop bounds and edge masks fold into constants (what took the
// fused kernel's factorisation from 84 to 58 us).  The large benchmark shape has an instantiation.
struct DynShape { static constexpr bool fixed = false; static constexpr int n = 0, m = 0, nc = 0, s = 0; };
template <int N_VARS, int N_EQ, int NC, int LC_START>
struct FixShape { static constexpr bool fixed = true; static constexpr int n = N_VARS, m = N_EQ, nc = NC, s = LC_START; };
using LargeShape = FixShape<114, 60, 24, 66>;

// ---- building rho*G with several waves ----------------------------------------------------------
// Shapes that need three or four register slots own a whole CU's LDS (one workgroup per CU), so
// their workgroup is launched with four waves, one per SIMD: wave 0 runs the solver, waves 1-3
// wait at a barrier and are only called on for the one step that parallelises trivially -- the
// n + 1 substitutions that build G, split by batches of RB right-hand sides -- and for the copy of
// G into LDS.  Commands go through one LDS word.
template <int SLOTS, bool MAT_IN_LDS>
constexpr int waves_for() { return (MAT_IN_LDS && SLOTS >= 3 && SLOTS <= 4) ? 4 : 1; }
template <int SLOTS>
constexpr int g_batch() { return (SLOTS <= 3) ? 16 : 8; }   // right-hand sides per substitution batch
constexpr int kCmdBuild = 1, kCmdDone = 2, kCmdAssemble = 3, kCmdFactor = 4, kCmdMatvec = 5;
// command words (doubles after the panel slack): [0] command, [1] rho, [2] finished-panel count of a
// shared factorisation, [3 + wave] that wave's bad-pivot flag
constexpr int kCtrlWords = 8;

__device__ __forceinline__ lds_f64 *ctrl_word(double *lds, int N, int nc) {
  return (lds_f64 *)lds + packed_doubles(N) + nc + 2 + kPanel;
}

// One batch: right-hand side index j = j0 + r - 1; j == -1 is [-b; b_eq] (WITH_RHS only, batch 0),
// j >= 0 the unit vector e_j.  Stores rho * (solution rows < n) as column j of the staging matrix.
template <int SLOTS, int RB, bool WITH_RHS>
__device__ __forceinline__ void g_build_batch(const lds_f64 *M, const int N, const int n, const int lane,
                                              const int j0, const double rho, double *stage,
                                              const double *vb, const double *vtail, double *xc) {
  constexpr int GS = (SLOTS < 2) ? SLOTS : 2;
  double X[RB][SLOTS];
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int j = j0 + r - 1;
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) {
      const int i = lane + kWave * t;
      double v = (i == j && j < n) ? 1.0 : 0.0;
      if constexpr (WITH_RHS) { if (j < 0) v = (i < n) ? -vb[t] : vtail[t]; }
      X[r][t] = v;
    }
  }
  ldlt_solve_multi<SLOTS, RB>(M, N, lane, X, (j0 == 0) ? 0 : j0 - 1);
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int j = j0 + r - 1;
#pragma unroll
    for (int t = 0; t < GS; ++t) {
      const int i = lane + kWave * t;
      if constexpr (WITH_RHS) { if (j < 0) xc[t] = (i < n) ? X[r][t] : 0.0; }
      if (j >= 0 && j < n && i < n) stage[(size_t)j * n + i] = rho * X[r][t];
    }
  }
}

// ---- G build on the matrix cores (four-wave workgroups) ---------------------------------------------
// Same scheme as p3_mfma of the fused kernel (fccqp_fused_impl.h), with runtime N: the N x 128 block
// X = [e_0 .. e_{n-1} | -b; b_eq | 0 ..] is split by column blocks of 16 over the four waves (two
// each); a wave keeps its N x 32 slice in MFMA accumulator tiles and runs the blocked forward /
// D^-1 / backward substitution on it: X_I -= L_IK X_K with A[i][k] = L[16I+i][16K+4r+k] one
// ds_read_b64 per lane and B = accumulator register r of tile X_K as it stands.  The 16 x 16 diagonal
// blocks of L are inverted in place first (all waves, one column per lane).  NRB = ceil(N / 16) is
// a template parameter so that the tiles have compile-time register indices.
// column j of the inverse of diagonal block blk, for pair p = 16 blk + j (MULTI: all four waves of
// a workgroup take part, else the one wave calls it once per 64 pairs)
template <bool MULTI>
__device__ __forceinline__ void invert_diag_blocks_share(lds_f64 *M, const int N, const int p) {
  const int nrb = (N + 15) / 16;
  const int blk = p >> 4, j = p & 15, r0 = 16 * blk;
  const bool live = blk < nrb;
  double y[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) y[i] = (i == j) ? 1.0 : 0.0;
  const lds_f64 *colp[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) {
    const int c = (live && r0 + k < N) ? r0 + k : 0;
    colp[k] = M + col_base(c, N) + (live ? r0 : 0);
  }
  const int nrow = live ? N - r0 : 0;
#pragma unroll
  for (int i = 1; i < 16; ++i) {
    double l[15];
#pragma unroll
    for (int k = 0; k < i; ++k) l[k] = colp[k][i];
#pragma unroll
    for (int k = 0; k < i; ++k) y[i] = fma(-((i < nrow) ? l[k] : 0.0), y[k], y[i]);
  }
  // every lane (of every wave) has read its block before any block is overwritten
  if constexpr (MULTI) workgroup_sync(); else wave_sync();
#pragma unroll
  for (int i = 1; i < 16; ++i)
    if (live && i > j && r0 + i < N) M[col_base(r0 + j, N) + r0 + i] = y[i];
  if constexpr (MULTI) workgroup_sync(); else wave_sync();
}

// A real call (noinline): inlined into the solver wave, its ~400 live registers push the solver's own
// state into scratch for the whole kernel (measured: 2.2 KB of scratch per lane, every phase slower).
// Arguments of a real call arrive in VGPRs; v_readfirstlane makes the wave-uniform ones scalar again.
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T *uni_p(T *p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)u);
  const unsigned hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double uni_d(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// RX: where [-b; b_eq] comes from and x_c goes to: LDS words (four-wave workgroups) or global memory
// (one-wave workgroups, whose LDS has no room to spare).  `wave` picks the 32 right-hand-side columns.
__device__ __forceinline__ lds_f64 *uni_ptr(lds_f64 *p) { return (lds_f64 *)(size_t)(unsigned)uni_i((int)(unsigned)(size_t)p); }
__device__ __forceinline__ double *uni_ptr(double *p) { return uni_p(p); }
template <int NRB, class RX>
__device__ __noinline__ void g_build_mfma(lds_f64 *M_in, const int N_in, const int n_in, const int lane,
                                          const int wave_in, const double rho_in, double *stage_in,
                                          RX *rhs_in, RX *xc_in) {
  constexpr int NCB = 2, NT = 4 * NRB, TC = 8;         // A operands are fetched TC k-steps at a time
  const int N = uni_i(N_in), n = uni_i(n_in), wave = uni_i(wave_in);
  const double rho = uni_d(rho_in);
  lds_f64 *const M = (lds_f64 *)(size_t)(unsigned)uni_i((int)(unsigned)(size_t)M_in);
  const RX *const rhs_s = uni_ptr(rhs_in);
  RX *const xc_s = uni_ptr(xc_in);
  double *const stage = uni_p(stage_in);
  const int col = lane & 15, grp = lane >> 4;
  d4 X[NRB][NCB];
#pragma unroll
  for (int I = 0; I < NRB; ++I) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * I + grp + 4 * r;
      const double rhs = (row < N) ? rhs_s[row < N ? row : 0] : 0.0;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int c = 32 * wave + 16 * cb + col;
        X[I][cb][r] = (c < n) ? ((row == c) ? 1.0 : 0.0) : ((c == n) ? rhs : 0.0);
      }
    }
  }
  // forward operand L[16I + col][4t + grp]: column base recomputed per k-step (a table of 4 NRB
  // pointers per lane is what pushed the inlined version over the register file)
  auto fwd_at = [&](const int t, const int I) -> double {
    const int c = (4 * t + grp < N) ? 4 * t + grp : N - 1;
    return M[col_base(c, N) + col + 16 * I];            // rows past N-1 land in allocated LDS
  };
  const lds_f64 *bwd[NRB];
#pragma unroll
  for (int I = 0; I < NRB; ++I) bwd[I] = M + col_base((16 * I + col < N) ? 16 * I + col : N - 1, N) + grp;
  // ---- forward ------------------------------------------------------------------------------------
#pragma unroll
  for (int I = 0; I < NRB; ++I) {
    d4 T[NCB];                                          // the diagonal product accumulates separately
#pragma unroll
    for (int ch = 0; ch < (NT + TC - 1) / TC; ++ch) {
      const int t0 = ch * TC;
      double aop[TC];
#pragma unroll
      for (int u = 0; u < TC; ++u) {
        const int t = t0 + u;
        if (t < 4 * I + 4) {
          const double v = fwd_at(t, I);
          // 16 (NRB - 1) < N <= 16 NRB: only the last row block / its k-steps can leave the matrix
          bool keep = true;
          if (I == NRB - 1) keep = keep && (16 * I + col < N);
          if (t >= 4 * (NRB - 1)) keep = keep && (4 * t + grp < N);
          if (t >= 4 * I) keep = keep && (col > 4 * (t - 4 * I) + grp);   // inverted diagonal block: strict lower
          aop[u] = (I == NRB - 1 || t >= 4 * (NRB - 1) || t >= 4 * I) ? (keep ? ((t >= 4 * I) ? v : -v) : 0.0) : -v;
        }
      }
#pragma unroll
      for (int u = 0; u < TC; ++u) {
        const int t = t0 + u;
        if (t < 4 * I) {
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
            X[I][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[u], X[t / 4][cb][t % 4], X[I][cb], 0, 0, 0);
        } else if (t < 4 * I + 4) {
          if (t == 4 * I) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) T[cb] = X[I][cb];
          }
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
            T[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[u], X[I][cb][t - 4 * I], T[cb], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) X[I][cb] = T[cb];
  }
  // ---- D^-1, Eigen's pseudo-inverse rule (the diagonal slots hold d) --------------------------------
#pragma unroll
  for (int I = 0; I < NRB; ++I) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * I + grp + 4 * r;
      const int rc = (row < N) ? row : 0;
      const double d = M[col_base(rc, N) + rc];
      const double rd = (row < N && fabs(d) > DBL_MIN) ? 1.0 / d : 0.0;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) X[I][cb][r] *= rd;
    }
  }
  // ---- backward: the diagonal product first needs all K > I applied, so it comes last -----------------
#pragma unroll
  for (int Ir = 0; Ir < NRB; ++Ir) {
    const int I = NRB - 1 - Ir;
#pragma unroll
    for (int ch = 0; ch < (NT + TC - 1) / TC; ++ch) {
      const int t0 = 4 * I + 4 + ch * TC;
      double aop[TC];
#pragma unroll
      for (int u = 0; u < TC; ++u) {
        const int t = t0 + u;
        if (t < NT) {
          const double v = bwd[I][4 * t];               // L[4t + grp][16I + col]
          bool keep = true;
          if (I == NRB - 1) keep = keep && (16 * I + col < N);
          if (t >= 4 * (NRB - 1)) keep = keep && (4 * t + grp < N);
          aop[u] = (I == NRB - 1 || t >= 4 * (NRB - 1)) ? (keep ? -v : 0.0) : -v;
        }
      }
#pragma unroll
      for (int u = 0; u < TC; ++u) {
        const int t = t0 + u;
        if (t < NT) {
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
            X[I][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[u], X[t / 4][cb][t % 4], X[I][cb], 0, 0, 0);
        }
      }
    }
    double dop[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = 4 * I + r;
      const double v = bwd[I][4 * t];
      bool keep = (4 * r + grp > col);
      if (I == NRB - 1) keep = keep && (16 * I + col < N) && (4 * t + grp < N);
      dop[r] = keep ? v : 0.0;
    }
    d4 T[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) T[cb] = X[I][cb];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
        T[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(dop[r], X[I][cb][r], T[cb], 0, 0, 0);
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) X[I][cb] = T[cb];
  }
  // ---- rho * G columns to the staging matrix, x_c to LDS ---------------------------------------------
#pragma unroll
  for (int I = 0; I < NRB; ++I) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * I + grp + 4 * r;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int c = 32 * wave + 16 * cb + col;
        if (row < n && c < n) stage[(size_t)c * n + row] = rho * X[I][cb][r];
        if (row < n && c == n) xc_s[row] = X[I][cb][r];
      }
    }
  }
}

// the whole cooperative build between barriers A and B of the kCmdBuild command; false if this
// shape is not covered (then the caller falls back to the readlane batches)
__device__ __forceinline__ bool g_build_mfma_supported(const int N, const int n) {
  const int nrb = (N + 15) / 16;
  return n + 1 <= 128 && nrb >= 9 && nrb <= 13;
}
template <int SLOTS>
__device__ __forceinline__ void g_build_mfma_share(lds_f64 *M, const int N, const int n, const int lane,
                                                   const int wave, const double rho, double *stage,
                                                   lds_f64 *scratch) {
  invert_diag_blocks_share<true>(M, N, wave * kWave + lane);
  lds_f64 *rhs_s = scratch;
  lds_f64 *xc_s = scratch + N;
  const int nrb = (N + 15) / 16;
  if constexpr (SLOTS == 3) {
    if (nrb == 9) g_build_mfma<9, lds_f64>(M, N, n, lane, wave, rho, stage, rhs_s, xc_s);
    else if (nrb == 10) g_build_mfma<10, lds_f64>(M, N, n, lane, wave, rho, stage, rhs_s, xc_s);
    else if (nrb == 11) g_build_mfma<11, lds_f64>(M, N, n, lane, wave, rho, stage, rhs_s, xc_s);
    else if (nrb == 12) g_build_mfma<12, lds_f64>(M, N, n, lane, wave, rho, stage, rhs_s, xc_s);
  } else {
    if (nrb == 13) g_build_mfma<13, lds_f64>(M, N, n, lane, wave, rho, stage, rhs_s, xc_s);
  }
}

// One-wave workgroups (N <= 128): the same build, the wave taking the 32-column slices one after
// the other; [-b; b_eq] and x_c travel through `gs` (2 N doubles of global memory behind the staging
// matrix).
__device__ __forceinline__ bool g_build_mfma_single_supported(const int N, const int n, const long long stride) {
  const int nrb = (N + 15) / 16;
  return nrb >= 2 && nrb <= 8 && n + 1 <= 128 && stride >= (long long)n * n + 2 * N;
}
__device__ __forceinline__ void g_build_mfma_single(lds_f64 *M, const int N, const int n, const int lane,
                                                    const double rho, double *stage, double *gs) {
  const int nrb = (N + 15) / 16;
  for (int p0 = 0; p0 < 16 * nrb; p0 += kWave) invert_diag_blocks_share<false>(M, N, p0 + lane);
  const int parts = (n + 1 + 31) / 32;
  for (int part = 0; part < parts; ++part) {
    switch (nrb) {
      case 2: g_build_mfma<2, double>(M, N, n, lane, part, rho, stage, gs, gs + N); break;
      case 3: g_build_mfma<3, double>(M, N, n, lane, part, rho, stage, gs, gs + N); break;
      case 4: g_build_mfma<4, double>(M, N, n, lane, part, rho, stage, gs, gs + N); break;
      case 5: g_build_mfma<5, double>(M, N, n, lane, part, rho, stage, gs, gs + N); break;
      case 6: g_build_mfma<6, double>(M, N, n, lane, part, rho, stage, gs, gs + N); break;
      case 7: g_build_mfma<7, double>(M, N, n, lane, part, rho, stage, gs, gs + N); break;
      default: g_build_mfma<8, double>(M, N, n, lane, part, rho, stage, gs, gs + N); break;
    }
  }
}

// ---- operator iterations of a four-wave workgroup: rho G in the helpers' REGISTERS --------------------------
// x = x_c + (rho G) w, n <= 128.  The three helper waves each own kMvBlocks blocks of 16 columns of
// rho G (helper h: blocks kMvBlocks (h - 1) ...), for both row slots (rows lane, 64 + lane): 96
// doubles per lane, loaded once from the staged G when it is built.  Per iteration a helper reads
// its share of w as one ds_read_b64 per block (Wq[lane] = w[16 q + (lane & 15)]: every 16-lane row
// takes a copy) and the FMA fetches its operand from lane l of the row itself (v_fmac_f64 with DPP
// row_newbcast:l) -- no LDS traffic for G or w inside the product, where the first version streamed
// its quarter of G (26 KB per wave) and all of w as broadcasts out of LDS on every iteration
// (2.2 us per iteration at n = 114).  The solver wave only sums the three partial results.
constexpr int kMvBlocks = 3;                               // 3 helpers x 3 blocks x 16 = 144 >= 128 columns
constexpr int kMvWLen = 16 * kMvBlocks * 3;                // doubles of w the helpers may read (zero padded)
template <int IDX>
__device__ __forceinline__ void dpp_mv_rows(double (&acc)[2][2], const double (&Wq)[kMvBlocks],
                                            const double (&G)[kMvBlocks][16][2]) {
  if constexpr (IDX < kMvBlocks * 16) {
    constexpr int qi = IDX / 16, l = IDX % 16;
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc[0][IDX & 1]) : "v"(Wq[qi]), "v"(G[qi][l][0]), "n"(l));
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc[1][IDX & 1]) : "v"(Wq[qi]), "v"(G[qi][l][1]), "n"(l));
    dpp_mv_rows<IDX + 1>(acc, Wq, G);
  }
}
// where w and the partial sums live once G itself is no longer kept in LDS
__device__ __forceinline__ bool mv_regs_ok(const int n, const int npack) {
  return n <= 2 * kWave && g_w_off(n) + kMvWLen + 4 * 2 * kWave <= npack;
}

// ---- presolve out of the K_rho factorisation ------------------------------------------------------------
// x0 = (K^-1 [-b; b_eq]).head(n) = (I - rho G)^-1 x_c  (Woodbury on K = K_rho - rho E E^T; derivation and
// accuracy in fccqp_fused_impl.h, "presolve").  W = I - rho G is symmetric positive definite exactly when
// K is non-singular, with eigenvalues in (0, 1]; a pivot <= kPresolveMinPivot sends the instance to the
// COD fallback.  This wave's share of W, packed lower, from the staged rho G (column-major, global):
constexpr double kPresolveMinPivot = 1e-10;
// (lane = row, 16 columns per trip with all their loads issued before the first store: one load in
// flight per trip made this 70 us at n = 114 -- 228 dependent L2 round trips)
__device__ __forceinline__ void presolve_w_share(lds_f64 *M, const double *stage, const int n, const int lane,
                                                 const int part, const int parts) {
  constexpr int CB = 16;
  for (int j0 = CB * part; j0 < n; j0 += CB * parts) {
    double v[CB][2];
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      const int j = (j0 + u < n) ? j0 + u : n - 1;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = lane + kWave * t;
        v[u][t] = (i < n && i >= j) ? stage[(size_t)j * n + i] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      const int j = j0 + u;
      if (j < n) {
        const int cj = col_base(j, n);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int i = lane + kWave * t;
          if (i < n && i >= j) M[cj + i] = ((i == j) ? 1.0 : 0.0) - v[u][t];
        }
      }
    }
  }
}

// waves 1..WAVES-1 of a multi-wave workgroup, for one instance
template <int SLOTS, int WAVES, class TIN, class SHP>
__device__ void helper_wave(const KernelArgs &a, double *lds, const int inst, const int wave, const int lane) {
  constexpr int RB = g_batch<SLOTS>();
  const int n = SHP::fixed ? SHP::n : a.n, N = SHP::fixed ? SHP::n + SHP::m : a.n + a.m;
  lds_f64 *M = (lds_f64 *)lds;
  volatile lds_f64 *ctrl = ctrl_word(lds, N, SHP::fixed ? SHP::nc : a.nc);
  const bool mv_regs = mv_regs_ok(n, packed_doubles(N));
  const int qbase = kMvBlocks * (wave - 1);
  int cmd = 0;
  bool have_cmd = false;                               // the operator loop below leaves with the next command read
  for (;;) {
    if (!have_cmd) {
      workgroup_sync();                                 // A: a command is posted
      cmd = (int)ctrl[0];
    }
    have_cmd = false;
    if (cmd == kCmdDone) break;
    if (cmd == kCmdAssemble) {
      const TIN *Q = reinterpret_cast<const TIN *>(a.Q) + (size_t)inst * n * n;
      const TIN *A = reinterpret_cast<const TIN *>(a.A) + (size_t)inst * (N - n) * n;
      assemble_share<TIN>(M, Q, A, n, N, (double)ctrl[1], lane, wave, WAVES);
      workgroup_sync();                                 // B: the matrix is complete
      continue;
    }
    if (cmd == kCmdMatvec) {                           // (LDS-resident G) this wave's quarter of the columns
      constexpr int GS = 2;
      double acc0[GS] = {0.0, 0.0}, acc1[GS] = {0.0, 0.0};
      const int nchunk = g_chunks(n), per = (nchunk + WAVES - 1) / WAVES;
      const int cb = wave * per, ce = ((wave + 1) * per < nchunk) ? (wave + 1) * per : nchunk;
      matvec_range<GS>(M, n, lane, cb, ce, acc0, acc1);
      lds_f64 *part = M + g_part_off(n);
#pragma unroll
      for (int t = 0; t < GS; ++t) part[wave * 2 * kWave + lane + kWave * t] = acc0[t] + acc1[t];
      workgroup_sync();                                 // B
      continue;
    }
    if (cmd == kCmdFactor) {
      const bool bad = ldlt_factor_share<SLOTS>(M, N, lane, wave, WAVES, ctrl + 2, (double)ctrl[1]);
      if (lane == 0) ctrl[3 + wave] = bad ? 1.0 : 0.0;
      workgroup_sync();                                 // B: factorisation complete
      continue;
    }
    double *stage = a.workspace + (size_t)inst * (size_t)a.workspace_stride;
    if (g_build_mfma_supported(N, n)) {
      g_build_mfma_share<SLOTS>(M, N, n, lane, wave, a.rho, stage, (lds_f64 *)ctrl + kCtrlWords);
    } else {
      for (int b = wave; b * RB <= n; b += WAVES)
        g_build_batch<SLOTS, RB, false>(M, N, n, lane, b * RB, a.rho, stage, nullptr, nullptr, nullptr);
    }
    agent_release();
    workgroup_sync();                                   // B: every column of G is staged, L is dead
    agent_acquire();
    if ((double)ctrl[1] != 0.0) {                      // the solver wave wants x0 out of G first:
      workgroup_sync();                                 // B2: ... it has built W = I - rho G in LDS;
      const bool wb = ldlt_factor_share<(SLOTS < 2 ? SLOTS : 2)>(M, n, lane, wave, WAVES, ctrl + 2, kPresolveMinPivot);
      if (lane == 0) ctrl[3 + wave] = wb ? 1.0 : 0.0;  // the n x n factorisation is shared like the KKT one
      workgroup_sync();                                 // B3: W is factorised (the solver wave solves alone)
      // B4: ... and has solved.  Only where rho G goes into LDS next, over the factor of W the solver wave is
      // still sweeping through (with G in the helpers' registers nobody writes LDS here)
      if (!mv_regs) workgroup_sync();
    }
    if (mv_regs) {
      // rho G stays out of LDS: this wave's columns into registers (staged column-major, global / L2).
      // The array lives in this scope only, and the operator iterations are served from a loop of
      // their own right here: its live range then holds no call, so it can sit in VGPRs (kept across
      // the command loop -- which calls the G build and the shared factorisation -- it was parked in
      // callee-saved AGPRs and every FMA operand cost two v_accvgpr_read).
      double Greg[kMvBlocks][16][2];                   // rows lane and 64 + lane, columns 16 (qbase + qi) + l
#pragma unroll
      for (int qi = 0; qi < kMvBlocks; ++qi) {
#pragma unroll
        for (int l = 0; l < 16; ++l) {
          const int c = 16 * (qbase + qi) + l;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int row = lane + kWave * t;
            Greg[qi][l][t] = (c < n && row < n) ? stage[(size_t)c * n + row] : 0.0;
          }
        }
      }
      workgroup_sync();                                 // C: G is in place
      const lds_f64 *wbuf = M + g_w_off(n);
      lds_f64 *part = M + g_w_off(n) + kMvWLen;
      for (;;) {
        workgroup_sync();                               // A
        cmd = (int)ctrl[0];
        if (cmd != kCmdMatvec) { have_cmd = true; break; }
        double Wq[kMvBlocks];
#pragma unroll
        for (int qi = 0; qi < kMvBlocks; ++qi) Wq[qi] = wbuf[16 * (qbase + qi) + (lane & 15)];
        double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
        // DPP reads of a VGPR need two wait states after a VALU write of it (see the fused kernel)
        asm volatile("s_nop 1" : "+v"(Wq[0]), "+v"(Wq[1]), "+v"(Wq[2]));
        dpp_mv_rows<0>(acc, Wq, Greg);
        part[wave * 2 * kWave + lane] = acc[0][0] + acc[0][1];
        part[wave * 2 * kWave + kWave + lane] = acc[1][0] + acc[1][1];
        workgroup_sync();                               // B
      }
      continue;
    }
    {
      const int total = n * n;
#pragma unroll 8
      for (int e = wave * kWave + lane; e < total; e += WAVES * kWave) M[e] = stage[e];
    }
    workgroup_sync();                                   // C: G is in LDS
  }
}

// Solves instance `inst`.  x0_given: the presolve was done elsewhere (fallback pass): x0 comes from
// a.x, the duals start at zero, the presolve flags are reported as "ran, numerical, COD".
template <int SLOTS, bool MAT_IN_LDS, class TIN, class SHP>
__device__ void solve_instance(const KernelArgs &a, double *lds, const int inst, const bool x0_given) {
  const int lane = threadIdx.x;
  const int n = SHP::fixed ? SHP::n : a.n, m = SHP::fixed ? SHP::m : a.m, nc = SHP::fixed ? SHP::nc : a.nc,
            s0 = SHP::fixed ? SHP::s : a.s, N = n + m;
  const int npack = packed_doubles(N);
  // The argument block reaches this function through a pointer (its address is taken for the real
  // calls), so a.field is a scratch load wherever the compiler cannot prove the copy unchanged -- in
  // the iteration loop, one per use per iteration.  The scalars of the loop are read once.
  const double opt_rho = a.rho, opt_eps_bound = a.eps_bound, opt_eps_fcone = a.eps_fcone;
  const int opt_max_iter = a.max_iter;
  // optional phase stamps (fccqp_phase_seconds): [0] up to the end of the K_rho assembly (incl. a direct
  // presolve), [1] factorisations, [3] operator build incl. the presolve out of G ([2] of it), [4] iterations
  const long long ph_begin = wall_clock64();
  long long ph_asm_end = ph_begin, ph_build = 0, ph_presolve = 0, ph_iter_begin = ph_begin;

  // matrix: LDS (ds_* instructions) or the per-block global workspace (global_* instructions)
  using MP = std::conditional_t<MAT_IN_LDS, lds_f64 *, double *>;
  MP M;
  if constexpr (MAT_IN_LDS) M = (lds_f64 *)lds;
  else M = a.workspace + (long long)blockIdx.x * a.workspace_stride;
  lds_f64 *cone_buf = (lds_f64 *)(MAT_IN_LDS ? (lds + npack) : lds);  // nc doubles of cross-lane scratch

  // TIN: storage type of Q / A_eq (double, or float for the fp32-storage entry points)
  const TIN *Q = reinterpret_cast<const TIN *>(a.Q) + (size_t)inst * n * n;
  const TIN *A = reinterpret_cast<const TIN *>(a.A) + (size_t)inst * m * n;
  const double *bq = a.b + (size_t)inst * n;
  const double *beq = a.b_eq + (size_t)inst * m;
  const double *mu = a.mu + (size_t)inst * (nc / 3);
  const double *lbp = a.lb + (size_t)inst * n;
  const double *ubp = a.ub + (size_t)inst * n;
  double *xs = a.x + (size_t)inst * n;
  double *mux_s = a.mu_x + (size_t)inst * n;
  double *mulc_s = a.mu_lc + (size_t)inst * nc;

  // ---- per-lane vectors (entry i = lane + 64 t) ------------------------------------------
  double vb[SLOTS], vlb[SLOTS], vub[SLOTS], vtail[SLOTS];  // b, lb, ub over [0,n); b_eq at [n,N)
  double x[SLOTS], mux[SLOTS], mulc[SLOTS];
  bool is_cone[SLOTS];
  bool all_inf = true;
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) {
    const int i = lane + kWave * t;
    vb[t] = (i < n) ? bq[i] : 0.0;
    vlb[t] = (i < n) ? lbp[i] : -HUGE_VAL;
    vub[t] = (i < n) ? ubp[i] : HUGE_VAL;
    vtail[t] = (i >= n && i < N) ? beq[i - n] : 0.0;
    is_cone[t] = (i >= s0 && i < s0 + nc);
    if (i < n) all_inf = all_inf && isinf(vlb[t]) && isinf(vub[t]);  // fcc_qp.cpp:132-133
    if (x0_given && !a.warm_start) {
      x[t] = (i < n) ? xs[i] : 0.0; mux[t] = 0.0; mulc[t] = 0.0;
    } else if (a.warm_start) {                                     // :136-139 (x0_given: x is the COD x0)
      x[t] = (i < n) ? xs[i] : 0.0;
      mux[t] = (i < n) ? mux_s[i] : 0.0;
      mulc[t] = is_cone[t] ? mulc_s[i - s0] : 0.0;
    } else {
      x[t] = 0.0; mux[t] = 0.0; mulc[t] = 0.0;
    }
  }
  const bool equality_constrained = (nc == 0) && __all(all_inf);
  int flags = equality_constrained ? kFlagEqualityOnly : 0;
  long long ticks = 0;

  // ---- assemble the packed lower KKT (fcc_qp.cpp:146-150) --------------------------------
  auto assemble = [&](double rho) {
    constexpr int WAVES = waves_for<SLOTS, MAT_IN_LDS>();
    if constexpr (WAVES > 1) {
      // all four waves fetch (the lone wave would otherwise sit through ~40 dependent HBM round trips)
      lds_f64 *ctrl = ctrl_word(lds, N, nc);
      if (lane == 0) { ctrl[0] = (double)kCmdAssemble; ctrl[1] = rho; }
      workgroup_sync();                                 // A (see helper_wave)
      assemble_share<TIN>(M, Q, A, n, N, rho, lane, 0, WAVES);
      workgroup_sync();                                 // B
    } else {
      assemble_share<TIN>(M, Q, A, n, N, rho, lane, 0, 1);
      wave_sync();
      if constexpr (!MAT_IN_LDS) global_phase_sync();
    }
  };

  // LDL^T of M in place; with four waves the panels are dealt round-robin (ldlt_factor_share)
  auto factor = [&](const double tiny) -> bool {
    constexpr int WAVES = waves_for<SLOTS, MAT_IN_LDS>();
    if constexpr (WAVES > 1) {
      volatile lds_f64 *ctrl = ctrl_word(lds, N, nc);
      if (lane == 0) { ctrl[0] = (double)kCmdFactor; ctrl[1] = tiny; ctrl[2] = 0.0; }
      workgroup_sync();                                 // A
      const bool mine = ldlt_factor_share<SLOTS>(M, N, lane, 0, WAVES, ctrl + 2, tiny);
      if (lane == 0) ctrl[3] = mine ? 1.0 : 0.0;
      workgroup_sync();                                 // B
      bool bad = false;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) bad = bad || (ctrl[3 + w] != 0.0);
      return bad;
    } else {
      const bool bad = ldlt_factor<SLOTS>(M, N, lane, tiny);
      if constexpr (!MAT_IN_LDS) global_phase_sync();
      return bad;
    }
  };

  // Can this instance run on the explicit operator x = x_c + (rho G) w (see the ADMM loop)?  Everything
  // but the iteration-count condition; needed here because G also yields the presolve.
  constexpr int GS = (SLOTS < 2) ? SLOTS : 2;        // register slots that hold variables (n <= 128)
  constexpr int WAVES = waves_for<SLOTS, MAT_IN_LDS>();
  bool single_mfma = false;
  if constexpr (MAT_IN_LDS && WAVES == 1)
    single_mfma = a.workspace != nullptr && g_build_mfma_single_supported(N, n, a.workspace_stride);
  const int gswitch = (WAVES > 1 || single_mfma) ? kGSwitchIterMulti : kGSwitchIter;
  const int w_off = g_w_off(n), w_len = g_w_len(n);
  bool g_possible = false;
  if constexpr (MAT_IN_LDS && SLOTS <= 4) {
    g_possible = a.workspace != nullptr && n <= 2 * kWave && n > 0 && w_off + w_len <= npack &&
                 a.workspace_stride >= (long long)n * n;
  }
  bool use_g = g_possible &&
               opt_max_iter - gswitch >= ((WAVES > 1 || single_mfma) ? kGMinRemainingMulti : kGMinRemaining);

  // ---- presolve (fcc_qp.cpp:159-178): skipped on warm start unless equality-only ---------
  // Three ways to x0 = (K^-1 [-b; b_eq]).head(n):
  //  (w) out of the K_rho factorisation, x0 = (I - rho G)^-1 x_c, when G is going to be built anyway
  //      (use_g): no factorisation of K, no second assembly -- and no breakdown on semi-definite costs;
  //  (d) the direct unpivoted LDL^T of K, as the reference does with its pivoted one: equality-only
  //      instances (no K_rho at all), shapes / iteration limits without G;
  //  (c) the COD fallback pass, where (d) breaks down and (w) is not possible, or W is singular.
  bool need_w = false;
  if (x0_given) {
    flags |= kFlagPresolveRan | kFlagPresolveNumerical | kFlagPresolveCod;
  } else if (equality_constrained || !a.warm_start) {
    flags |= kFlagPresolveRan;
    if (!equality_constrained && use_g) {
      need_w = true;                                   // (w)
    } else {
      // breakdown threshold of the unpivoted presolve (see ldlt_panel): 16 N eps max_i |Q_ii|
      double qmax = 0.0;
#pragma unroll
      for (int t = 0; t < SLOTS; ++t) {
        const int i = lane + kWave * t;
        const double q = (i < n) ? fabs((double)Q[(size_t)i * n + i]) : 0.0;
        qmax = (qmax < q) ? q : qmax;
      }
      const double tiny = 16.0 * (double)N * DBL_EPSILON * wave_max(qmax);
      assemble(0.0);
      const long long t0 = wall_clock64();
      const bool bad = factor(tiny);
      ticks += wall_clock64() - t0;
      if (bad && !equality_constrained && g_possible) {
        need_w = true;                                 // (d) broke down: (w) after all
        use_g = true;
      } else if (bad) {
        flags |= kFlagPresolveNumerical;
        if (a.fb_list) {
          // where the reference switches to its COD fallback (fcc_qp.cpp:164-166): hand the
          // instance to the fallback pass and stop here
          if (lane == 0) {
            a.fb_list[atomicAdd(a.fb_count, 1)] = inst;
            a.flags[inst] = flags;
          }
          return;
        }
      }
      if (!need_w) {
        double r[SLOTS];
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
          const int i = lane + kWave * t;
          r[t] = (i < n) ? -vb[t] : vtail[t];  // [-b; b_eq]  (:149-150)
        }
        ldlt_solve<SLOTS, MP>(M, N, lane, r);
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
          const int i = lane + kWave * t;
          x[t] = (i < n) ? r[t] : 0.0;       // x_ = sol.head(n)  (:174)
        }
      }
    }
  }

  int n_iter = 0;
  double x_res_norm = 0.0, lc_res_norm = 0.0;

  // ---- DoADMM (fcc_qp.cpp:57-112) ----------------------------------------------------------
  if (!equality_constrained) {
    assemble(opt_rho);                                   // :62-64
    const long long t0 = wall_clock64();
    ph_asm_end = t0;
    const bool bad = factor(0.0);                      // :68 (info() unchecked in the reference)
    ticks += wall_clock64() - t0;
    ph_iter_begin = wall_clock64();
    if (bad) flags |= kFlagAdmmNumerical;

    double xbar[SLOTS], lcbar[SLOTS];
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) { xbar[t] = x[t]; lcbar[t] = x[t]; }  // :74-75
    // per-lane constants of the projections, and the last residuals (for the reported norms)
    int cidx[SLOTS], ccomp[SLOTS];
    double vmu[SLOTS], vkc[SLOTS], xr_last[SLOTS], lr_last[SLOTS];
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) {
      const int i = lane + kWave * t;
      cidx[t] = is_cone[t] ? (i - s0) / 3 : 0;
      ccomp[t] = is_cone[t] ? (i - s0) - 3 * cidx[t] : 0;
      vmu[t] = is_cone[t] ? mu[cidx[t]] : 0.0;
      vkc[t] = cone_lane_constant(vmu[t], ccomp[t] == 2);
      xr_last[t] = 0.0; lr_last[t] = 0.0;
    }

    // The first kGSwitchIter iterations solve the KKT system by two triangular sweeps, like the
    // reference (:83-87).  An instance still running after that gets the explicit operator
    //   x = x_c + (rho G) w,   G = (K_rho^-1)[0:n,0:n],  x_c = (K_rho^-1 [-b; b_eq])[0:n],
    //   w = x_bar - mu_x (contact segment: lambda_c_bar - mu_lambda_c)          -- same solve, :81-87
    // built by n + 1 simultaneous substitutions (ldlt_solve_multi), staged through a.workspace and
    // copied into LDS over L, which is not needed any more: an iteration becomes an n x n mat-vec
    // (~15x fewer instructions than the sweeps at N = 174).  Building G costs about 30 sweep
    // iterations, so it is only done for instances that have already shown they iterate.
    // A cold instance that takes its presolve from G (need_w, see above) builds G before the first
    // iteration instead -- the pass iter == -1 of the loop below -- and iterates on the operator
    // from the start.
    constexpr int RB = g_batch<SLOTS>();
    // LDS doubles after the switch: rho*G [n][n] (column-major) | w [w_len], zero-padded | partial sums
    const int g_off = 0;
    bool g_ready = false; (void)g_ready;
    const bool coop_mv = (WAVES > 1) && g_part_off(n) + WAVES * 2 * kWave <= npack;
    const bool mv_regs = (WAVES > 1) && mv_regs_ok(n, npack);     // same predicate as helper_wave
    double xc[GS];
#pragma unroll
    for (int t = 0; t < GS; ++t) xc[t] = 0.0;

    n_iter = opt_max_iter;                               // :78
    // The iteration is written as three pieces -- the KKT solve by sweeps, the KKT solve by the
    // operator, and everything after the solve -- and two loops without a call in them: with the G
    // build (a real call, ~400 registers) inside the one iteration loop, every value of the solver wave
    // that did not fit the callee-saved registers lived in scratch and was reloaded from memory in
    // each iteration (ten scratch loads, ~2 us per iteration on the large shape).
    bool g_failed = false;                             // W singular: the instance goes to the COD pass
    int iter = 0;
    bool done = false;
    const int first_op = use_g ? (need_w ? 0 : gswitch) : opt_max_iter;   // first iteration on the operator
    // (no lambdas here: captured by reference, some of the per-lane arrays stayed in scratch memory)
#pragma nounroll
    for (int phase = 0; phase < 2 && !done; ++phase) {
      int iter_end = (first_op < opt_max_iter) ? first_op : opt_max_iter;
      if (phase == 1) {
        if (!(use_g && (need_w || iter < opt_max_iter))) break;
        const bool with_presolve = need_w;
        const long long tb0 = wall_clock64();
          if constexpr (MAT_IN_LDS && SLOTS <= 4) {
          double *stage = a.workspace + (size_t)inst * (size_t)a.workspace_stride;
          if constexpr (WAVES > 1) {
            if (g_build_mfma_supported(N, n)) {        // [-b; b_eq] where every wave can read it
              lds_f64 *rhs_s = ctrl_word(lds, N, nc) + kCtrlWords;
    #pragma unroll
              for (int t = 0; t < SLOTS; ++t) {
                const int i = lane + kWave * t;
                if (i < N) rhs_s[i] = (i < n) ? -vb[t] : vtail[t];
              }
            }
            if (lane == 0) { ctrl_word(lds, N, nc)[0] = (double)kCmdBuild; ctrl_word(lds, N, nc)[1] = with_presolve ? 1.0 : 0.0; }
            workgroup_sync();                           // A (see helper_wave)
          }
          bool by_mfma = false;
          if constexpr (WAVES > 1) by_mfma = g_build_mfma_supported(N, n);
          if constexpr (WAVES > 1) {
            if (by_mfma) g_build_mfma_share<SLOTS>(M, N, n, lane, 0, opt_rho, stage, ctrl_word(lds, N, nc) + kCtrlWords);
          }
          if constexpr (WAVES == 1) {
            if (single_mfma) {
              double *gs = stage + (size_t)n * n;
    #pragma unroll
              for (int t = 0; t < SLOTS; ++t) {
                const int i = lane + kWave * t;
                if (i < N) gs[i] = (i < n) ? -vb[t] : vtail[t];
              }
              agent_release_acquire();
              wave_sync();
              g_build_mfma_single(M, N, n, lane, opt_rho, stage, gs);
              by_mfma = true;
            }
          }
          if (!by_mfma) {
            g_build_batch<SLOTS, RB, true>(M, N, n, lane, 0, opt_rho, stage, vb, vtail, xc);
            for (int b = WAVES; b * RB <= n; b += WAVES)
              g_build_batch<SLOTS, RB, false>(M, N, n, lane, b * RB, opt_rho, stage, nullptr, nullptr, nullptr);
          }
          // the staged columns are read back by other lanes (and waves)
          agent_release();
          if constexpr (WAVES > 1) workgroup_sync();    // B
          else wave_sync();
          agent_acquire();
          if constexpr (WAVES > 1) {
            if (by_mfma) {                             // x_c came back through LDS
              const lds_f64 *xc_s = ctrl_word(lds, N, nc) + kCtrlWords + N;
    #pragma unroll
              for (int t = 0; t < GS; ++t) {
                const int i = lane + kWave * t;
                xc[t] = (i < n) ? xc_s[i] : 0.0;
              }
            }
          } else {
            if (by_mfma) {                             // ... or through the staging buffer
              const double *xc_g = stage + (size_t)n * n + N;
    #pragma unroll
              for (int t = 0; t < GS; ++t) {
                const int i = lane + kWave * t;
                xc[t] = (i < n) ? xc_g[i] : 0.0;
              }
            }
          }
          bool wbad = false;
          if (with_presolve) {
            // presolve: x0 = (I - rho G)^-1 x_c.  W goes over the dead factor in LDS (this wave alone:
            // 4 us at n = 114), its n x n factorisation is shared by the waves like the KKT one, this
            // wave solves.
            presolve_w_share(M, stage, n, lane, 0, 1);
            const long long tw = wall_clock64();
            if constexpr (WAVES > 1) {
              volatile lds_f64 *ctrl = ctrl_word(lds, N, nc);
              if (lane == 0) ctrl[2] = 0.0;          // finished-panel count of the shared factorisation
              workgroup_sync();                       // B2
              const bool mine = ldlt_factor_share<GS>(M, n, lane, 0, WAVES, ctrl + 2, kPresolveMinPivot);
              if (lane == 0) ctrl[3] = mine ? 1.0 : 0.0;
              workgroup_sync();                       // B3
#pragma unroll
              for (int w = 0; w < WAVES; ++w) wbad = wbad || (ctrl[3 + w] != 0.0);
            } else {
              wave_sync();
              wbad = ldlt_factor<GS>(M, n, lane, kPresolveMinPivot);
            }
            ticks += wall_clock64() - tw;
            ph_presolve = wall_clock64() - tw;
            double xw[GS];
    #pragma unroll
            for (int t = 0; t < GS; ++t) xw[t] = xc[t];
            ldlt_solve<GS, lds_f64 *>(M, n, lane, xw);
    #pragma unroll
            for (int t = 0; t < SLOTS; ++t) {
              x[t] = 0.0;
              if (t < GS) x[t] = (lane + kWave * t < n) ? xw[t < GS ? t : 0] : 0.0;
              xbar[t] = x[t]; lcbar[t] = x[t];           // :74-75
            }
            wave_sync();
            if constexpr (WAVES > 1) {
              // B4 (helper_wave): the helpers copy rho G into LDS over the factor of W -- not before this wave has
              // solved with it.  (Found by tools/fuzz_parity.py in round 4: shapes with N > 128 whose LDS has room
              // for G and w but not for the partial sums of a shared mat-vec, e.g. (89, 40), (94, 40), returned an x0
              // swept through a half-overwritten factor on every cold start.)
              if (!mv_regs) workgroup_sync();
            }
          }
          if (!(WAVES > 1 && mv_regs)) {              // (four waves: the helpers keep G in registers)
            const int total = n * n;
    #pragma unroll 8
            for (int e = lane; e < total; e += WAVES * kWave) M[g_off + e] = stage[e];
          }
          if constexpr (WAVES > 1) workgroup_sync();    // C
          else wave_sync();
          g_ready = true;
          if (wbad) {
            // W singular to working precision: K is singular; the reference's COD gives the
            // minimum-norm solution (fcc_qp.cpp:164-166,176) -> fallback pass
            flags |= kFlagPresolveNumerical;
            if (a.fb_list) {
              if (lane == 0) {
                a.fb_list[atomicAdd(a.fb_count, 1)] = inst;
                a.flags[inst] = flags;
              }
              g_failed = true;
            }
          }
          }

        if (g_failed) return;
        ph_build = wall_clock64() - tb0;
        iter_end = opt_max_iter;
      }
#pragma nounroll
      for (; iter < iter_end; ++iter) {
        if (phase == 1) {
            if constexpr (MAT_IN_LDS && SLOTS <= 4) {
            lds_f64 *wbuf = M + w_off;
      #pragma unroll
            for (int t = 0; t < GS; ++t) {
              const int i = lane + kWave * t;
              if (i < n) wbuf[i] = is_cone[t] ? (lcbar[t] - mulc[t]) : (xbar[t] - mux[t]);
            }
            // zero pad (use_g checked the room): padded pairs multiply a clamped, valid column by w = 0
            const int nchunk = g_chunks(n);
            const int wpad = mv_regs ? kMvWLen : w_len;
            for (int i = n + lane; i < wpad; i += kWave) wbuf[i] = 0.0;
            double acc0[GS], acc1[GS];
      #pragma unroll
            for (int t = 0; t < GS; ++t) { acc0[t] = xc[t]; acc1[t] = 0.0; }
            if constexpr (WAVES > 1) {
              if (mv_regs) {
                // the helpers hold rho G in registers (helper_wave): post w, collect three partial sums
                if (lane == 0) ctrl_word(lds, N, nc)[0] = (double)kCmdMatvec;
                workgroup_sync();                         // A: w is in LDS
                workgroup_sync();                         // B: the helpers' partial sums are in LDS
                const lds_f64 *part = M + w_off + kMvWLen;
      #pragma unroll
                for (int w = 1; w < WAVES; ++w) {
      #pragma unroll
                  for (int t = 0; t < GS; ++t) acc0[t] += part[w * 2 * kWave + lane + kWave * t];
                }
              } else if (coop_mv) {
                // the four waves take a quarter of the columns each; partial sums meet in LDS
                const int per = (nchunk + WAVES - 1) / WAVES;
                if (lane == 0) ctrl_word(lds, N, nc)[0] = (double)kCmdMatvec;
                workgroup_sync();                         // A: w is in LDS
                matvec_range<GS>(M, n, lane, 0, (per < nchunk) ? per : nchunk, acc0, acc1);
                workgroup_sync();                         // B: the helpers' partial sums are in LDS
                const lds_f64 *part = M + g_part_off(n);
      #pragma unroll
                for (int w = 1; w < WAVES; ++w) {
      #pragma unroll
                  for (int t = 0; t < GS; ++t) acc0[t] += part[w * 2 * kWave + lane + kWave * t];
                }
              } else {
                wave_sync();
                matvec_range<GS>(M, n, lane, 0, nchunk, acc0, acc1);
              }
            } else {
              wave_sync();
              matvec_range<GS>(M, n, lane, 0, nchunk, acc0, acc1);
            }
      #pragma unroll
            for (int t = 0; t < SLOTS; ++t) x[t] = 0.0;
      #pragma unroll
            for (int t = 0; t < GS; ++t) {
              const int i = lane + kWave * t;
              x[t] = (i < n) ? acc0[t] + acc1[t] : 0.0;
            }
            }

        } else {
            double r[SLOTS];
      #pragma unroll
            for (int t = 0; t < SLOTS; ++t) {
              const int i = lane + kWave * t;
              double q = -opt_rho * (xbar[t] - mux[t]);                    // :81
              if (is_cone[t]) q = -opt_rho * (lcbar[t] - mulc[t]);          // :82 overwrites the segment
              r[t] = (i < n) ? -(vb[t] + q) : vtail[t];                   // :83
            }
            ldlt_solve<SLOTS, MP>(M, N, lane, r);                             // :86
      #pragma unroll
            for (int t = 0; t < SLOTS; ++t) {
              const int i = lane + kWave * t;
              x[t] = (i < n) ? r[t] : 0.0;                                // :87
            }

        }
        // slack update :90-92
        wave_sync();
  #pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
          const int i = lane + kWave * t;
          if (is_cone[t]) cone_buf[i - s0] = x[t] + mulc[t];
        }
        wave_sync();
  #pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
          const int i = lane + kWave * t;
          xr_last[t] = 0.0; lr_last[t] = 0.0;
          if (i < n) {
            const double v = x[t] + mux[t];
            const double tt = (vub[t] < v) ? vub[t] : v;   // std::min(x, ub)
            xbar[t] = (tt < vlb[t]) ? vlb[t] : tt;         // std::max(., lb)  constraint_utils.cpp:43
            xr_last[t] = x[t] - xbar[t];                    // :95
            mux[t] += xr_last[t];                           // :101
          }
          if (is_cone[t]) {
            const int base = 3 * cidx[t];
            lcbar[t] = project_cone_component(cone_buf[base], cone_buf[base + 1], cone_buf[base + 2], vmu[t], vkc[t],
                                              x[t] + mulc[t], ccomp[t] == 2);
            lr_last[t] = x[t] - lcbar[t];                   // :96
            mulc[t] += lr_last[t];                          // :102
          }
        }
        // exit test :105-109.  norm_inf(r) < eps  <=>  every |r_i| < eps: one ballot per iteration
        // instead of four wave reductions; the norms themselves (:97-98) are only reported and are
        // reduced once after the loop.  An empty contact set has norm 0.
        bool ok = true;
  #pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
          const int i = lane + kWave * t;
          if (i < n) ok = ok && (fabs(xr_last[t]) < opt_eps_bound);
          if (is_cone[t]) ok = ok && (fabs(lr_last[t]) < opt_eps_fcone);
        }
        if (nc == 0) ok = ok && (0.0 < opt_eps_fcone);
        if (n == 0) ok = ok && (0.0 < opt_eps_bound);

        if (__all(ok)) { n_iter = iter; done = true; break; }
      }
    }
    if (opt_max_iter > 0) {
      // :97-98  abs(max(maxCoeff, -minCoeff)) == max_i |r_i| of the last iteration executed
      double xm = 0.0, lm = 0.0;
#pragma unroll
      for (int t = 0; t < SLOTS; ++t) {
        const double ax = fabs(xr_last[t]), al = fabs(lr_last[t]);
        xm = (xm < ax) ? ax : xm;
        lm = (lm < al) ? al : lm;
      }
      x_res_norm = wave_max(xm);
      lc_res_norm = wave_max(lm);
    }
  }

  // ---- epilogue (fcc_qp.cpp:184-186) -------------------------------------------------------
  double bv = 0.0, cv = 0.0;
  wave_sync();
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) {
    const int i = lane + kWave * t;
    if (is_cone[t]) cone_buf[i - s0] = x[t];
  }
  wave_sync();
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) {
    const int i = lane + kWave * t;
    if (i < n) {
      const double tt = (vub[t] < x[t]) ? vub[t] : x[t];
      const double p = (tt < vlb[t]) ? vlb[t] : tt;
      const double d = x[t] - p;
      bv += d * d;                                         // constraint_utils.cpp:61-65
    }
    if (is_cone[t] && ((i - s0) % 3 == 0)) {
      const int c = (i - s0) / 3;
      const double fx = cone_buf[3 * c], fy = cone_buf[3 * c + 1], fz = cone_buf[3 * c + 2];
      const double v = sqrt(fx * fx + fy * fy) - mu[c] * fz;
      cv += (0.0 < v) ? v : 0.0;                           // constraint_utils.cpp:48-59
    }
  }
  bv = sqrt(wave_sum(bv));
  cv = wave_sum(cv);

  // ---- write back state/solution and details ----------------------------------------------
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) {
    const int i = lane + kWave * t;
    if (i < n) { xs[i] = x[t]; mux_s[i] = mux[t]; }
    if (is_cone[t]) mulc_s[i - s0] = mulc[t];
  }
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
      const long long t_end = wall_clock64();
      a.phase_ticks[0 * B + inst] = ph_asm_end - ph_begin;
      a.phase_ticks[1 * B + inst] = ticks;
      a.phase_ticks[2 * B + inst] = ph_presolve;
      a.phase_ticks[3 * B + inst] = ph_build;
      a.phase_ticks[4 * B + inst] = (t_end - ph_iter_begin) - ph_build;
      a.phase_ticks[5 * B + inst] = t_end - ph_begin;
      a.phase_ticks[6 * B + inst] = 0;
      a.phase_ticks[7 * B + inst] = ph_begin;
    }
  }
}

template <int SLOTS, bool MAT_IN_LDS, class TIN, class SHP = DynShape>
__global__ __launch_bounds__((kWave * waves_for<SLOTS, MAT_IN_LDS>())) void generic_kernel(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int WAVES = waves_for<SLOTS, MAT_IN_LDS>();
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  // wave 0 runs the solver (all of it when WAVES == 1); see helper_wave for the others
  auto run = [&](const int inst, const bool from_list) {
    if (wave == 0) {
      if (from_list) {                                   // fcc_qp.cpp:165,176
        // cod_stride < 0: the list launch was given enough LDS for the COD's work matrices (they are
        // dead before solve_instance lays the packed KKT over them)
        if (a.cod_stride < 0) cod::cod_presolve<TIN, true>(a, inst, (double *)lds);
        else cod::cod_presolve<TIN, false>(a, inst, a.cod_ws + (long long)blockIdx.x * a.cod_stride);
      }
      solve_instance<SLOTS, MAT_IN_LDS, TIN, SHP>(a, lds, inst, from_list);
      if constexpr (WAVES > 1) {
        if (lane == 0) ctrl_word(lds, SHP::fixed ? SHP::n + SHP::m : a.n + a.m, SHP::fixed ? SHP::nc : a.nc)[0] = (double)kCmdDone;
        workgroup_sync();
      }
    } else {
      if constexpr (WAVES > 1) helper_wave<SLOTS, WAVES, TIN, SHP>(a, lds, inst, wave, lane);
    }
  };
  if (a.inst_list) {  // list mode (fallback launch): persistent blocks walk the fallback list
    const int total = *a.inst_count;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      *a.next_count = 0;
      // (list mode does not append: fb_list is null.  fb_count then points at a word of page-locked host memory in
      // which the host finds, some time later, that a launch had a list to walk and how long it was --
      // fccqp_capi.cpp sizes the NEXT fallback launches by it.  An empty list, the rule, writes nothing: the store
      // crosses PCIe and the end of the kernel waits for it.)
      if (a.fb_count && total != 0) *a.fb_count = total;
    }
    for (int li = blockIdx.x; li < total; li += gridDim.x) run(a.inst_list[li], true);
  } else {
    // block -> instance map (fccqp_device.h)
    int inst;
    if constexpr (WAVES > 1) {
      __shared__ int claimed;
      if (wave == 0) {
        const int c = block_instance(a);
        if (lane == 0) claimed = c;
      }
      workgroup_sync();
      inst = claimed;
    } else {
      inst = block_instance(a);
    }
    if (inst >= 0) run(inst, false);
  }
}

inline int slots_for(int N) { return (N + kWave - 1) / kWave; }

template <int SLOTS, class TIN>
hipError_t launch_slots(const KernelArgs &a, bool mat_in_lds, size_t lds_bytes, int blocks, hipStream_t stream) {
  dim3 grid(blocks), block(kWave);
  if (mat_in_lds) block = dim3(kWave * waves_for<(SLOTS > 4 ? 4 : SLOTS), true>());
  if constexpr (SLOTS > 4) {
    if (mat_in_lds) return hipErrorInvalidValue;   // N > 256 never fits the 160 KiB LDS
  }
  if (mat_in_lds) {
    auto k = generic_kernel<(SLOTS > 4 ? 4 : SLOTS), true, TIN>;
    if constexpr (SLOTS == 3) {
      if (a.n == LargeShape::n && a.m == LargeShape::m && a.nc == LargeShape::nc && a.s == LargeShape::s)
        k = generic_kernel<3, true, TIN, LargeShape>;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, block, lds_bytes, stream, a);
  } else {
    auto k = generic_kernel<SLOTS, false, TIN>;
    hipLaunchKernelGGL(k, grid, block, lds_bytes, stream, a);
  }
  return hipGetLastError();
}

// the five small per-instance vectors of an fp32-storage solve, float -> double, in one launch
struct WidenJob { const float *src[5]; double *dst[5]; size_t count[5]; };
__global__ void widen_kernel(const WidenJob job) {
  const float *src = job.src[blockIdx.y];
  double *dst = job.dst[blockIdx.y];
  const size_t count = job.count[blockIdx.y];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = (double)src[i];
}

}  // namespace

hipError_t launch_widen5(const float *const src[5], double *const dst[5], const size_t count[5],
                         hipStream_t stream) {
  WidenJob job;
  size_t most = 0;
  for (int k = 0; k < 5; ++k) {
    job.src[k] = src[k]; job.dst[k] = dst[k]; job.count[k] = count[k];
    most = count[k] > most ? count[k] : most;
  }
  if (most == 0) return hipSuccess;
  const size_t want = (most + 255) / 256;
  const int blocks = (int)(want < 1024 ? want : 1024);
  hipLaunchKernelGGL(widen_kernel, dim3(blocks, 5), dim3(256), 0, stream, job);
  return hipGetLastError();
}

size_t generic_lds_bytes(int n, int m, int nc) {
  const size_t N = (size_t)n + m;
  // + one panel of slack: the last panel's broadcast reads run up to 15 rows past the matrix
  // + the command words of the multi-wave workgroups (ctrl_word)
  // four-wave shapes (N > 128): + 2N doubles of scratch for the MFMA G build ([-b; b_eq] in, x_c
  // out).  Not for the one-wave shapes: the walking shape fits four workgroups per CU by < 1 KB.
  const size_t scratch = (N > 2 * kWave) ? 2 * N : 0;
  return sizeof(double) * ((size_t)packed_doubles((int)N) + (size_t)nc + 2 + kPanel + kCtrlWords + scratch);
}

long long generic_workspace_doubles(int n, int m, int nc, size_t lds_limit) {
  if (generic_lds_bytes(n, m, nc) <= lds_limit) return 0;
  const long long N = (long long)n + m;
  return (long long)packed_doubles((int)N) + 8;
}

template <class TIN>
static hipError_t launch_generic_blocks(const KernelArgs &a_in, size_t lds_limit, int blocks, hipStream_t stream) {
  KernelArgs a = a_in;
  const int N = a.n + a.m;
  if (N > kGenericMaxN) return hipErrorInvalidValue;
  if (a.batch <= 0 || blocks <= 0) return hipSuccess;
  const bool in_lds = generic_lds_bytes(a.n, a.m, a.nc) <= lds_limit;
  size_t lds = in_lds ? generic_lds_bytes(a.n, a.m, a.nc) : sizeof(double) * ((size_t)a.nc + 2);
  if (a.inst_list && in_lds && N <= 2 * kWave) {
    // list mode (COD pass) of a one-wave shape: the COD's work matrices in LDS if they fit
    const size_t cod_bytes = sizeof(double) * (size_t)cod_workspace_doubles(a.n, a.m);
    if (cod_bytes <= lds_limit) {
      if (cod_bytes > lds) lds = cod_bytes;
      a.cod_stride = -1;
    }
  }
  const int slots = slots_for(N > 0 ? N : 1);
  switch (slots) {
    case 1: return launch_slots<1, TIN>(a, in_lds, lds, blocks, stream);
    case 2: return launch_slots<2, TIN>(a, in_lds, lds, blocks, stream);
    case 3: return launch_slots<3, TIN>(a, in_lds, lds, blocks, stream);
    case 4: return launch_slots<4, TIN>(a, in_lds, lds, blocks, stream);
    default: return launch_slots<8, TIN>(a, in_lds, lds, blocks, stream);     // 256 < N <= 512 (kGenericMaxN)
  }
}

hipError_t launch_generic(const KernelArgs &a, size_t lds_limit, hipStream_t stream, bool qa_f32) {
  return qa_f32 ? launch_generic_blocks<float>(a, lds_limit, a.batch, stream)
                : launch_generic_blocks<double>(a, lds_limit, a.batch, stream);
}

hipError_t launch_generic_list(const KernelArgs &a, size_t lds_limit, int blocks, hipStream_t stream,
                               bool qa_f32) {
  return qa_f32 ? launch_generic_blocks<float>(a, lds_limit, blocks, stream)
                : launch_generic_blocks<double>(a, lds_limit, blocks, stream);
}

long long cod_workspace_doubles_host(int n, int m) { return cod_workspace_doubles(n, m); }

}  // namespace fccqp
