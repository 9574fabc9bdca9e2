// fccqp_coop_impl.h -- shape-specialised gfx950 kernel: ONE WORKGROUP OF W WAVEFRONTS PER QP, the 16 x 16 register
// tiles of fccqp_fused_impl.h shared out over the waves.  For the shapes the one-wave kernel cannot hold (num_vars
// >= 64: BASELINE configs[3], FCCQP(114, 60, 24, 66), needs 126 live tiles = 1008 registers per lane) and for small
// batches of the shapes it can (one QP then runs on four SIMDs instead of one).
//
// Hot path replaced (reference /root/reference): FCCQP::Solve src/fcc_qp.cpp:114-191, FCCQP::DoADMM :57-112,
// projections src/constraint_utils.cpp:5-65, Eigen LDLT call sites fcc_qp.cpp:68,86,161,174.
//
// Same algebra, same elimination order and pivot rules as the one-wave kernel (read its header first):
//   P0  K_rho from HBM straight into accumulator-layout tiles; tile COLUMN J of the upper triangle belongs to wave
//       J mod W, right-hand-side tile column cb of [e_0 .. e_{n-1} | -b; b_eq] to wave cb mod W.  Every wave runs its
//       own fully unrolled specialisation of the code (the kernel switches on the wave index once), so tile indices
//       stay compile-time constants and a wave's tiles stay in its registers from the first load to the Gram product.
//   P1  right-looking U^T D U by row panels.  Panel p: the owner of column p factorises the diagonal tile (tile_ldl)
//       and publishes L^-1 and 1/d in LDS | barrier | every wave solves its own tiles of row p (V_pj = L^-1 C_pj,
//       Y_p = L^-1 X_p) and publishes V_pj (2 KB per tile, stored and re-read lane for lane: register r of a tile IS
//       the MFMA A operand of k-step r) | barrier | every wave updates its own columns,
//       C_ij -= V_pi^T D^-1 V_pj, X_i -= V_pi^T D^-1 Y_p, with V_pi read from LDS.  Two barriers per panel, the
//       published tiles double-buffered by panel parity.
//   P3  Gram product G = Y^T D^-1 Y: a sum over the row blocks of Y, and row block p is final when panel p has solved
//       it -- its term is added inside panel p (Y_p published beside V_p, same barriers; round 5).  rho G goes through
//       an n x n LDS image to the ADMM layout, W = I - rho G stays in registers.
//   Pw  presolve x0 = W^-1 x_c (Woodbury, as in the one-wave kernel).  num_vars > 64: by conjugate gradients on the rows
//       of rho G the ADMM waves hold (rs_cg_presolve); the direct path -- W factorised by the same panel scheme, its
//       factor written packed to LDS, the two triangular sweeps by wave 0 -- is the fallback, and the only path for
//       num_vars <= 64.
//   P4  ADMM iterations.  Rows (variables) live in lanes, 64 per "row half" (RH = 1 or 2 halves), the columns of
//       rho G are split over CSA = min(W / RH, 4 / RH) waves per half (one ADMM wave per SIMD; the others have left
//       the kernel by then): every wave holds a 64 x (n / CSA) block of rho G in
//       registers, forms its partial mat-vec with DPP-fed FMAs, the partial sums meet in LDS (barrier), and every
//       wave of a half repeats the (cheap) projections for that half's rows, so only w crosses waves again
//       (second barrier, not needed when RH == 1).
//       num_vars > 64: the row-split layout (RowSplit below) -- four ADMM waves own 32 slots of variables each, the lane
//       halves split the columns: one exchange and one barrier per iteration.
//   P5  metrics and write-back.
// Two-launch form (num_vars > 64, FCCQP_KERNEL_COOP_SPLIT): coop_kernel<.., 1> = P0 .. P3 at one QP per CU, hand-over of
// the ADMM waves' rows of rho G through a device workspace (SplitWorkspace), coop_admm_kernel = Pw (conjugate gradients)
// + P4 + P5 at three QPs per CU, coop_kernel<.., 2> = the whole solve again for the instances whose conjugate gradients
// declined.  Bit-identical to the one-launch form.
// Not bitwise equal to the one-wave kernel (the mat-vec sums its columns in another order); both are within 1e-12
// of the oracle on the synthetic workloads.
#pragma once
#include "fccqp_fused_impl.h"

#ifdef FCCQP_TICKET_DISPATCH
#error "the ticketed block -> instance experiment (fccqp_device.h) draws its ticket in wave 0 only: not wired into the multi-wave cooperative kernel"
#endif

namespace fccqp {
namespace coop {
using namespace fused;

__device__ __forceinline__ void wg_barrier() {
  // (the LDS wait written out: the compiler's own is not relied on, see agent_release() in fccqp_device.h)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

using lds_i32 = __attribute__((address_space(3))) int;

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ---- ADMM layout for n > 64 ("row split"): FOUR ADMM waves, each owning a quarter of the variables (32 slots, one
// per lane of the lower half of the wave, mirrored in the upper half); the two 32-lane halves of a wave split the
// COLUMNS of rho G: the lower half multiplies the variables of waves 0 and 1, the upper half those of waves 2 and 3,
// and one v_permlane32_swap pair adds the halves.  Against splitting the columns over waves (the layout of the
// shapes with n <= 64, and of this one through round 3) no partial sum crosses a wave any more: ONE exchange of w and
// ONE barrier per iteration instead of two of each.  Slot q of wave a is position p = 32 a + q.  Up to two contacts
// per wave, contact j of the wave in slots 16 j .. 16 j + 2 (one contact per 16-lane row: a force component reaches
// its lanes by one v_mov_b64_dpp); the other variables fill the remaining slots in increasing order, the waves'
// shares as even as they come.  A struct of its own: Shape takes constants from its functions.
template <int N_VARS, int NC, int LC_START>
struct RowSplit {
  static constexpr int n = N_VARS, nc = NC, s = LC_START, kContacts = NC / 3;
  static constexpr int cap(int a) { return n / 4 + ((a < n % 4) ? 1 : 0); }
  static constexpr int wave_contacts(int a) { return (kContacts - 2 * a >= 2) ? 2 : ((kContacts - 2 * a >= 1) ? 1 : 0); }
  static constexpr int wave_others(int a) { return cap(a) - 3 * wave_contacts(a); }
  static constexpr int others_before(int a) {
    int k = 0;
    for (int q = 0; q < a; ++q) k += wave_others(q);
    return k;
  }
  static constexpr bool slot_is_cone(int a, int q) { return (q >> 4) < wave_contacts(a) && (q & 15) < 3; }
  // slot q of wave a -> variable, -1: none (no loops: the kernel evaluates it per lane with `a` a compile-time constant)
  static constexpr int slot_variable(int a, int q) {
    const int nca = wave_contacts(a), j = q >> 4, t = q & 15;
    const int c0 = (nca >= 1) ? ((q < 3) ? q : 3) : 0;
    const int c1 = (nca >= 2) ? ((q < 16) ? 0 : ((q - 16 < 3) ? q - 16 : 3)) : 0;
    const int oi = q - c0 - c1;
    const int k = others_before(a) + oi;
    const int other = (oi < wave_others(a)) ? ((k < s) ? k : k + nc) : -1;
    return (j < nca && t < 3) ? s + 3 * (2 * a + j) + t : other;
  }
  static constexpr int pos_variable(int p) { return slot_variable(p >> 5, p & 31); }
  static constexpr int max_slot() {
    int m = 0;
    for (int p = 0; p < 128; ++p) if (pos_variable(p) >= 0 && (p & 31) > m) m = p & 31;
    return m;
  }
  static constexpr bool every_variable_once() {
    for (int v = 0; v < n; ++v) {
      int hits = 0;
      for (int p = 0; p < 128; ++p) hits += (pos_variable(p) == v) ? 1 : 0;
      if (hits != 1) return false;
    }
    return true;
  }
  static constexpr bool fits() { return kContacts <= 8 && wave_others(0) >= 0 && wave_others(1) >= 0 && wave_others(2) >= 0 && wave_others(3) >= 0; }
};

template <int N_VARS, int N_EQ, int NC, int LC_START, int WAVES>
struct Shape {
  static constexpr int n = N_VARS, m = N_EQ, nc = NC, s = LC_START, N = n + m, W = WAVES;
  static constexpr int NRB = (N + 15) / 16;          // row / column blocks of K_rho (identity-padded to whole tiles)
  static constexpr int NXB = (n + 16) / 16;          // column blocks of the right-hand sides [E | r]
  static constexpr int cbd = n / 16;                 // the block that holds the dense column r = [-b; b_eq]
  static constexpr int NW = (n + 15) / 16;           // row / column blocks of G and W
  static constexpr int RH = (n + 63) / 64;           // row halves of the ADMM phase
  static constexpr int CS = W / RH;                  // waves per row half
  // column splits of the ADMM mat-vec: at most one ADMM wave per SIMD.  Every wave of a row half repeats that half's
  // projections, exchange and exit test (~65 instructions against 114 / CS FMAs), and two waves on one SIMD only
  // take turns at its issue slots: with eight waves the loop ran 1460 cycles per iteration, issue-bound on
  // redundant work.  The waves beyond RH * CSA leave the kernel in front of the loop.
  static constexpr int CSA = (RH * CS > 4) ? ((4 / RH > 1) ? 4 / RH : 1) : CS;
  static constexpr int kPos = 64 * RH;               // lane positions (one per variable, some unused)
  // n > 64: the row-split ADMM layout (RowSplit above) -- four ADMM waves, rows over waves, columns over lane halves
  static constexpr bool kRowSplit = RH == 2;
  using RS = RowSplit<N_VARS, NC, LC_START>;
  static constexpr int kRsOdd = kRowSplit ? RS::max_slot() - 15 : 1;     // FMAs of a block of slots 16 .. 31 (slots 0 .. 15: sixteen)
  static constexpr int kRsLen = 2 * (16 + kRsOdd);                        // entries of rho G a lane holds
  static_assert(n >= 1 && n <= 128 && m >= 1, "two row halves at most");
  static_assert(W == 2 || W == 4 || W == 8, "waves per workgroup");
  static_assert(W % RH == 0 && CS >= 1 && CSA >= 1 && CSA <= CS && CSA <= NW, "every column split owns at least one 16-column block");
  static_assert(!kRowSplit || (W >= 4 && RS::fits() && RS::every_variable_once() && kRsOdd >= 1 && kRsOdd <= 15),
                "row-split ADMM layout: four ADMM waves, at most eight contacts, more than 64 variables, slot 31 of every wave free (it carries the exit word)");
  static_assert(nc % 3 == 0 && nc > 0 && s + nc <= n, "fcc_qp.cpp:32-33; shapes without contacts run on the generic kernel");
  static constexpr int kContacts = nc / 3;
  static_assert(4 * kContacts <= kPos, "one quad of lanes per contact");
  using Reduced = ZeroFilledSquare<n>;              // factor of W: zero-filled square, select-free sweeps (fccqp_fused_impl.h)
  static constexpr int ownC(int J) { return J % W; }
  // (round 5, measured and dropped: pairing the right-hand-side column owners on the SIMDs by their Gram-product load --
  // (0, 7), (1, 6), (2, 5), (3, 4): nine accumulator tiles per SIMD where the round-robin order has 6 / 8 / 10 / 12 --
  // shortens the Gram product by 1.3 us per QP and lengthens the factorisation, whose balance it upsets, by 4)
  static constexpr int ownX(int cb) { return cb % W; }
  // column split c of the mat-vec owns the 16-column blocks [qlo(c), qlo(c + 1))
  static constexpr int qlo(int c) { return (NW * c) / CSA; }
  // position (row half * 64 + lane) -> variable: contact c in positions 4c .. 4c+2, the other variables in the
  // fourth position of those quads, then behind the contacts, in increasing order; -1: no variable
  __device__ static __forceinline__ int pos_variable(const int pos) {
    const int q = pos >> 2, r = pos & 3;
    const bool in_quad = q < kContacts;
    const int k = in_quad ? q : pos - 3 * kContacts;
    const int other = (k < s) ? k : k + nc;
    return (in_quad && r < 3) ? s + 3 * q + r : ((k < n - nc) ? other : -1);
  }
  // ---- LDS carve (doubles) ----
  // diagonal tile exchange: the tile on its way to the column-per-lane layout (owner only), then L^-1 and 1/d in
  // TWO slots by panel parity: the owner of column p + 1 factorises its diagonal tile while the others are still in
  // the trailing update of panel p (look-ahead), whose L^-1 / 1/d must stay readable until every wave has passed on
  // (tile columns 18 doubles apart, as in the one-wave kernel: a 128-byte stride puts a row of the tile on one LDS bank)
  static constexpr int kTS = 18, kTL = 16 * kTS;
  static constexpr int kSblk = 0, kSinv = kTL, kSrcp = 3 * kTL;        // Sinv[2][kTL], Srcp[2][16]
  static constexpr int kRcp = 3 * kTL + 32;                             // 1/d of every row of K_rho (16 NRB)
  static constexpr int kRhs = kRcp + 16 * NRB;                          // [-b; b_eq]; later x_c, x0
  static constexpr int kRhsLen = cmax(16 * NRB, kPos + 64);
  static constexpr int kRed = kRhs + kRhsLen;                           // end-of-solve reductions [RH][4]
  static constexpr int kRed2 = kRed;                                    // ... row-split layout: [4 waves][4]
  static constexpr int kInts = kRed + 16;                               // int words: flags
  static constexpr int kBig = (kInts + 16 + 1) & ~1;
  static constexpr int kTile = 256;                                     // doubles per published tile
  static constexpr int kVbufLen = 2 * NRB * kTile;                      // V_pj of two panels
  static constexpr int kYbufLen = 2 * NXB * kTile;                      // Y[K][.] of two row blocks
  static constexpr int kImage = (n * n + 1) & ~1;                       // rho G, row-major
  static constexpr int kWfac = (Reduced::kPacked + 1) & ~1;             // packed factor of W ...
  static constexpr int kWfacLen = kWfac + 2 * NW * kTile;               // ... and the tiles its panels publish
  // ADMM exchange: partial sums [2][CS][kPos] (by iteration parity), then w by variable (one copy, or one per wave when RH == 1), then the
  // per-iteration exit words
  static constexpr int kWv = 16 * NW + 16;                              // (+ a dummy slot for lanes without a variable)
  // (row split: w by position, [2][128] doubles by iteration parity)
  static constexpr int kRsW = 0;
  static constexpr int kPart = 0, kWvOff = 2 * CS * kPos, kAdmmLen = kRowSplit ? kRsW + 256 + 64 : kWvOff + ((RH == 1) ? W : 1) * kWv;
  static constexpr int kBigLen = cmax(kVbufLen + kYbufLen, cmax(cmax(kImage, kWfacLen), kAdmmLen));   // (V and Y tiles side by side: the fused Gram product)
  static constexpr int kLdsDoubles = kBig + kBigLen;
  static_assert(kLdsDoubles * 8 <= 160 * 1024, "LDS of one CU");
};

// ---- P0: the own columns of the upper triangle of K_rho, from HBM into accumulator-layout tiles ----------------
// (element map and clamping as issue_kkt_loads / convert_kkt_tiles of the one-wave kernel, without a border)
template <class SH, int WV, class TIN>
__device__ __forceinline__ void load_own_tiles(const TIN *__restrict__ Q, const TIN *__restrict__ A, const double rho,
                                               const int lane, d4 (&C)[SH::NRB][SH::NRB]) {
  constexpr int N = SH::N, n = SH::n, NRB = SH::NRB;
  const int col = lane & 15, grp = lane >> 4;
  TIN raw[NRB][NRB][4];
#pragma unroll
  for (int I = 0; I < NRB; ++I) {
    if (16 * I < n) {
#pragma unroll
      for (int J = I; J < NRB; ++J) {
        if (SH::ownC(J) == WV) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * I + grp + 4 * r, j = 16 * J + col;
            const int lo = (I == J && j < i) ? j : i, hi = (I == J && j < i) ? i : j;
            const int loc = (lo < n) ? lo : n - 1;
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
#pragma unroll
  for (int I = 0; I < NRB; ++I) {
#pragma unroll
    for (int J = I; J < NRB; ++J) {
      if (SH::ownC(J) == WV) {
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
}

// publish / fetch one tile: register r of lane l at [256 t + 64 r + l]
__device__ __forceinline__ void tile_publish(lds_f64 *buf, const int lane, const d4 &t) {
#pragma unroll
  for (int r = 0; r < 4; ++r) buf[64 * r + lane] = t[r];
}
__device__ __forceinline__ d4 tile_fetch(const lds_f64 *buf, const int lane) {
  d4 t;
#pragma unroll
  for (int r = 0; r < 4; ++r) t[r] = buf[64 * r + lane];
  return t;
}

// diagonal tile of a panel: the owner factorises it and leaves L^-1 (strictly lower part) and 1/d in the exchange area
template <int kValid, bool WITH_INVERSE, bool KEEP_L>
__device__ __forceinline__ bool diag_tile(lds_f64 *L, const int lane, const d4 &Cpp, double (&T)[16], double &rmine, const double max_rcp,
                                          const int slot) {
  const int col = lane & 15, grp = lane >> 4;
  constexpr int TS = 18, TL = 16 * TS;      // (Shape::kTS, kTL)
  lds_f64 *const Sblk = L + 0, *const Sinv = L + TL + TL * slot, *const Srcp = L + 3 * TL + 16 * slot;
#pragma unroll
  for (int r = 0; r < 4; ++r) Sblk[TS * col + grp + 4 * r] = Cpp[r];
  lds_fence();
  double Y[16];
  const lds_d2 *colp = reinterpret_cast<const lds_d2 *>(Sblk + TS * col);
  const bool bad = tile_ldl<kValid, WITH_INVERSE, KEEP_L>(T, Y, rmine, col, colp, max_rcp);
  if (lane < 16) {
    if constexpr (WITH_INVERSE) {
      lds_d2 *outp = reinterpret_cast<lds_d2 *>(Sinv + TS * col);
#pragma unroll
      for (int ip = 0; ip < 8; ++ip) outp[ip] = d2{Y[2 * ip], Y[2 * ip + 1]};
    }
    Srcp[col] = rmine;
  }
  return bad;
}

// which steps of a panel's trailing work (coop_factor_forward: `trailing`) wave WV takes part in -- step t < nV: row block
// P + 1 + t of the trailing update, step nV + I: row block I of the panel's Gram term
template <class SH, int WV, int P, bool kOwnNext>
struct TrailingSteps {
  static constexpr int NRB = SH::NRB, NXB = SH::NXB, cbd = SH::cbd;
  static constexpr int nV = NRB - 1 - P, nY = SH::NW, nT = nV + nY;
  static constexpr bool zero_tile(int K, int cb) { return K < cb && cb != cbd; }
  static constexpr bool needed(const int t) {
    if (t < nV) {
      const int I = P + 1 + t;
      bool any = false;
      for (int J = I; J < NRB; ++J) any = any || (SH::ownC(J) == WV && !(kOwnNext && I == P + 1 && J == P + 1));
      for (int cb = 0; cb < NXB; ++cb) any = any || (SH::ownX(cb) == WV && !zero_tile(P, cb));
      return any;
    }
    const int I = t - nV;
    if (zero_tile(P, I)) return false;
    bool any = false;
    for (int J = I; J <= cbd; ++J) any = any || (SH::ownX(J) == WV && !zero_tile(P, J));
    return any;
  }
  static constexpr int next_needed(const int t) {
    int u = t;
    while (u < nT && !needed(u)) ++u;
    return u;
  }
};

// ---- P1: K_rho = U^T D U with the forward solve of [E | r] fused in, tile columns shared out over the waves ----
// The Gram product G = Y^T D^-1 Y is a sum over the row blocks of Y, and row block P of Y is final the moment
// panel P has solved it -- so its term is added right there: the owners publish Y_p beside V_p (same barrier), and every
// wave adds (Y_p[.][I])^T D_p^-1 Y_p[.][J] to the accumulator tiles Gt[I][J] of its own columns J behind the trailing
// update.  Against a Gram product of its own behind the factorisation (coop_gram: publish + barrier per row block)
// eleven barriers and their LDS round trips go, the matrix-core work fills waits of the panels, and a row block of Y
// is dead as soon as its panel ends.
template <class SH, int WV>
__device__ __forceinline__ void coop_factor_forward(lds_f64 *L, const int lane, d4 (&C)[SH::NRB][SH::NRB],
                                                    d4 (&X)[SH::NRB][SH::NXB], d4 (&Gt)[SH::NW][SH::NXB]) {
  constexpr int N = SH::N, n = SH::n, NRB = SH::NRB, NXB = SH::NXB, cbd = SH::cbd;
  const int col = lane & 15, grp = lane >> 4;
  lds_f64 *const Sinv = L + SH::kSinv, *const Srcp = L + SH::kSrcp, *const rhs = L + SH::kRhs;
  lds_f64 *const Vbuf = L + SH::kBig;
  lds_f64 *const Yfbuf = L + SH::kBig + SH::kVbufLen;          // Y_p of two panels, behind the V tiles
  static_assert(SH::kVbufLen + SH::kYbufLen <= SH::kBigLen, "room for the published Y tiles");
  constexpr int IMAX = SH::NW - 1;                             // last row block of G that is needed
#pragma unroll
  for (int J = 0; J <= cbd; ++J)
    if (SH::ownX(J) == WV)
#pragma unroll
      for (int I = 0; I <= ((J < IMAX) ? J : IMAX); ++I) Gt[I][J] = d4{0.0, 0.0, 0.0, 0.0};
  lds_i32 *const iflags = (lds_i32 *)(L + SH::kInts);
  auto zero_tile = [](int K, int cb) { return K < cb && cb != cbd; };
  auto kfirst = [](int cb) { return (cb == cbd) ? 0 : cb; };
  auto x_initial = [&](const int I, const int cb) {
    d4 t;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * I + grp + 4 * r, c = 16 * cb + col;
      double v = 0.0;
      if (I == cb) v = (row == c && c < n) ? 1.0 : 0.0;
      if (cb == cbd) v = (c == n) ? ((row < N) ? rhs[row < N ? row : 0] : 0.0) : v;
      t[r] = v;
    }
    return t;
  };
  // the diagonal tile of panel Pn, by its owner: L^-1 and 1/d into slot Pn & 1, 1/d also into the per-row table
  auto factor_diag = [&](auto pn) {
    constexpr int Pn = decltype(pn)::value;
    constexpr int kValid = (N - 16 * Pn < 16) ? N - 16 * Pn : 16;
    double T[16], rmine;
    const bool bad = diag_tile<kValid, true, false>(L, lane, C[Pn][Pn], T, rmine, 0.0, Pn & 1);
    if (lane < 16) L[SH::kRcp + 16 * Pn + col] = rmine;
    if (bad && lane == 0) iflags[0] = 1;
  };
  // ---- the trailing work of panel P as ONE stream of A operands out of LDS: the published tiles V_p[.][I] of the rows
  // I > P (trailing update of the own columns of C and X), then the published tiles Y_p[.][I] (the panel's term of the
  // Gram product in the own accumulator tiles).  Step t's tile is fetched while step t - 1 computes: left to the
  // compiler every row waited for its own LDS round trip (ds_read, s_waitcnt, then the MFMAs), which two waves per SIMD
  // only partly hide.  `skip_first`: the owner of column P + 1 has updated and factorised its diagonal tile already.
  auto trailing = [&](auto pc, auto skipc, const double (&nrv)[4]) {
    constexpr int P = decltype(pc)::value;
    constexpr bool kOwnNext = decltype(skipc)::value != 0;
    constexpr int nV = NRB - 1 - P, nY = IMAX + 1, nT = nV + nY;
    using TS = TrailingSteps<SH, WV, P, kOwnNext>;
    auto needed = [](const int t) constexpr { return TS::needed(t); };
    auto next_needed = [](const int t) constexpr { return TS::next_needed(t); };
    auto a_tile = [&](const int t) {
      return (t < nV) ? Vbuf + ((P & 1) * NRB + (P + 1 + t)) * SH::kTile : Yfbuf + ((P & 1) * NXB + (t - nV)) * SH::kTile;
    };
    constexpr int t0 = next_needed(0);
    d4 nxt = d4{0.0, 0.0, 0.0, 0.0};
    if constexpr (t0 < nT) nxt = tile_fetch(a_tile(t0), lane);
    static_for<0, nT>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      if constexpr (t < nV) {
        // (right-hand-side tiles that come into being with this panel)
        constexpr int I = P + 1 + t;
#pragma unroll
        for (int cb = 0; cb < NXB; ++cb)
          if (SH::ownX(cb) == WV && !zero_tile(P, cb) && P == kfirst(cb)) X[I][cb] = x_initial(I, cb);
      }
      if constexpr (needed(t)) {
        const d4 a = nxt;
        constexpr int tn = next_needed(t + 1);
        if constexpr (tn < nT) {
          nxt = tile_fetch(a_tile(tn), lane);
#ifndef FCCQP_COOP_NO_PREFETCH
          __builtin_amdgcn_sched_barrier(0);           // (the fetch stays in front of this step's MFMAs)
#endif
        }
        if constexpr (t < nV) {
          constexpr int I = P + 1 + t;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (16 * P + 4 * r < N) {
              const double aop = a[r] * nrv[r];        // A[i][k] = -(V_pi[4r + k][i] / d_{4r + k})
#pragma unroll
              for (int J = I; J < NRB; ++J)
                if (SH::ownC(J) == WV && !(kOwnNext && I == P + 1 && J == P + 1))
                  C[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, C[P][J][r], C[I][J], 0, 0, 0);
#pragma unroll
              for (int cb = 0; cb < NXB; ++cb)
                if (SH::ownX(cb) == WV && !zero_tile(P, cb))
                  X[I][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, X[P][cb][r], X[I][cb], 0, 0, 0);
            }
          }
        } else {
          constexpr int I = t - nV;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (16 * P + 4 * r < N) {
              const double aop = a[r] * -nrv[r];
#pragma unroll
              for (int J = I; J <= cbd; ++J)
                if (SH::ownX(J) == WV && !zero_tile(P, J))
                  Gt[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, X[P][J][r], Gt[I][J], 0, 0, 0);
            }
          }
        }
      }
    });
  };
  if constexpr (SH::ownC(0) == WV) factor_diag(int_c<0>{});
  wg_barrier();
  static_for<0, NRB>([&](auto pc) {
    constexpr int P = decltype(pc)::value;
    double Ainv[4], nrv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double li = Sinv[SH::kTL * (P & 1) + SH::kTS * (4 * r + grp) + col];
      Ainv[r] = (col > 4 * r + grp) ? li : 0.0;
      nrv[r] = -Srcp[16 * (P & 1) + grp + 4 * r];
    }
    // (2) the own tiles of row panel P: V_pj = L^-1 C_pj (published), Y_p = L^-1 X_p
#pragma unroll
    for (int J = P + 1; J < NRB; ++J) {
      if (SH::ownC(J) == WV) {
        d4 t = C[P][J];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * P + 4 * r < N) t = __builtin_amdgcn_mfma_f64_16x16x4f64(Ainv[r], C[P][J][r], t, 0, 0, 0);
        C[P][J] = t;
        tile_publish(Vbuf + ((P & 1) * NRB + J) * SH::kTile, lane, t);
      }
    }
#pragma unroll
    for (int cb = 0; cb < NXB; ++cb) {
      if (SH::ownX(cb) == WV && !zero_tile(P, cb)) {
        if (P == kfirst(cb)) X[P][cb] = x_initial(P, cb);
        d4 t = X[P][cb];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * P + 4 * r < N) t = __builtin_amdgcn_mfma_f64_16x16x4f64(Ainv[r], X[P][cb][r], t, 0, 0, 0);
        X[P][cb] = t;
        tile_publish(Yfbuf + ((P & 1) * NXB + cb) * SH::kTile, lane, t);
      }
    }
    if constexpr (P + 1 == NRB) {
      wg_barrier();
      trailing(pc, int_c<0>{}, nrv);
    }
    if constexpr (P + 1 < NRB) {
      wg_barrier();
      // (3) the rows below, own columns only; the A operand V_pi comes from the published tiles.
      // Look-ahead: the owner of column P + 1 finishes ITS diagonal tile first and factorises it at once, beside the
      // other waves' trailing updates.  (Measured: neutral, 60.4 us either way for the large shape -- the owner's own
      // remaining updates still queue behind its 1.5 us diagonal tile and the panel barrier waits for them; a gain
      // would need point-to-point flags instead of the barrier.  Kept: it costs nothing and frees 13 registers.)
      constexpr bool kOwnNext = SH::ownC(P + 1) == WV;
      if constexpr (kOwnNext) {
        const d4 v = tile_fetch(Vbuf + ((P & 1) * NRB + (P + 1)) * SH::kTile, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * P + 4 * r < N)
            C[P + 1][P + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[r] * nrv[r], C[P][P + 1][r], C[P + 1][P + 1], 0, 0, 0);
        factor_diag(int_c<P + 1>{});
      }
      trailing(pc, int_c<kOwnNext ? 1 : 0>{}, nrv);
      wg_barrier();      // the next diagonal tile is published; every wave is done with the V tiles of panel P
    }
  });
}

// ---- P3, output half: rho G and x_c out of the accumulator tiles of the Gram product Y^T D^-1 Y (own tile columns J:
// the owner of X column J).  Leaves rho G in the LDS image (row-major n x n), x_c in xcv[0..n), and the own upper tiles
// of W = I - rho G in Wt.
// (The accumulation itself lives in the panels of coop_factor_forward since round 5; through round 4 it was a loop of
// its own here -- publish row block K of Y, barrier, accumulate -- 17.4 us per QP for the large shape against 4.5 us
// for this output half plus 8 us more in the factorisation.)
template <class SH, int WV>
__device__ __forceinline__ void coop_gram_output(lds_f64 *L, const int lane, const double rho, const d4 (&Gt)[SH::NW][SH::NXB],
                                                 d4 (&Wt)[SH::NW][SH::NW]) {
  constexpr int n = SH::n, NW = SH::NW, cbd = SH::cbd;
  const int col = lane & 15, grp = lane >> 4;
  lds_f64 *const img = L + SH::kBig, *const xcv = L + SH::kRhs;
  constexpr int IMAX = NW - 1;                      // last row block of G that is needed
  wg_barrier();      // every wave is done with the published Y tiles: the image goes over them
#pragma unroll
  for (int J = 0; J <= cbd; ++J) {
    if (SH::ownX(J) == WV) {
#pragma unroll
      for (int I = 0; I <= ((J < IMAX) ? J : IMAX); ++I) {
        const bool interior = (16 * I + 15 < n) && (16 * J + 15 < n);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * I + grp + 4 * r, c = 16 * J + col;
          const double gv = Gt[I][J][r];
          const double g = rho * gv;
          const bool inside = interior || (row < n && c < n);
          if (inside) {
            img[row * n + c] = g;
            if (J != I) img[c * n + row] = g;
          }
          if (J == cbd && c == n && row < n) xcv[row] = gv;
          if (J < NW) {
            const double unit = (I == J && row == c) ? 1.0 : 0.0;
            if (interior) Wt[I][J][r] = unit - g;
            else Wt[I][J][r] = select_lanes(__builtin_amdgcn_ballot_w64(inside), unit - g, unit);   // (VOP3 select)
          }
        }
      }
    }
  }
  wg_barrier();
}

// ---- Pw: W = U^T D U (n x n) by the same panel scheme (with the same look-ahead); the factor goes packed to LDS
// for the two sweeps ------------------------------------------------------------------------------------------
template <class SH, int WV>
__device__ __forceinline__ void coop_factor_store(lds_f64 *L, const int lane, const double min_pivot, d4 (&C)[SH::NW][SH::NW]) {
  using SW = typename SH::Reduced;
  constexpr int n = SH::n, NW = SH::NW;
  const int col = lane & 15, grp = lane >> 4;
  lds_f64 *const M = L + SH::kBig;                          // packed factor
  lds_f64 *const Vbuf = L + SH::kBig + SH::kWfac;
  lds_f64 *const Sblk = L + SH::kSblk, *const Sinv = L + SH::kSinv, *const Srcp = L + SH::kSrcp;
  lds_i32 *const iflags = (lds_i32 *)(L + SH::kInts);
  if constexpr (SW::kZeroFilled) {
    // zeros on and above the diagonal of the square: every wave clears a slice of it (the rho G image it lies over is
    // dead: barrier in front of this function), L then goes on top
    static_assert(SW::kDinv % 2 == 0, "whole 16-byte stores");
    lds_d2 *const z2 = reinterpret_cast<lds_d2 *>(M);
    constexpr int kPairs = SW::kDinv / 2, kPerWave = (kPairs + SH::W - 1) / SH::W;
#pragma unroll
    for (int k = 0; k < (kPerWave + 63) / 64; ++k) {
      const int idx = WV * kPerWave + 64 * k + lane;
      if (64 * k + lane < kPerWave && idx < kPairs) z2[idx] = d2{0.0, 0.0};
    }
    wg_barrier();
  }
  // diagonal tile of panel Pn, by its owner: column 16 Pn + col of L (tile_ldl leaves T = L d), 1/d in the diagonal slots
  // (the slots behind the square in the zero-filled layout)
  auto factor_diag = [&](auto pn) {
    constexpr int Pn = decltype(pn)::value;
    constexpr int kValid = (n - 16 * Pn < 16) ? n - 16 * Pn : 16;
    constexpr bool kLast = (Pn + 1 == NW);
    double T[16], rmine;
    const bool bad = diag_tile<kValid, !kLast, true>(L, lane, C[Pn][Pn], T, rmine, 1.0 / min_pivot, Pn & 1);
    if (lane < 16 && col < kValid) {
      const int jcol = 16 * Pn + col, cb = SW::cbase(jcol);
      M[SW::kZeroFilled ? SW::kDinv + jcol : cb + jcol] = rmine;
      // (above the diagonal: a dummy slot.  The address select written out as VOP3 on a ballot, as in the one-wave
      // kernel: the compiler's `v_cmp -> VCC, s_nop 1, v_cndmask_b32_e32` costs 32 cycles per store on this -- serial --
      // path, this 10; `colv`: hidden from CSE, or the masks are hoisted out of the panels and spilled)
      int colv = col;
      asm volatile("" : "+v"(colv));
      const int below0 = (int)(M - L) + cb + 16 * Pn, dummy = SH::kSblk + colv;
#pragma unroll
      for (int i = 1; i < kValid; ++i) {
        const int idx = select32_lanes(__builtin_amdgcn_ballot_w64(i > colv), below0 + i, dummy);
        L[idx] = T[i] * rmine;
      }
    }
    if (bad && lane == 0) iflags[1] = 1;
  };
  if constexpr (SH::ownX(0) == WV) factor_diag(int_c<0>{});
  wg_barrier();
  static_for<0, NW>([&](auto pc) {
    constexpr int P = decltype(pc)::value;
    if constexpr (P + 1 < NW) {
      double Ainv[4], rv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double li = Sinv[SH::kTL * (P & 1) + SH::kTS * (4 * r + grp) + col];
        Ainv[r] = (col > 4 * r + grp) ? li : 0.0;
        rv[r] = Srcp[16 * (P & 1) + grp + 4 * r];
      }
#pragma unroll
      for (int J = P + 1; J < NW; ++J) {
        if (SH::ownX(J) == WV) {
          d4 t = C[P][J];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * P + 4 * r < n) t = __builtin_amdgcn_mfma_f64_16x16x4f64(Ainv[r], C[P][J][r], t, 0, 0, 0);
          C[P][J] = t;
          tile_publish(Vbuf + ((P & 1) * NW + J) * SH::kTile, lane, t);
          // L[16 J + col][16 P + row] = U_pj[row][col] = V_pj[row][col] / d_row
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int jrow = 16 * P + grp + 4 * r, irow = 16 * J + col;
            if (16 * J + 15 < n || irow < n) M[SW::cbase(jrow) + irow] = t[r] * rv[r];
          }
        }
      }
      wg_barrier();
      constexpr bool kOwnNext = SH::ownX(P + 1) == WV;
      if constexpr (kOwnNext) {              // look-ahead: the next diagonal tile first, then its factorisation
        const d4 v = tile_fetch(Vbuf + ((P & 1) * NW + (P + 1)) * SH::kTile, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          C[P + 1][P + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[r] * -rv[r], C[P][P + 1][r], C[P + 1][P + 1], 0, 0, 0);
        factor_diag(int_c<P + 1>{});
      }
#pragma unroll
      for (int I = P + 1; I < NW; ++I) {
        bool any = false;
#pragma unroll
        for (int J = I; J < NW; ++J) any = any || (SH::ownX(J) == WV && !(kOwnNext && I == P + 1 && J == P + 1));
        if (any) {
          const d4 v = tile_fetch(Vbuf + ((P & 1) * NW + I) * SH::kTile, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double aop = v[r] * -rv[r];
#pragma unroll
            for (int J = I; J < NW; ++J)
              if (SH::ownX(J) == WV && !(kOwnNext && I == P + 1 && J == P + 1))
                C[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, C[P][J][r], C[I][J], 0, 0, 0);
          }
        }
      }
      wg_barrier();
    }
  });
}

// x := (L D L^T)^-1 x, rows lane + 64 t (one wave, SLOTS rows per lane), M as left by coop_factor_store.
// Column-oriented substitution; the only dependent chain is v_readlane(x_k) -> fma.  The L operands of the next U
// steps are fetched while the current U run (two register buffers, as solve_lds_one_slot of the one-wave kernel):
// left to the compiler every step waited for its own LDS round trip (12.9 us for n = 114 instead of ~6).
template <class SW, int SLOTS>
__device__ __forceinline__ void solve_lds_slots(const lds_f64 *M, const int lane, double (&x)[SLOTS]) {
  constexpr int N = SW::N, U = 8;
  static_assert(N <= 64 * SLOTS, "SLOTS rows per lane");
  int rc[SLOTS], cbr[SLOTS];
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) {
    rc[t] = (lane + 64 * t < N) ? lane + 64 * t : N - 1;
    cbr[t] = SW::cbase(rc[t]);
  }
  // ---- forward, unit lower: x_i -= L[i][k] x_k, i > k; L[i][k] = M[cbase(k) + i] -------------------------------
  {
    double cur[U][SLOTS], nxt[U][SLOTS];
    auto load_fwd = [&](double (&buf)[U][SLOTS], const int k0) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) buf[u][t] = M[SW::cbase((k0 + u < N) ? k0 + u : N - 1) + rc[t]];
    };
    load_fwd(cur, 0);
#pragma unroll
    for (int k0 = 0; k0 < N - 1; k0 += U) {
      if (k0 + U < N - 1) load_fwd(nxt, k0 + U);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + u;
        if (k < N - 1) {
          const double xk = bcast_lane(x[k / 64], k % 64);
#pragma unroll
          for (int t = 0; t < SLOTS; ++t) {
            if (64 * t + 63 > k) {
              const double a = SW::kZeroFilled ? cur[u][t] : ((lane + 64 * t > k && lane + 64 * t < N) ? cur[u][t] : 0.0);
              x[t] = fma(-a, xk, x[t]);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) cur[u][t] = nxt[u][t];
    }
  }
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) x[t] *= M[SW::kZeroFilled ? SW::kDinv + rc[t] : cbr[t] + rc[t]];          // 1/d (pseudo-inverse rule folded in)
  // ---- backward with L^T: x_i -= L[k][i] x_k, i < k; L[k][i] = M[cbase(i) + k]: this lane's own columns --------
  {
    double cur[U][SLOTS], nxt[U][SLOTS];
    auto load_bwd = [&](double (&buf)[U][SLOTS], const int k0) {      // k = k0, k0 - 1, ...
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) buf[u][t] = M[cbr[t] + ((k0 - u >= 1) ? k0 - u : 1)];
    };
    load_bwd(cur, N - 1);
#pragma unroll
    for (int k0 = N - 1; k0 >= 1; k0 -= U) {
      if (k0 - U >= 1) load_bwd(nxt, k0 - U);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 - u;
        if (k >= 1) {
          const double xk = bcast_lane(x[k / 64], k % 64);
#pragma unroll
          for (int t = 0; t < SLOTS; ++t) {
            if (64 * t < k) {
              const double a = SW::kZeroFilled ? cur[u][t] : ((lane + 64 * t < k) ? cur[u][t] : 0.0);
              x[t] = fma(-a, xk, x[t]);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) cur[u][t] = nxt[u][t];
    }
  }
}

// ---- row-split ADMM layout (Shape::kRowSplit): the mat-vec of one lane half, ONE asm statement (fccqp_fused_impl.h:
// a second statement costs a second s_waitcnt and an s_nop).  Blocks 0 and 2 are slots 0 .. 15 of the half's two
// waves (sixteen FMAs each), blocks 1 and 3 their slots 16 .. 15 + L.  Mrow: [16 | L | 16 | L].
#define FCCQP_MV_RS(L)                                                                                                \
  asm(FCCQP_MV16(0) FCCQP_MV##L(1) FCCQP_MV16(2) FCCQP_MV##L(3)                                                       \
      : [acc] "+v"(acc) : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), FCCQP_MVOPS(0, m0), FCCQP_MVOPS(1, m1), \
        FCCQP_MVOPS(2, m2), FCCQP_MVOPS(3, m3))
template <int L, int NN>
__device__ __forceinline__ void rs_matvec(double &acc, const double (&Wq)[4], const double (&Mrow)[NN]) {
  static_assert(NN == 2 * (16 + L) && L >= 1 && L <= 16, "[16 | L | 16 | L]");
  double m0[16], m1[16], m2[16], m3[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    m0[i] = Mrow[i];
    m1[i] = Mrow[16 + ((i < L) ? i : L - 1)];
    m2[i] = Mrow[16 + L + i];
    m3[i] = Mrow[32 + L + ((i < L) ? i : L - 1)];
  }
  const double w0 = Wq[0], w1 = Wq[1], w2 = Wq[2], w3 = Wq[3];
  FCCQP_MV_DISPATCH(FCCQP_MV_RS);
}
// sum of the two 32-lane halves of a wave, in every lane: v_permlane32_swap of a register with itself leaves the lower
// half's values in both halves of one result and the upper half's in the other
__device__ __forceinline__ double add_lane_halves(const double v) {
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  const v2u lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
  const v2u hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
  return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}

// sum over the lanes of a 16-lane row, in every lane of the row (the first four steps of wave_sum)
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_get<0xB1, 0xf>(v, 0.0);   // quad_perm [1,0,3,2]
  v += dpp_get<0x4E, 0xf>(v, 0.0);   // quad_perm [2,3,0,1]
  v += dpp_get<0x141, 0xf>(v, 0.0);  // row_half_mirror
  v += dpp_get<0x140, 0xf>(v, 0.0);  // row_mirror
  return v;
}

// ---- presolve by conjugate gradients in the row-split layout (round 5) -------------------------------------------
// x0 = W^-1 x_c, W = I - rho G (fcc_qp.cpp:159-178 in its Woodbury form; see cg_presolve of the one-wave kernel for why
// conjugate gradients end in at most n - m + 1 steps here: 29-31 on the large shape's workloads).  The four ADMM waves
// hold the rows of rho G already (Mrow); what an iteration of the textbook method needs beyond the mat-vec is two
// inner products, one behind the other -- two more exchange-and-barrier steps when the vectors live one entry per slot.
// Two things keep an iteration at ONE exchange, ONE barrier and ONE reduction:
//   * EVERY wave keeps the WHOLE of r and q = W p in the layout the mat-vec reads its operand in (four registers per
//     lane: the 64 positions of the lane half's two waves, the same in both 16-lane rows of the half).  The one vector
//     that crosses the waves is w = W r, by the exchange the ADMM loop uses; everything else every wave computes for
//     itself -- redundantly and, the arithmetic being the same instruction sequence on the same data, bit for bit the
//     same in all four waves, which therefore also agree on when to stop;
//   * the Chronopoulos-Gear form of the method: the mat-vec is applied to r, q = W p follows by the recurrence
//     q = w + beta q, and the two inner products of an iteration, r.r and r.w, are available at the same time --
//     one 16-lane row of every lane half sums the one, the other row the other (the rows hold the same data), one
//     row reduction and one addition of the lane halves for both.  (Emulated against the textbook form on the
//     workloads at rho = 5e-5 .. 1: same iteration counts, x0 within 1e-13 .. 1e-9 of the direct solve alike.)
// ~0.4 us per iteration against two barriers and an LDS round trip per PANEL for the factorisation of W plus its two
// sweeps (20.9 + 4.4 us for n = 114).
// The waves that hold no rows (4 .. 7) wait at the same barriers and read the verdict from an LDS word written in
// front of each barrier (two slots by barrier parity): 0 go on, 1 done, 2 declined -- the direct path needs them.
// Declines where the one-wave kernel's does: a diagonal entry of W <= kPresolveMinPivot (published as a NaN in the
// first exchange, so that all waves see it), a Rayleigh quotient p.Wp / p.p <= 1e-6, no convergence to 1e-13 |x_c| in
// n - m + 10 steps.
template <class SH, int WV>
__device__ __forceinline__ bool rs_cg_presolve(lds_f64 *const buf, lds_i32 *const verdict, const int lane, const bool isvar,
                                               const double (&Mrow)[SH::kRsLen], const double xc, const double gdiag, double &x0) {
  // buf: [2][128] doubles by barrier parity (the ADMM loop's exchange area); verdict: [2] words by barrier parity
  constexpr int kCap = (SH::n - SH::m > 0 ? SH::n - SH::m : SH::n) + 10;
  const int half = lane >> 5, slot = lane & 31, pos = 32 * (WV & 3) + slot;
  int nb = 0;                                                  // barriers passed
  auto barrier_with = [&](const int code) {
    if (WV == 0 && lane == 0) verdict[nb & 1] = code;
    wg_barrier();
    ++nb;
  };
  if constexpr (WV >= 4) {
    int code;
    do {
      wg_barrier();
      code = verdict[nb & 1];
      ++nb;
    } while (code == 0);
    return code == 1;
  } else {
    const unsigned long long odd_rows = __builtin_amdgcn_ballot_w64((lane & 16) != 0);
    // publish this slot's entry, pass the barrier, fetch the whole vector in the mat-vec's operand layout
    auto exchange = [&](const double mine, double (&v)[4]) {
      const int par = nb & 1;
      buf[128 * par + pos] = mine;
      barrier_with(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = buf[128 * par + 64 * half + 16 * q + (lane & 15)];
      lds_fence();
    };
    // (a.a, a.b) over all 128 positions: even 16-lane rows sum a.a, odd rows a.b -- both rows of a lane half hold the
    // same 64 entries --, one row reduction, one addition of the halves; lane 0 then holds the one, lane 16 the other
    auto two_dots = [&](const double (&va)[4], const double (&vb)[4], double &aa, double &ab) {
      double d1 = va[0] * va[0], d2 = va[0] * vb[0];
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        d1 = fma(va[q], va[q], d1);
        d2 = fma(va[q], vb[q], d2);
      }
      const double t = add_lane_halves(row16_sum(select_lanes(odd_rows, d2, d1)));
      aa = bcast_lane(t, 0);
      ab = bcast_lane(t, 16);
    };
    auto apply_w = [&](double (&v)[4], const double v_own) {     // (W v)_own = v_own - ((rho G) v)_own
      double acc = 0.0;
      // (v may have been written by VALU instructions: two wait states in front of the mat-vec's DPP reads, which the
      // hazard recogniser does not see inside the asm statement; tests/test_code_objects.py checks the built code)
      asm volatile("s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
      rs_matvec<SH::kRsOdd>(acc, v, Mrow);
      return v_own - add_lane_halves(acc);
    };
    // r0 = x_c (x = 0); a wave that finds a vanishing diagonal entry of W poisons its share of the first exchange
    const bool bad_diag = __builtin_amdgcn_ballot_w64(isvar && !(1.0 - gdiag > kPresolveMinPivot)) != 0;
    double rF[4], wF[4], qF[4];
    double r_own = xc, x_own = 0.0;
    exchange(bad_diag ? __builtin_nan("") : xc, rF);
    double w_own = apply_w(rF, r_own);
    exchange(w_own, wF);
    double gamma, delta;
    two_dots(rF, wF, gamma, delta);
    const double stop = kCgTolSquared * gamma;
    int code = 0;                                               // 1 done, 2 declined
    if (!(gamma > 0.0)) code = (gamma == 0.0) ? 1 : 2;          // x_c == 0: x0 = 0; NaN: decline (uniform: gamma is)
    else if (!(delta > kCgMinRayleigh * gamma)) code = 2;         // (p = r: the Rayleigh quotient rule of cg_presolve)
    double alpha = gamma * fast_rcp(delta);
    double pp = gamma;                                            // p.p by its recurrence (see cg_presolve)
    double p_own = r_own, q_own = w_own;
#pragma unroll
    for (int q = 0; q < 4; ++q) qF[q] = wF[q];
#pragma nounroll
    for (int it = 0; it < kCap && code == 0; ++it) {
      x_own = fma(alpha, p_own, x_own);
      r_own = fma(-alpha, q_own, r_own);
#pragma unroll
      for (int q = 0; q < 4; ++q) rF[q] = fma(-alpha, qF[q], rF[q]);
      w_own = apply_w(rF, r_own);
      exchange(w_own, wF);
      double gamma_new;
      two_dots(rF, wF, gamma_new, delta);
      if (!(gamma_new > stop)) { code = (gamma_new <= stop) ? 1 : 2; break; }
      const double beta = gamma_new * fast_rcp(gamma);
      const double pwp = delta - beta * gamma_new * fast_rcp(alpha);      // p.Wp of the new direction
      pp = fma(beta * beta, pp, gamma_new);
      if (!(pwp > kCgMinRayleigh * pp)) { code = 2; break; }              // (also NaN)
      alpha = gamma_new * fast_rcp(pwp);
      p_own = fma(beta, p_own, r_own);
      q_own = fma(beta, q_own, w_own);
#pragma unroll
      for (int q = 0; q < 4; ++q) qF[q] = fma(beta, qF[q], wF[q]);
      gamma = gamma_new;
    }
    if (code == 0) code = 2;                                    // the cap
    barrier_with(code);                                         // the waiting waves learn the verdict here
    x0 = x_own;
    return code == 1;
  }
}

// ---- the lane's role in the row-split ADMM layout: wave A4 (0 .. 3) of the four that hold rows of rho G ------------
template <class SH, int A4>
struct RsLane {
  using RS = typename SH::RS;
  int half, slot, pos, var, vi, cidx, ccomp;
  bool isvar, is_cone;
  __device__ __forceinline__ explicit RsLane(const int lane) {
    half = lane >> 5; slot = lane & 31; pos = 32 * A4 + slot;
    var = RS::slot_variable(A4, slot);
    isvar = var >= 0;
    vi = isvar ? var : 0;
    is_cone = RS::slot_is_cone(A4, slot);
    cidx = is_cone ? 2 * A4 + (slot >> 4) : 0;
    ccomp = is_cone ? (slot & 15) : 0;
  }
};
// device stamps of the phases in front of the ADMM loop (100 MHz ticks; phase_ticks of the kernel arguments)
struct CoopStamps { long long t_begin, ticks, ph_asm, ph_p3, ph_solve, ph_wfac, t4; };

// ---- P4 + P5 in the row-split layout (Shape::kRowSplit, RowSplit): the ADMM iterations (fcc_qp.cpp:74-110), the
// epilogue metrics (:184-186) and the write-back, as seen by ADMM wave A4.  Called by the whole-solve kernel
// (coop_wave, SECOND = false) and by the second launch of the two-launch form (coop_admm_kernel, SECOND = true): the
// same instruction sequence on the same data, hence bit-identical results in both forms.
// wbuf: the exchange area, [2][128] doubles of w by iteration parity + 64 words for the lanes that carry no exit
// word; red: [4][4] doubles for the end-of-solve reductions.  Mrow / xc / x: this lane's entries of rho G, x_c and the
// starting point (x0 of the presolve, or the warm-start state).
// Per iteration ONE LDS exchange (w by position, double-buffered by iteration parity) and ONE barrier; the per-wave
// exit words ride on the same barrier.
template <class SH, int A4, bool SECOND>
__device__ __forceinline__ void rs_admm_epilogue(const KernelArgs &a, const int inst, lds_f64 *const wbuf, lds_f64 *const red,
                                                 const int lane, const double (&Mrow)[SH::kRsLen], const double xc, double x,
                                                 const int flags, const CoopStamps &st) {
  constexpr int n = SH::n, nc = SH::nc, s0 = SH::s;
  const RsLane<SH, A4> rl(lane);
  const int half = rl.half, slot = rl.slot, pos = rl.pos, vi = rl.vi, cidx = rl.cidx, ccomp = rl.ccomp;
  const bool isvar = rl.isvar, is_cone = rl.is_cone;
  const double *mup = a.mu + (size_t)inst * (nc / 3);
  const double *lbp = a.lb + (size_t)inst * n;
  const double *ubp = a.ub + (size_t)inst * n;
  double *xs = a.x + (size_t)inst * n;
  double *mux_s = a.mu_x + (size_t)inst * n;
  double *mulc_s = a.mu_lc + (size_t)inst * nc;
  lds_f64 *const w_store[2] = {wbuf + pos, wbuf + 128 + pos};                                       // this slot's w
  const lds_f64 *const w_fetch[2] = {wbuf + 64 * half + (lane & 15), wbuf + 128 + 64 * half + (lane & 15)};   // + 16 q
  const double vlb = isvar ? lbp[vi] : -HUGE_VAL;
  const double vub = isvar ? ubp[vi] : HUGE_VAL;
  const double vmu = is_cone ? mup[cidx] : 0.0;
  double mux = 0.0, mulc = 0.0;
  if (a.warm_start) {
    mux = isvar ? mux_s[vi] : 0.0;
    mulc = is_cone ? mulc_s[is_cone ? vi - s0 : 0] : 0.0;
  }
  double xbar = x, lcbar = x;                        // :74-75
  double xr = 0.0, lr = 0.0;
  const double vkc = cone_lane_constant(vmu, ccomp == 2);
  const double dir_xy = (ccomp == 2) ? 0.0 : 1.0, dir_z = (ccomp == 2) ? 1.0 : 0.0;
  const double eps_b = a.eps_bound, eps_f = a.eps_fcone;
  const double xc_half = half ? 0.0 : xc;            // x_c rides in the lower half's accumulator
  const int max_iter = a.max_iter;
  int n_iter = max_iter;                             // :78
  double Wq[4];
  wg_barrier();                                      // everyone has x (and is done with the factor of W): the exchange area is free
  const unsigned long long cone_lanes = __builtin_amdgcn_ballot_w64(is_cone),
                           flag_lanes = __builtin_amdgcn_ballot_w64(slot == 31),          // where the exit words travel ...
                           flag_cols = __builtin_amdgcn_ballot_w64((lane & 15) == 15);   // ... and arrive (Wq[1], Wq[3])
  // (the exit word goes out by a store of its own BEHIND w's -- every lane stores, the lanes of slot 31 to the slot,
  // the others to a word of their own behind the copies: no EXEC round trip --, so that w does not wait for the
  // compare-and-mask chain of the exit test)
  lds_f64 *const f_store[2] = {(slot == 31) ? w_store[0] : wbuf + 256 + lane, (slot == 31) ? w_store[1] : wbuf + 256 + lane};
  auto publish_w = [&](const int par) {
    // w = x_bar - mu_x, contact segment overwritten by lambda_c_bar - mu_lambda_c (:81-82).  Both lane halves hold
    // the slot's value and store it to the same word.  (A lane without a variable has w == 0.)
    *w_store[par] = select_lanes(cone_lanes, lcbar - mulc, xbar - mux);
  };
  auto publish_flag = [&](const int par, const bool wave_fails) { *f_store[par] = wave_fails ? 1.0 : 0.0; };
  auto fetch_w = [&](const int par) {
#pragma unroll
    for (int q = 0; q < 4; ++q) Wq[q] = w_fetch[par][16 * q];
  };
  if (max_iter > 0) {
    publish_w(0);
    lds_fence();
    publish_flag(0, true);                           // ("not converged": there is no iteration -1 to leave after)
    wg_barrier();
    fetch_w(0);
    lds_fence();
  }
  // lane 15 of every 16-lane row of Wq[1] / Wq[3]: slot 31 of the two waves whose columns this lane half multiplies
  auto everyone_passed = [&]() {
    return ((__builtin_amdgcn_ballot_w64(Wq[1] != 0.0) | __builtin_amdgcn_ballot_w64(Wq[3] != 0.0)) & flag_cols) == 0;
  };
  // One iteration, publishing to / fetching from the copies of parity PAR.  The exit test of the PREVIOUS iteration
  // (:105-109) is evaluated behind this one's mat-vec: its words arrive with w, and looking at them first would put
  // their compare-and-branch chain between the loads and the FMAs that wait for the same loads anyway; an iteration
  // that finds the previous one converged has only computed an accumulator it throws away.  true: leave.
  auto iteration = [&](auto parc) -> bool {
    constexpr int PAR = decltype(parc)::value;
    double acc = xc_half;
    rs_matvec<SH::kRsOdd>(acc, Wq, Mrow);
    if (everyone_passed()) return true;
    x = add_lane_halves(acc);
    const double xm = x + mulc;
    {
      // std::max(std::min(x(i), ub(i)), lb(i)), constraint_utils.cpp:43, as v_min_f64 / v_max_f64 (see the one-wave
      // kernel; written as asm: the builtins re-canonicalise the loop-invariant bounds in every iteration here)
      double tt;
      asm("v_min_f64 %0, %1, %2" : "=v"(tt) : "v"(x + mux), "v"(vub));
      asm("v_max_f64 %0, %1, %2" : "=v"(xbar) : "v"(tt), "v"(vlb));
      xr = x - xbar;                    // :95
      mux += xr;                        // :101
    }
    {
      const double cfx = contact_bcast<0, true>(xm), cfy = contact_bcast<1, true>(xm), cfz = contact_bcast<2, true>(xm);
      lcbar = project_cone_component_lean(cfx, cfy, cfz, vmu, vkc, xm, dir_xy, dir_z);
      lr = x - lcbar;                   // :96
      mulc += lr;                       // :102
    }
    publish_w(PAR);
    lds_fence();
    // exit test :105-109, this wave's part (a lane without a variable has x_res == 0 and passes)
    const bool wave_fails = (__builtin_amdgcn_ballot_w64(!(fabs(xr) < eps_b)) |
                             (cone_lanes & __builtin_amdgcn_ballot_w64(!(fabs(lr) < eps_f)))) != 0;
    publish_flag(PAR, wave_fails);
    wg_barrier();
    fetch_w(PAR);
    lds_fence();
    return false;
  };
  // two iterations per trip: the parity of the exchange copies is a compile-time constant in each (a wave may be a
  // whole iteration ahead of the slowest reader of the previous copies, never two)
  {
    int iter = 0;
    bool left = false;
#pragma nounroll
    while (iter < max_iter) {
      if (iteration(int_c<1>{})) { left = true; break; }
      if (++iter >= max_iter) break;
      if (iteration(int_c<0>{})) { left = true; break; }
      ++iter;
    }
    // (`iter` iterations ran; the last one's words are in Wq either way)
    if (max_iter > 0 && (left || everyone_passed())) n_iter = iter - 1;
  }
  const long long ph_p4 = wall_clock64() - st.t4;

  // ---- P5: epilogue (fcc_qp.cpp:184-186) and write-back; the lower lane half of every ADMM wave reports ---------
  const bool mine = isvar && half == 0;
  const double ffy = contact_bcast<1, true>(x), ffz = contact_bcast<2, true>(x);
  double bv = 0.0, cv = 0.0;
  if (mine) {
    const double tt = min2(x, vub);
    const double pj = max2(tt, vlb);
    const double dd = x - pj;
    bv = dd * dd;                                        // constraint_utils.cpp:61-65
  }
  if (is_cone && ccomp == 0 && half == 0) {
    const double v = sqrt(x * x + ffy * ffy) - vmu * ffz;
    cv = (0.0 < v) ? v : 0.0;                            // constraint_utils.cpp:48-59
  }
  bv = wave_sum(bv);
  cv = wave_sum(cv);
  double xrn = 0.0, lrn = 0.0;
  if (max_iter > 0) {
    xrn = wave_max(isvar ? fabs(xr) : 0.0);              // :97-98
    lrn = wave_max(is_cone ? fabs(lr) : 0.0);
  }
  if (mine) {
    xs[vi] = x;
    mux_s[vi] = mux;                                     // duals are part of the warm-start state (fcc_qp.hpp:149-150)
  }
  if (is_cone && half == 0) mulc_s[vi - s0] = mulc;
  if (lane == 0) {
    red[4 * (A4) + 0] = xrn; red[4 * (A4) + 1] = lrn; red[4 * (A4) + 2] = bv; red[4 * (A4) + 3] = cv;
  }
  wg_barrier();
  if (A4 == 0 && lane == 0) {
    double r0 = red[0], r1 = red[1], r2 = red[2], r3 = red[3];
#pragma unroll
    for (int hh = 1; hh < 4; ++hh) {
      r0 = max2(r0, red[4 * hh + 0]); r1 = max2(r1, red[4 * hh + 1]);
      r2 += red[4 * hh + 2]; r3 += red[4 * hh + 3];
    }
    a.n_iter[inst] = n_iter;
    a.flags[inst] = flags;
    a.metrics[0 * (size_t)a.batch + inst] = r0;
    a.metrics[1 * (size_t)a.batch + inst] = r1;
    a.metrics[2 * (size_t)a.batch + inst] = sqrt(r2);
    a.metrics[3 * (size_t)a.batch + inst] = r3;
    if constexpr (SECOND) {
      // (second launch: the first one left its stamps; slot 5, the instance's total, takes this launch on top)
      if (a.phase_ticks) {
        const size_t B = (size_t)a.batch;
        a.phase_ticks[2 * B + inst] = st.ph_solve;
        a.phase_ticks[4 * B + inst] = ph_p4;
        a.phase_ticks[5 * B + inst] += wall_clock64() - st.t_begin;
      }
    } else {
      a.factor_ticks[inst] = st.ticks;
      if (a.phase_ticks) {
        const size_t B = (size_t)a.batch;
        a.phase_ticks[0 * B + inst] = st.ph_asm;
        a.phase_ticks[1 * B + inst] = st.ticks;
        a.phase_ticks[2 * B + inst] = st.ph_solve;
        a.phase_ticks[3 * B + inst] = st.ph_p3;
        a.phase_ticks[4 * B + inst] = ph_p4;
        a.phase_ticks[5 * B + inst] = wall_clock64() - st.t_begin;
        a.phase_ticks[6 * B + inst] = st.ph_wfac;     // (of slot 1: the factorisation of W alone)
        a.phase_ticks[7 * B + inst] = st.t_begin;
      }
    }
  }
}

// ---- the hand-over between the launches of the two-launch form (row-split shapes) --------------------------------
// Per instance, in the kernel arguments' workspace (stride a.workspace_stride >= kDoubles): what the four ADMM waves
// hold in registers behind the Gram product -- entry e of lane l of wave w of rho G's rows at [(w kRsLen + e) 64 + l],
// x_c at [(4 kRsLen + w) 64 + l], the diagonal entry of rho G of the lane's variable at [(4 kRsLen + 4 + w) 64 + l]:
// every wave instruction stores / loads 512 contiguous bytes.  Flags and stamps travel in their output arrays.
template <class SH>
struct SplitWorkspace {
  static constexpr int kRowsOff = 0, kXcOff = 4 * SH::kRsLen * 64, kGdOff = kXcOff + 4 * 64, kDoubles = kGdOff + 4 * 64;
};
// internal flag bit between the launches: conjugate gradients declined this instance in the second launch; the third
// launch (the whole-solve kernel behind a gate) solves it again from the start and leaves the public flags only
constexpr int kFlagRedo = 1 << 30;

// ---- the whole solve as seen by wave WV of the workgroup ------------------------------------------------------
// SPLIT = true (row-split shapes only): first launch of the two-launch form -- the wave stops behind the Gram product
// and leaves its rows of rho G, x_c and the diagonal of rho G in the workspace (SplitWorkspace) for coop_admm_kernel,
// which runs the presolve by conjugate gradients, the iterations and the epilogue.
template <class SH, int WV, class TIN, bool SPLIT = false>
__device__ __forceinline__ void coop_wave(const KernelArgs &a, const int inst, lds_f64 *L) {
  static_assert(!SPLIT || SH::kRowSplit, "the two-launch form exists for the row-split ADMM layout (num_vars > 64)");
  constexpr int n = SH::n, m = SH::m, nc = SH::nc, s0 = SH::s, N = SH::N, NRB = SH::NRB, NW = SH::NW, RH = SH::RH, CS = SH::CS;
  const int lane = threadIdx.x & 63;
  lds_i32 *const iflags = (lds_i32 *)(L + SH::kInts);
  lds_f64 *const rhs = L + SH::kRhs;

  const TIN *__restrict__ Q = reinterpret_cast<const TIN *>(a.Q) + (size_t)inst * n * n;
  const TIN *__restrict__ A = reinterpret_cast<const TIN *>(a.A) + (size_t)inst * m * n;
  const double *__restrict__ bq = a.b + (size_t)inst * n;
  const double *__restrict__ beq = a.b_eq + (size_t)inst * m;
  const double *mup = a.mu + (size_t)inst * (nc / 3);
  const double *lbp = a.lb + (size_t)inst * n;
  const double *ubp = a.ub + (size_t)inst * n;
  double *xs = a.x + (size_t)inst * n;
  double *mux_s = a.mu_x + (size_t)inst * n;
  double *mulc_s = a.mu_lc + (size_t)inst * nc;

  // ADMM role of this wave.  n <= 64: row half h (always 0), column split c.  n > 64 (Shape::kRowSplit): waves 0 .. 3
  // own 32 slots of variables each (lane & 31; both lane halves hold the slot's variable), the lane half picks the
  // column half of rho G.
  constexpr bool kRS = SH::kRowSplit;
  using RS = typename SH::RS;
  constexpr int h = kRS ? 0 : WV % RH, c = kRS ? 0 : WV / RH;
  const int half = lane >> 5, slot = lane & 31;
  const int pos = kRS ? 32 * (WV & 3) + slot : 64 * h + lane;
  const int var = kRS ? RS::slot_variable(WV & 3, slot) : SH::pos_variable(pos);
  const bool isvar = var >= 0;
  const int vi = isvar ? var : 0;
  const bool is_cone = kRS ? RS::slot_is_cone(WV & 3, slot) : (((pos >> 2) < SH::kContacts) && ((pos & 3) < 3));
  const int cidx = is_cone ? (kRS ? 2 * (WV & 3) + (slot >> 4) : (pos >> 2)) : 0;
  const int ccomp = is_cone ? (kRS ? (slot & 15) : (pos & 3)) : 0;

  const long long t_begin = wall_clock64();
  long long ticks = 0, ph_asm = 0, ph_p3 = 0, ph_solve = 0, ph_wfac = 0;
  [[maybe_unused]] long long ph_p4 = 0;
  int flags = 0;

  // [-b; b_eq] (fcc_qp.cpp:149-150) into LDS, by wave 0; flag words cleared
  if constexpr (WV == 0) {
#pragma unroll
    for (int t = 0; t * 64 < 16 * NRB; ++t) {
      const int i = lane + 64 * t;
      if (i < 16 * NRB) {
        double v = 0.0;
        if (i < n) v = -bq[i];
        else if (i < N) v = beq[i - n];
        rhs[i] = v;
      }
    }
    if (lane < 16) iflags[lane] = 0;
  }

  constexpr bool kAdmmWave = kRS ? (WV < 4) : (c < SH::CSA);            // (the others: q0 / q1 are placeholders, nothing is read)
  constexpr int q0 = (kAdmmWave && !kRS) ? SH::qlo(c) : 0, q1 = (kAdmmWave && !kRS) ? SH::qlo(c + 1) : 1;
  constexpr int col0 = 16 * q0, ncol = kRS ? SH::kRsLen : ((16 * q1 < n) ? 16 * q1 : n) - col0, NQL = kRS ? 4 : q1 - q0;
  double Mrow[ncol];
  double xc = 0.0, x = 0.0, gdiag = 0.0;
  long long t4;
  {
    d4 Wt[NW][NW];
    {
      d4 C[NRB][NRB], X[NRB][SH::NXB];
      load_own_tiles<SH, WV, TIN>(Q, A, a.rho, lane, C);
      const long long t0 = wall_clock64();
      ph_asm = t0 - t_begin;
      wg_barrier();                                   // rhs and the cleared flags are visible
      d4 Gt[NW][SH::NXB];
      coop_factor_forward<SH, WV>(L, lane, C, X, Gt);
      const long long t3 = wall_clock64();
      ticks = t3 - t0;
      coop_gram_output<SH, WV>(L, lane, a.rho, Gt, Wt);
      t4 = wall_clock64();
      ph_p3 = t4 - t3;
    }
    if (iflags[0]) flags |= kFlagAdmmNumerical;       // info() is not checked by the reference (cpp:68)
    // this lane's row of rho G (columns of the own split) and its entry of x_c
    {
      const lds_f64 *img = L + SH::kBig;
      xc = isvar ? rhs[vi] : 0.0;
      if constexpr (kRS) gdiag = isvar ? img[vi * n + vi] : 0.0;
      if constexpr (kAdmmWave && kRS && !SPLIT) {
        // entry e of block k: the variable in slot 16 (k & 1) + e of wave 2 half + (k >> 1) -- its column of this row
        constexpr int LO = SH::kRsOdd;
        static_for<0, SH::kRsLen>([&](auto ec) {
          constexpr int e = decltype(ec)::value;
          constexpr int k = (e < 16) ? 0 : ((e < 16 + LO) ? 1 : ((e < 32 + LO) ? 2 : 3));
          constexpr int i = e - ((k == 0) ? 0 : ((k == 1) ? 16 : ((k == 2) ? 16 + LO : 32 + LO)));
          constexpr int p0 = 32 * (k >> 1) + 16 * (k & 1) + i;                  // position in the lower half's columns
          constexpr int v0 = RS::pos_variable(p0), v1 = RS::pos_variable(64 + p0);
          const int cv = half ? v1 : v0;
          const double g = img[vi * n + ((cv >= 0) ? cv : 0)];
          Mrow[e] = (isvar && cv >= 0) ? g : 0.0;
        });
      } else if constexpr (kAdmmWave) {
#pragma unroll
        for (int j = 0; j < ncol; ++j) Mrow[j] = isvar ? img[vi * n + col0 + j] : 0.0;
      }
    }
    if constexpr (SPLIT) {
      // ---- hand-over to coop_admm_kernel (SplitWorkspace) -------------------------------------------------------------
      // (every wave of the workgroup takes part: with eight waves, wave w + 4 has the lane roles of ADMM wave w and
      // gathers the second half of its entries out of the image -- the four waves that would otherwise only wait)
      {
        using WS = SplitWorkspace<SH>;
        double *ws = a.workspace + (size_t)inst * (size_t)a.workspace_stride;
        const lds_f64 *img = L + SH::kBig;
        constexpr int kParts = SH::W / 4, part = WV / 4, LO = SH::kRsOdd;
        constexpr int e_lo = (SH::kRsLen * part) / kParts, e_hi = (SH::kRsLen * (part + 1)) / kParts;
        static_for<e_lo, e_hi>([&](auto ec) {
          constexpr int e = decltype(ec)::value;
          constexpr int k = (e < 16) ? 0 : ((e < 16 + LO) ? 1 : ((e < 32 + LO) ? 2 : 3));
          constexpr int i = e - ((k == 0) ? 0 : ((k == 1) ? 16 : ((k == 2) ? 16 + LO : 32 + LO)));
          constexpr int p0 = 32 * (k >> 1) + 16 * (k & 1) + i;                  // position in the lower half's columns
          constexpr int v0 = RS::pos_variable(p0), v1 = RS::pos_variable(64 + p0);
          const int cv = half ? v1 : v0;
          const double g = img[vi * n + ((cv >= 0) ? cv : 0)];
          ws[WS::kRowsOff + ((WV & 3) * SH::kRsLen + e) * 64 + lane] = (isvar && cv >= 0) ? g : 0.0;
        });
        if constexpr (part == 0) {
          ws[WS::kXcOff + (WV & 3) * 64 + lane] = xc;
          ws[WS::kGdOff + (WV & 3) * 64 + lane] = gdiag;
        }
      }
      if (WV == 0 && lane == 0) {
        a.flags[inst] = flags;
        a.factor_ticks[inst] = ticks;
        if (a.phase_ticks) {
          const size_t B = (size_t)a.batch;
          a.phase_ticks[0 * B + inst] = ph_asm;
          a.phase_ticks[1 * B + inst] = ticks;
          a.phase_ticks[2 * B + inst] = 0;
          a.phase_ticks[3 * B + inst] = ph_p3;
          a.phase_ticks[4 * B + inst] = 0;
          a.phase_ticks[5 * B + inst] = wall_clock64() - t_begin;
          a.phase_ticks[6 * B + inst] = 0;
          a.phase_ticks[7 * B + inst] = t_begin;
        }
      }
      return;
    }
    if (a.warm_start) x = isvar ? xs[vi] : 0.0;       // fcc_qp.cpp:136-139
    wg_barrier();                                     // the image is dead: the factor of W goes over it

    // ---- presolve (fcc_qp.cpp:159-178; skipped on warm start): x0 = (I - rho G)^-1 x_c, see the one-wave kernel ----
    bool presolved = false;
    if constexpr (kRS) {
      // conjugate gradients first (rs_cg_presolve above); the factorisation of W below is the fallback, with its own
      // singularity rule and the COD hand-off
      if (!a.warm_start) {
        flags |= kFlagPresolveRan;
        if constexpr (kAdmmWave) {
          presolved = rs_cg_presolve<SH, WV>(L + SH::kBig + SH::kRsW, (lds_i32 *)(L + SH::kInts) + 8, lane, isvar, Mrow, xc, gdiag, x);
        } else {
          double unused_x = 0.0;
          const double no_rows[SH::kRsLen] = {};
          presolved = rs_cg_presolve<SH, WV>(L + SH::kBig + SH::kRsW, (lds_i32 *)(L + SH::kInts) + 8, lane, false, no_rows, 0.0, 0.0, unused_x);
        }
        if (presolved) {
          flags |= kFlagPresolveCg;
          const long long t5 = wall_clock64();
          ph_solve = t5 - t4;
          t4 = t5;
        }
      }
    }
    if (!a.warm_start && !presolved) {
      flags |= kFlagPresolveRan;
      const long long tw = wall_clock64();
      coop_factor_store<SH, WV>(L, lane, kPresolveMinPivot, Wt);
      ph_wfac = wall_clock64() - tw;
      ticks += ph_wfac;
      const bool wbad = iflags[1] != 0;
      if (wbad) {
        flags |= kFlagPresolveNumerical;
        if (a.fb_list) {
          // hand the instance to the fallback pass (COD + generic kernel in list mode) and stop here
          if (WV == 0 && lane == 0) {
            a.fb_list[atomicAdd(a.fb_count, 1)] = inst;
            a.flags[inst] = flags;
          }
          return;
        }
      }
      if constexpr (WV == 0) {
        using SW = typename SH::Reduced;
        double r0[RH];
#pragma unroll
        for (int t = 0; t < RH; ++t) r0[t] = (lane + 64 * t < n) ? rhs[(lane + 64 * t < n) ? lane + 64 * t : 0] : 0.0;
        solve_lds_slots<SW, RH>(L + SH::kBig, lane, r0);
        lds_fence();
        // x0 by variable index, over x_c (which every wave holds in registers by now)
#pragma unroll
        for (int t = 0; t < RH; ++t) rhs[lane + 64 * t] = r0[t];
      }
      wg_barrier();
      x = isvar ? rhs[vi] : 0.0;
      const long long t5 = wall_clock64();
      ph_solve = t5 - t4;
      t4 = t5;
    }
  }

  // ---- P4: ADMM iterations (fcc_qp.cpp:74-110) ---------------------------------------------------------------------
  // (a wave without a column split is done: s_barrier counts the surviving waves of the workgroup only)
  if constexpr (!kAdmmWave) return;
  if constexpr (kRS) {
    // ---- row-split layout: P4 + P5 (rs_admm_epilogue above) -------------------------------------------------------------
    const CoopStamps st{t_begin, ticks, ph_asm, ph_p3, ph_solve, ph_wfac, t4};
    rs_admm_epilogue<SH, (WV & 3), false>(a, inst, L + SH::kBig + SH::kRsW, L + SH::kRed2, lane, Mrow, xc, x, flags, st);
    return;
  } else {
  // ---- n <= 64: the columns of rho G split over CSA waves, partial sums through LDS ------------------------------
  static_assert(RH == 1, "column-split layout: one row half (n > 64 takes the row-split layout above)");
  constexpr int CSA = SH::CSA;
  lds_f64 *const part = L + SH::kBig + SH::kPart;
  lds_f64 *const wv = L + SH::kBig + SH::kWvOff + ((RH == 1) ? WV : 0) * SH::kWv;
  lds_i32 *const exitw = iflags + 4;                 // [2][RH]
  const double vlb = isvar ? lbp[vi] : -HUGE_VAL;
  const double vub = isvar ? ubp[vi] : HUGE_VAL;
  const double vmu = is_cone ? mup[cidx] : 0.0;
  double mux = 0.0, mulc = 0.0;
  if (a.warm_start) {
    mux = isvar ? mux_s[vi] : 0.0;
    mulc = is_cone ? mulc_s[is_cone ? vi - s0 : 0] : 0.0;
  }
  double xbar = x, lcbar = x;                        // :74-75
  double xr = 0.0, lr = 0.0;
  const double vkc = cone_lane_constant(vmu, ccomp == 2);
  const double eps_b = a.eps_bound, eps_f = a.eps_fcone;
  lds_f64 *const wslot = wv + (isvar ? var : 16 * NW);
  const int max_iter = a.max_iter;
  int n_iter = max_iter;                             // :78
  double Wq[NQL];
  wg_barrier();                                      // everyone has x (and is done with the factor of W): the exchange area is free
  const unsigned long long var_lanes = __builtin_amdgcn_ballot_w64(isvar), cone_lanes = __builtin_amdgcn_ballot_w64(is_cone),
                           z_lanes = __builtin_amdgcn_ballot_w64(ccomp == 2);
  auto publish_w = [&]() {
    // w = x_bar - mu_x, contact segment overwritten by lambda_c_bar - mu_lambda_c (:81-82)
    const double w = select_lanes(cone_lanes, lcbar - mulc, xbar - mux);
    if (RH == 1 || c == 0) *wslot = w;
  };
  auto fetch_w = [&]() {
#pragma unroll
    for (int q = 0; q < NQL; ++q) Wq[q] = wv[16 * (q0 + q) + (lane & 15)];
  };
  if (max_iter > 0) {
    publish_w();
    if constexpr (RH == 1) lds_fence(); else wg_barrier();
    fetch_w();
    lds_fence();
  }
#pragma nounroll
  for (int iter = 0; iter < max_iter; ++iter) {
    double acc[kMvAcc] = {0.0, 0.0, 0.0, 0.0};
    // (Wq comes straight from the LDS reads, an s_waitcnt in between: no VALU write in front of the DPP reads;
    // tests/test_code_objects.py checks the built code for that)
    dpp_matvec<0, ncol, NQL>(acc, Wq, Mrow);
    if constexpr (CSA > 1) {
      // (double-buffered by iteration parity: a wave may write the next partial sums while a slower one still
      // reads these; the barrier of the next iteration separates it from the one after)
      lds_f64 *const pp = part + (iter & 1) * CS * SH::kPos;
      pp[c * SH::kPos + pos] = mv_total(acc);
      wg_barrier();
      x = xc;
#pragma unroll
      for (int cc = 0; cc < CSA; ++cc) x += pp[cc * SH::kPos + pos];
    } else {
      x = xc + mv_total(acc);
    }
    const double xm = x + mulc;
    const double cfx = quad_bcast<0>(xm), cfy = quad_bcast<1>(xm), cfz = quad_bcast<2>(xm);
    {
      const double v = x + mux;
      const double tt = min2(v, vub);   // constraint_utils.cpp:43
      xbar = max2(tt, vlb);
      xr = x - xbar;                    // :95
      mux += xr;                        // :101
    }
    {
      lcbar = project_cone_component(cfx, cfy, cfz, vmu, vkc, xm, z_lanes);
      lr = x - lcbar;                   // :96
      mulc += lr;                       // :102
    }
    publish_w();
    // exit test :105-109 (see the one-wave kernel): one ballot per row half
    const bool half_fails = ((var_lanes & __builtin_amdgcn_ballot_w64(!(fabs(xr) < eps_b))) |
                             (cone_lanes & __builtin_amdgcn_ballot_w64(!(fabs(lr) < eps_f)))) != 0;
    bool any_fails = half_fails;
    if constexpr (RH == 1) {
      lds_fence();                                   // (w went to this wave's own copy)
    } else {
      if (c == 0 && lane == 0) exitw[(iter & 1) * RH + h] = half_fails ? 1 : 0;
      wg_barrier();
      any_fails = false;
#pragma unroll
      for (int hh = 0; hh < RH; ++hh) any_fails = any_fails || (exitw[(iter & 1) * RH + hh] != 0);
    }
    fetch_w();
    lds_fence();
    if (!any_fails) {
      n_iter = iter;
      break;
    }
  }
  ph_p4 = wall_clock64() - t4;

  // ---- P5: epilogue (fcc_qp.cpp:184-186) and write-back; one wave per row half reports ------------------------------
  const double ffy = quad_bcast<1>(x), ffz = quad_bcast<2>(x);
  double bv = 0.0, cv = 0.0;
  if (isvar) {
    const double tt = min2(x, vub);
    const double p = max2(tt, vlb);
    const double dd = x - p;
    bv = dd * dd;                                        // constraint_utils.cpp:61-65
  }
  if (is_cone && ccomp == 0) {
    const double v = sqrt(x * x + ffy * ffy) - vmu * ffz;
    cv = (0.0 < v) ? v : 0.0;                            // constraint_utils.cpp:48-59
  }
  bv = wave_sum(bv);
  cv = wave_sum(cv);
  double xrn = 0.0, lrn = 0.0;
  if (max_iter > 0) {
    xrn = wave_max(isvar ? fabs(xr) : 0.0);              // :97-98
    lrn = wave_max(is_cone ? fabs(lr) : 0.0);
  }
  lds_f64 *const red = L + SH::kRed;
  if (c == 0) {
    if (isvar) {
      xs[vi] = x;
      mux_s[vi] = mux;                                   // duals are part of the warm-start state (fcc_qp.hpp:149-150)
    }
    if (is_cone) mulc_s[vi - s0] = mulc;
    if (lane == 0) {
      red[4 * h + 0] = xrn; red[4 * h + 1] = lrn; red[4 * h + 2] = bv; red[4 * h + 3] = cv;
    }
  }
  wg_barrier();
  if (WV == 0 && lane == 0) {
    double r0 = red[0], r1 = red[1], r2 = red[2], r3 = red[3];
#pragma unroll
    for (int hh = 1; hh < RH; ++hh) {
      r0 = max2(r0, red[4 * hh + 0]); r1 = max2(r1, red[4 * hh + 1]);
      r2 += red[4 * hh + 2]; r3 += red[4 * hh + 3];
    }
    a.n_iter[inst] = n_iter;
    a.flags[inst] = flags;
    a.metrics[0 * (size_t)a.batch + inst] = r0;
    a.metrics[1 * (size_t)a.batch + inst] = r1;
    a.metrics[2 * (size_t)a.batch + inst] = sqrt(r2);
    a.metrics[3 * (size_t)a.batch + inst] = r3;
    a.factor_ticks[inst] = ticks;
    if (a.phase_ticks) {
      const size_t B = (size_t)a.batch;
      a.phase_ticks[0 * B + inst] = ph_asm;
      a.phase_ticks[1 * B + inst] = ticks;
      a.phase_ticks[2 * B + inst] = ph_solve;
      a.phase_ticks[3 * B + inst] = ph_p3;
      a.phase_ticks[4 * B + inst] = ph_p4;
      a.phase_ticks[5 * B + inst] = wall_clock64() - t_begin;
      a.phase_ticks[6 * B + inst] = ph_wfac;        // (of slot 1: the factorisation of W alone)
      a.phase_ticks[7 * B + inst] = t_begin;
    }
  }
  }   // (n <= 64: column-split layout)
}

// MODE 0: the whole solve in one launch.  MODE 1: first launch of the two-launch form (up to the Gram product).
// MODE 2: third launch of the two-launch form -- the whole solve again, for the instances the second launch marked
// (kFlagRedo: conjugate gradients declined; the direct factorisation of W, its singularity rule and the COD hand-off
// live in this kernel); every other workgroup leaves at once.
template <class SH, class TIN, int MODE = 0>
__global__ __launch_bounds__(kWave * SH::W) void coop_kernel(const KernelArgs a) {
  __shared__ __attribute__((aligned(16))) double lds_raw[SH::kLdsDoubles];
  lds_f64 *const L = (lds_f64 *)lds_raw;
  const int inst = block_instance(a);
  if (inst < 0) return;
  if constexpr (MODE == 2) {
    if (!(a.flags[inst] & kFlagRedo)) return;
  }
  constexpr bool SPLIT = MODE == 1;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  switch (wave) {
    case 0: coop_wave<SH, 0, TIN, SPLIT>(a, inst, L); break;
    case 1: coop_wave<SH, 1, TIN, SPLIT>(a, inst, L); break;
    case 2: if constexpr (SH::W > 2) coop_wave<SH, 2, TIN, SPLIT>(a, inst, L); break;
    case 3: if constexpr (SH::W > 2) coop_wave<SH, 3, TIN, SPLIT>(a, inst, L); break;
    case 4: if constexpr (SH::W > 4) coop_wave<SH, 4, TIN, SPLIT>(a, inst, L); break;
    case 5: if constexpr (SH::W > 4) coop_wave<SH, 5, TIN, SPLIT>(a, inst, L); break;
    case 6: if constexpr (SH::W > 4) coop_wave<SH, 6, TIN, SPLIT>(a, inst, L); break;
    case 7: if constexpr (SH::W > 4) coop_wave<SH, 7, TIN, SPLIT>(a, inst, L); break;
    default: break;
  }
}

// ---- second launch of the two-launch form: presolve by conjugate gradients, P4 and P5 for one instance per workgroup
// of FOUR waves, from what coop_kernel<SH, TIN, 1> left in the workspace.  The whole-solve kernel holds 8 x 252
// registers and 145 KB of LDS per QP -- one QP per CU -- from the first load to the last store, and runs the conjugate
// gradients and the iterations on one wave per SIMD that spends most of an iteration waiting: for the LDS round trip of
// w, for the barrier, for its own dependent chain (950 cycles for ~110 VALU instructions; SQ counters of that kernel:
// 54 % of the wave time parked at an s_waitcnt, profiles/r05_large_sq_counters.json).  This kernel needs the rows of
// rho G (2 kRsLen registers per lane), ~50 registers of state and 2.7 KB of LDS, so WGS workgroups -- independent QPs --
// share a CU and fill each other's waits.
// An instance whose conjugate gradients decline (rs_cg_presolve: W = I - rho G singular or ill-conditioned) is marked
// kFlagRedo and left alone; coop_kernel<SH, TIN, 2> solves it from the start.
template <class SH, int WGS>
__global__ __launch_bounds__(kWave * 4, WGS) void coop_admm_kernel(const KernelArgs a) {
  static_assert(SH::kRowSplit, "row-split ADMM layout");
  using WS = SplitWorkspace<SH>;
  __shared__ __attribute__((aligned(16))) double lds_raw[256 + 64 + 16 + 2];
  lds_f64 *const L = (lds_f64 *)lds_raw;
  const int inst = block_instance(a);
  if (inst < 0) return;
  const long long t_begin = wall_clock64();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const double *ws = a.workspace + (size_t)inst * (size_t)a.workspace_stride;
  auto run = [&](auto wc) {
    constexpr int A4 = decltype(wc)::value;
    const RsLane<SH, A4> rl(lane);
    int flags = a.flags[inst];
    double Mrow[SH::kRsLen];
#pragma unroll
    for (int e = 0; e < SH::kRsLen; ++e) Mrow[e] = ws[WS::kRowsOff + (A4 * SH::kRsLen + e) * 64 + lane];
    const double xc = ws[WS::kXcOff + A4 * 64 + lane];
    double x = 0.0;
    long long ph_solve = 0, t4 = t_begin;
    if (a.phase_ticks) {
      // (stamp of the hand-over's arrival: slot 6 of the phase stamps, which the first launch leaves at zero)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (A4 == 0 && lane == 0) a.phase_ticks[6 * (size_t)a.batch + inst] = wall_clock64() - t_begin;
    }
    if (a.warm_start) {
      x = rl.isvar ? a.x[(size_t)inst * SH::n + rl.vi] : 0.0;       // fcc_qp.cpp:136-139
    } else {
      // presolve (fcc_qp.cpp:159-178): x0 = (I - rho G)^-1 x_c by conjugate gradients
      const double gdiag = ws[WS::kGdOff + A4 * 64 + lane];
      flags |= kFlagPresolveRan;
      const bool presolved = rs_cg_presolve<SH, A4>(L, (lds_i32 *)(L + 256 + 64 + 16), lane, rl.isvar, Mrow, xc, gdiag, x);
      if (!presolved) {
        if (A4 == 0 && lane == 0) a.flags[inst] = kFlagRedo;
        return;
      }
      flags |= kFlagPresolveCg;
      t4 = wall_clock64();
      ph_solve = t4 - t_begin;
    }
    const CoopStamps st{t_begin, 0, 0, 0, ph_solve, 0, t4};
    rs_admm_epilogue<SH, A4, true>(a, inst, L, L + 256 + 64, lane, Mrow, xc, x, flags, st);
  };
  switch (wave) {
    case 0: run(int_c<0>{}); break;
    case 1: run(int_c<1>{}); break;
    case 2: run(int_c<2>{}); break;
    case 3: run(int_c<3>{}); break;
    default: break;
  }
}

}  // namespace coop
}  // namespace fccqp
