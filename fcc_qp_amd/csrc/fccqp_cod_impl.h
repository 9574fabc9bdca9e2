// fccqp_cod_impl.h -- presolve fallback for a rank-deficient / semi-definite KKT matrix, on the
// device (included by fccqp_generic.hip, whose list-mode kernel calls cod_presolve()).
//
// Reference behaviour replaced (/root/reference/src/fcc_qp.cpp:164-166,173-177): when the presolve
// LDL^T reports a numerical issue, the reference solves K x = [-b; b_eq] with
// Eigen::CompleteOrthogonalDecomposition, i.e. takes the MINIMUM-NORM LEAST-SQUARES solution.
// The main kernels detect the condition (zero / non-finite pivot), append the instance to a device
// list and stop; the fallback launch (generic kernel in list mode, a few persistent one-wave blocks)
// computes this x0 for every listed instance and runs the ADMM phase from it.  Nothing returns to
// the host in between.
//
// Algorithm (the published COD: column-pivoted Householder QR, rank from
// |R_kk| > eps * N * max|R_kk|, a second Householder QR on the transposed leading rows,
// x = P Z^T [T^-1 (Q^T b)_1 ; 0]) -- the same steps as the oracle's restatement
// (oracle/fccqp_oracle.c: fccqp_oracle_cod_solve), with the sums done as wave reductions.
//
// This is a rare path (K singular to working precision): one wavefront per listed instance, written
// for clarity.  The two N x N work matrices live in LDS when they fit (N <= 98: the walking shape,
// 158 KB, one workgroup per CU) -- in a global workspace the QR is bound by uncoalesced L2 traffic
// (one column per lane) and an all-singular batch of 4096 took 224 ms whatever the grid size.
#pragma once
#include <cfloat>

#include "fccqp_device.h"

namespace fccqp {
namespace cod {

constexpr int kWave = 64;

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(v, off, kWave);
    v = (v < o) ? o : v;
  }
  return v;
}
// LDSW: the workspace is in LDS (in-order for one wave: the fence only pins the compiler); otherwise it
// is global memory, through which the lanes exchange data: agent scope (see global_phase_sync in
// fccqp_generic.hip for why the wavefront-scope fence is not relied on there)
template <bool LDSW>
__device__ __forceinline__ void wsync() {
  if constexpr (LDSW) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  else agent_release_acquire();            // waits written out: see fccqp_device.h
  __builtin_amdgcn_wave_barrier();
}

// Householder vector for x[0..len): returns tau, x[0] := beta, x[1..] := essential part.
// (same conventions as make_householder in the oracle)
template <bool LDSW>
__device__ double householder(double *x, int len, int lane) {
  double part = 0.0;
  for (int i = 1 + lane; i < len; i += kWave) part += x[i] * x[i];
  const double tail_sq = wsum(part);
  const double c0 = x[0];
  if (tail_sq <= DBL_MIN) {
    for (int i = 1 + lane; i < len; i += kWave) x[i] = 0.0;
    wsync<LDSW>();
    return 0.0;
  }
  double beta = sqrt(c0 * c0 + tail_sq);
  if (c0 >= 0.0) beta = -beta;
  const double den = c0 - beta;
  for (int i = 1 + lane; i < len; i += kWave) x[i] /= den;
  wsync<LDSW>();
  if (lane == 0) x[0] = beta;
  wsync<LDSW>();
  return (beta - c0) / beta;
}

// y[0..len) -= tau * v (v^T y) with v = [1; vess[1..len)]   (one target vector, wave-parallel)
template <bool LDSW>
__device__ void apply_reflector(const double *vess, double tau, double *y, int len, int lane) {
  double part = (lane == 0) ? y[0] : 0.0;
  for (int i = 1 + lane; i < len; i += kWave) part += vess[i] * y[i];
  const double s = wsum(part) * tau;
  if (lane == 0) y[0] -= s;
  for (int i = 1 + lane; i < len; i += kWave) y[i] -= s * vess[i];
  wsync<LDSW>();
}

// Minimum-norm least-squares x0 of instance `inst` into a.x; W = this block's workspace
// (cod_workspace_doubles(n, m) doubles).
template <class TIN, bool LDSW>
__device__ void cod_presolve(const KernelArgs &a, const int inst, double *W) {
  const int lane = threadIdx.x;
  const int n = a.n, m = a.m, N = n + m;
  double *A = W;                                  // N x N, column-major: A[i + j*N]
  double *B = A + (size_t)N * N;                  // N x N scratch for the second QR
  double *c = B + (size_t)N * N;                  // Q^T rhs
  double *w = c + N;
  double *tau = w + N;
  double *tz = tau + N;
  double *y = tz + N;
  int *perm = reinterpret_cast<int *>(y + N);
  {

    const TIN *Q = reinterpret_cast<const TIN *>(a.Q) + (size_t)inst * n * n;
    const TIN *Ae = reinterpret_cast<const TIN *>(a.A) + (size_t)inst * m * n;
    const double *bq = a.b + (size_t)inst * n;
    const double *beq = a.b_eq + (size_t)inst * m;
    // full symmetric KKT [[Q_sym, A^T],[A, 0]] with Q taken from its lower triangle, like the
    // reference's M_kkt_pre_ as seen by LDLT; COD itself reads the full matrix (fcc_qp.cpp:165)
    for (int e = lane; e < N * N; e += kWave) {
      const int i = e % N, j = e / N;
      double v;
      if (i < n && j < n) v = (double)Q[(size_t)i * n + j];   // the reference hands COD the matrix as assembled
      else if (i >= n && j < n) v = (double)Ae[(size_t)(i - n) * n + j];
      else if (i < n && j >= n) v = (double)Ae[(size_t)(j - n) * n + i];
      else v = 0.0;
      A[e] = v;
    }
    for (int i = lane; i < N; i += kWave) {
      c[i] = (i < n) ? -bq[i] : beq[i - n];
      perm[i] = i;
      tau[i] = 0.0;
    }
    wsync<LDSW>();

    // ---- column-pivoted Householder QR, c := Q^T rhs on the fly --------------------------------
    double maxpivot = 0.0;
    for (int k = 0; k < N; ++k) {
      // pivot: first column with the largest remaining squared norm
      double best = -1.0;
      int besti = k;
      for (int j = k + lane; j < N; j += kWave) {
        double sn = 0.0;
        for (int i = k; i < N; ++i) sn += A[i + (size_t)j * N] * A[i + (size_t)j * N];
        if (sn > best) { best = sn; besti = j; }
      }
      // wave arg-max with "first index wins" on ties
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off, kWave);
        const int oi = __shfl_xor(besti, off, kWave);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
      }
      if (besti != k) {
        for (int i = lane; i < N; i += kWave) {
          const double t = A[i + (size_t)k * N];
          A[i + (size_t)k * N] = A[i + (size_t)besti * N];
          A[i + (size_t)besti * N] = t;
        }
        if (lane == 0) { const int t = perm[k]; perm[k] = perm[besti]; perm[besti] = t; }
        wsync<LDSW>();
      }
      double *col = A + k + (size_t)k * N;
      const int len = N - k;
      const double tk = householder<LDSW>(col, len, lane);
      if (lane == 0) tau[k] = tk;
      const double beta = col[0];
      if (fabs(beta) > maxpivot) maxpivot = fabs(beta);
      if (tk != 0.0) {
        // columns j > k, one column per lane
        for (int j = k + 1 + lane; j < N; j += kWave) {
          double *cj = A + k + (size_t)j * N;
          double s = cj[0];
          for (int i = 1; i < len; ++i) s += col[i] * cj[i];
          s *= tk;
          cj[0] -= s;
          for (int i = 1; i < len; ++i) cj[i] -= s * col[i];
        }
        wsync<LDSW>();
        apply_reflector<LDSW>(col, tk, c + k, len, lane);
      }
      wsync<LDSW>();
    }
    const double thresh = DBL_EPSILON * (double)N * maxpivot;
    int rank = 0;
    for (int k = 0; k < N; ++k)
      if (fabs(A[k + (size_t)k * N]) > thresh) ++rank;   // uniform: every lane counts the same

    for (int i = lane; i < N; i += kWave) y[i] = 0.0;
    wsync<LDSW>();
    if (rank == N) {
      // back substitution R y = c (column oriented)
      for (int i = lane; i < N; i += kWave) w[i] = c[i];
      wsync<LDSW>();
      for (int k = N - 1; k >= 0; --k) {
        const double yk = w[k] / A[k + (size_t)k * N];
        wsync<LDSW>();
        if (lane == 0) y[k] = yk;
        for (int i = lane; i < k; i += kWave) w[i] -= A[i + (size_t)k * N] * yk;
        wsync<LDSW>();
      }
    } else if (rank > 0) {
      const int r = rank;
      // B = R1^T (N x r), B[i + j*N] = R[j][i] for i >= j
      for (int e = lane; e < N * r; e += kWave) {
        const int i = e % N, j = e / N;
        B[i + (size_t)j * N] = (i >= j) ? A[j + (size_t)i * N] : 0.0;
      }
      for (int i = lane; i < N; i += kWave) tz[i] = 0.0;
      wsync<LDSW>();
      for (int k = 0; k < r; ++k) {
        double *col = B + k + (size_t)k * N;
        const int len = N - k;
        const double tk = householder<LDSW>(col, len, lane);
        if (lane == 0) tz[k] = tk;
        if (tk != 0.0) {
          for (int j = k + 1 + lane; j < r; j += kWave) {
            double *cj = B + k + (size_t)j * N;
            double s = cj[0];
            for (int i = 1; i < len; ++i) s += col[i] * cj[i];
            s *= tk;
            cj[0] -= s;
            for (int i = 1; i < len; ++i) cj[i] -= s * col[i];
          }
        }
        wsync<LDSW>();
      }
      // Tt is upper r x r in B; solve Tt^T u = c1 by forward substitution (sequential in i)
      for (int i = lane; i < N; i += kWave) w[i] = 0.0;
      wsync<LDSW>();
      for (int i = 0; i < r; ++i) {
        double part = 0.0;
        for (int j = lane; j < i; j += kWave) part += B[j + (size_t)i * N] * w[j];
        const double s = c[i] - wsum(part);
        wsync<LDSW>();
        if (lane == 0) w[i] = s / B[i + (size_t)i * N];
        wsync<LDSW>();
      }
      // y = Zt [u; 0] = H_0 ... H_{r-1} [u; 0]
      for (int k = r - 1; k >= 0; --k) {
        const double tk = tz[k];
        if (tk != 0.0) apply_reflector<LDSW>(B + k + (size_t)k * N, tk, w + k, N - k, lane);
      }
      for (int i = lane; i < N; i += kWave) y[i] = w[i];
      wsync<LDSW>();
    }
    // x[perm[j]] = y[j]; only the first n entries are the solution x_ (fcc_qp.cpp:176)
    double *xs = a.x + (size_t)inst * n;
    for (int j = lane; j < N; j += kWave) {
      const int p = perm[j];
      if (p < n) xs[p] = y[j];
    }
    wsync<LDSW>();
  }
}

}  // namespace cod

inline long long cod_workspace_doubles(int n, int m) {
  const long long N = (long long)n + m;
  return 2 * N * N + 6 * N + 8;
}

}  // namespace fccqp
