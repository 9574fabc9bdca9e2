/*
 * fccqp_oracle.c -- CPU restatement of the reference FCCQP solver.  TEST INFRASTRUCTURE ONLY.
 * See fccqp_oracle.h for scope, provenance and the "parity unpinned" statement.
 *
 * Every function cites the reference file:line it follows.  Reference root: /root/reference.
 * The Eigen routines (LDLT, CompleteOrthogonalDecomposition) are not in the reference tree
 * (empty submodule pinned to Eigen 3.3.90, CMakeLists.txt:10); their published algorithms are
 * restated here and anchored on the reference's call sites fcc_qp.cpp:68,86,161,164-166,173-177.
 */
#include "fccqp_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_seconds(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ====================================================================================
 * constraint_utils.cpp
 * ==================================================================================== */

/* reference: constraint_utils.cpp:5-25  project_to_friction_cone(Vector3d, mu) */
void fccqp_oracle_project_cone3(const double f[3], double mu, double out[3]) {
  /* :7  f.head<2>().norm() -- Eigen's norm() is sqrt(squaredNorm()), not hypot */
  double norm_fxy = sqrt(f[0] * f[0] + f[1] * f[1]);
  /* :10-12 inside the cone */
  if (mu * f[2] >= norm_fxy) {
    out[0] = f[0]; out[1] = f[1]; out[2] = f[2];
    return;
  }
  /* :15-17 polar cone -> origin */
  if (f[2] < -mu * norm_fxy) {
    out[0] = 0.0; out[1] = 0.0; out[2] = 0.0;
    return;
  }
  /* :20-23 project onto the ray on the cone surface */
  double xy_ratio = mu * f[2] / norm_fxy;
  double r[3] = {xy_ratio * f[0], xy_ratio * f[1], f[2]};
  /* Eigen normalize(): z = squaredNorm(); if (z > 0) *this /= sqrt(z);  (zero vector untouched) */
  double z = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  if (z > 0.0) {
    double nrm = sqrt(z);
    r[0] /= nrm; r[1] /= nrm; r[2] /= nrm;
  }
  double d = r[0] * f[0] + r[1] * f[1] + r[2] * f[2];
  out[0] = d * r[0]; out[1] = d * r[1]; out[2] = d * r[2];
}

/* reference: constraint_utils.cpp:27-35 */
int fccqp_oracle_project_cones(const double *f, int nf, const double *mu, int n_mu, double *out) {
  for (int i = 0; i < nf; ++i) out[i] = 0.0; /* :29 VectorXd::Zero */
  for (int i = 0; i < nf / 3; ++i) {
    if (i >= n_mu) return -1; /* :32 friction_coeffs.at(i) throws std::out_of_range */
    fccqp_oracle_project_cone3(f + 3 * i, mu[i], out + 3 * i);
  }
  return 0;
}

/* reference: constraint_utils.cpp:37-46   out(i) = std::max(std::min(x(i), ub(i)), lb(i)) */
void fccqp_oracle_project_bounds(const double *x, const double *lb, const double *ub, int n,
                                 double *out) {
  for (int i = 0; i < n; ++i) {
    /* std::min(a,b) = (b < a) ? b : a ; std::max(a,b) = (a < b) ? b : a */
    double t = (ub[i] < x[i]) ? ub[i] : x[i];
    out[i] = (t < lb[i]) ? lb[i] : t;
  }
}

/* reference: constraint_utils.cpp:48-59 */
int fccqp_oracle_cone_violation(const double *f, int nf, const double *mu, int n_mu,
                                double *violation) {
  double v = 0.0;
  for (int i = 0; i < nf / 3; ++i) {
    int start = 3 * i;
    double fz = f[start + 2];
    if (i >= n_mu) return -1; /* :55 .at(i) */
    double m = mu[i];
    double nrm = sqrt(f[start] * f[start] + f[start + 1] * f[start + 1]);
    double t = nrm - m * fz;
    v += (0.0 < t) ? t : 0.0; /* std::max(0., t) */
  }
  *violation = v;
  return 0;
}

/* reference: constraint_utils.cpp:61-65   (x - project_to_bounds(x)).norm() */
double fccqp_oracle_bound_violation(const double *x, const double *lb, const double *ub, int n) {
  double ss = 0.0;
  for (int i = 0; i < n; ++i) {
    double t = (ub[i] < x[i]) ? ub[i] : x[i];
    double p = (t < lb[i]) ? lb[i] : t;
    double d = x[i] - p;
    ss += d * d;
  }
  return sqrt(ss);
}

/* reference: constraint_utils.cpp:67-75 */
int fccqp_oracle_validate_bounds(const double *lb, const double *ub, int n) {
  for (int i = 0; i < n; ++i)
    if (lb[i] > ub[i]) return 0;
  return 1;
}

/* ====================================================================================
 * Eigen::LDLT<MatrixXd, Lower>  (Eigen 3.3.x  Cholesky/LDLT.h, ldlt_inplace<Lower>::unblocked)
 * call sites: fcc_qp.cpp:68 (ADMM), :161 (presolve), info() at :164,:173
 * ==================================================================================== */
#define AT(a, N, i, j) ((a)[(size_t)(j) * (size_t)(N) + (size_t)(i)])

static void swapd(double *x, double *y) { double t = *x; *x = *y; *y = t; }

int fccqp_oracle_ldlt_compute(double *a, int N, int *transp, double *temp) {
  if (N <= 1) { /* size <= 1: identity transposition, always "Success" */
    for (int i = 0; i < N; ++i) transp[i] = i;
    return 0;
  }
  int found_zero_pivot = 0;
  int ret = 1;
  for (int k = 0; k < N; ++k) {
    /* pivot = largest |diagonal| of the remaining block.  The algorithm is left-looking,
     * so the diagonal entries k+1.. are NOT yet updated: the search sees stale values. */
    int big = k;
    double bigv = fabs(AT(a, N, k, k));
    for (int i = k + 1; i < N; ++i) {
      double v = fabs(AT(a, N, i, i));
      if (v > bigv) { bigv = v; big = i; } /* maxCoeff keeps the FIRST maximum */
    }
    transp[k] = big;
    if (k != big) {
      /* symmetric row/column swap touching the lower triangle only */
      for (int j = 0; j < k; ++j) swapd(&AT(a, N, k, j), &AT(a, N, big, j));
      for (int i = big + 1; i < N; ++i) swapd(&AT(a, N, i, k), &AT(a, N, i, big));
      swapd(&AT(a, N, k, k), &AT(a, N, big, big));
      for (int i = k + 1; i < big; ++i) {
        double tmp = AT(a, N, i, k);
        AT(a, N, i, k) = AT(a, N, big, i);
        AT(a, N, big, i) = tmp;
      }
    }
    /* partition  A10 = row k, cols <k ; A20 = rows >k, cols <k ; A21 = rows >k, col k */
    int rs = N - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; ++j) temp[j] = AT(a, N, j, j) * AT(a, N, k, j);
      double dot = 0.0;
      for (int j = 0; j < k; ++j) dot += AT(a, N, k, j) * temp[j];
      AT(a, N, k, k) -= dot;
      if (rs > 0) {
        /* A21 -= A20 * temp  (column-oriented gemv, like Eigen's col-major product) */
        for (int j = 0; j < k; ++j) {
          double tj = temp[j];
          const double *col = &AT(a, N, k + 1, j);
          double *dst = &AT(a, N, k + 1, k);
          for (int i = 0; i < rs; ++i) dst[i] -= col[i] * tj;
        }
      }
    }
    double akk = AT(a, N, k, k);
    int pivot_is_valid = fabs(akk) > 0.0;
    if (k == 0 && !pivot_is_valid) {
      /* whole diagonal is zero: succeed only if the matrix is zero */
      for (int j = 0; j < N; ++j) {
        transp[j] = j;
        for (int i = j + 1; i < N; ++i)
          if (AT(a, N, i, j) != 0.0) ret = 0;
      }
      return ret ? 0 : 1;
    }
    if (rs > 0 && pivot_is_valid) {
      for (int i = k + 1; i < N; ++i) AT(a, N, i, k) /= akk;
    } else if (rs > 0) {
      for (int i = k + 1; i < N; ++i)
        if (AT(a, N, i, k) != 0.0) ret = 0;
    }
    if (found_zero_pivot && pivot_is_valid) ret = 0; /* factorisation failed */
    else if (!pivot_is_valid) found_zero_pivot = 1;
  }
  return ret ? 0 : 1;
}

/* Eigen LDLT::_solve_impl:  dst = P b; L^-1; D^+ (|d| > DBL_MIN else 0); L^-T; P^T.
 * call sites: fcc_qp.cpp:86 (per ADMM iteration), :174 (presolve) */
void fccqp_oracle_ldlt_solve(const double *a, int N, const int *transp, double *x) {
  for (int k = 0; k < N; ++k)
    if (transp[k] != k) swapd(&x[k], &x[transp[k]]);
  /* unit-lower forward substitution (column oriented) */
  for (int j = 0; j < N; ++j) {
    double xj = x[j];
    for (int i = j + 1; i < N; ++i) x[i] -= AT(a, N, i, j) * xj;
  }
  const double tol = DBL_MIN; /* std::numeric_limits<double>::min() */
  for (int i = 0; i < N; ++i) {
    double d = AT(a, N, i, i);
    if (fabs(d) > tol) x[i] /= d;
    else x[i] = 0.0;
  }
  /* unit-upper (L^T) backward substitution */
  for (int i = N - 1; i >= 0; --i) {
    double s = x[i];
    for (int j = i + 1; j < N; ++j) s -= AT(a, N, j, i) * x[j];
    x[i] = s;
  }
  for (int k = N - 1; k >= 0; --k)
    if (transp[k] != k) swapd(&x[k], &x[transp[k]]);
}

/* ====================================================================================
 * Eigen::CompleteOrthogonalDecomposition<MatrixXd>  compute + solve   (presolve fallback,
 * call sites fcc_qp.cpp:165,176).  A P = Q [T 0; 0 0] Z ; solve returns the minimum-norm
 * least-squares solution.  Rank rule (ColPivHouseholderQR default threshold):
 * |R_ii| > eps * N * max|R_ii|.
 * ==================================================================================== */

/* Householder vector for x (length len): on return x[0] = beta (the new leading entry),
 * x[1..] = essential part v (v0 = 1 implied), returns tau.  H = I - tau v v^T. */
static double make_householder(double *x, int len) {
  double tail_sq = 0.0;
  for (int i = 1; i < len; ++i) tail_sq += x[i] * x[i];
  double c0 = x[0];
  if (tail_sq <= DBL_MIN) { /* Eigen: tailSqNorm <= tol -> tau = 0 */
    for (int i = 1; i < len; ++i) x[i] = 0.0;
    return 0.0;
  }
  double beta = sqrt(c0 * c0 + tail_sq);
  if (c0 >= 0.0) beta = -beta;
  for (int i = 1; i < len; ++i) x[i] /= (c0 - beta);
  x[0] = beta;
  return (beta - c0) / beta;
}

int fccqp_oracle_cod_solve(double *a, int N, const double *rhs, double *x) {
  if (N == 0) return 0;
  int *perm = (int *)malloc(sizeof(int) * (size_t)N);
  double *tau = (double *)calloc((size_t)N, sizeof(double));
  double *c = (double *)malloc(sizeof(double) * (size_t)N);
  double *w = (double *)malloc(sizeof(double) * (size_t)N);
  for (int j = 0; j < N; ++j) perm[j] = j;
  memcpy(c, rhs, sizeof(double) * (size_t)N);

  /* column-pivoted Householder QR, applying each reflector to c as we go (c := Q^T rhs) */
  double maxpivot = 0.0;
  for (int k = 0; k < N; ++k) {
    int big = k;
    double bigv = -1.0;
    for (int j = k; j < N; ++j) {
      double s = 0.0;
      for (int i = k; i < N; ++i) s += AT(a, N, i, j) * AT(a, N, i, j);
      if (s > bigv) { bigv = s; big = j; }
    }
    if (big != k) {
      for (int i = 0; i < N; ++i) swapd(&AT(a, N, i, k), &AT(a, N, i, big));
      int t = perm[k]; perm[k] = perm[big]; perm[big] = t;
    }
    double *col = &AT(a, N, k, k);
    int len = N - k;
    tau[k] = make_householder(col, len);
    double beta = col[0];
    if (fabs(beta) > maxpivot) maxpivot = fabs(beta);
    if (tau[k] != 0.0) {
      for (int j = k + 1; j < N; ++j) {
        double *cj = &AT(a, N, k, j);
        double s = cj[0];
        for (int i = 1; i < len; ++i) s += col[i] * cj[i];
        s *= tau[k];
        cj[0] -= s;
        for (int i = 1; i < len; ++i) cj[i] -= s * col[i];
      }
      double s = c[k];
      for (int i = 1; i < len; ++i) s += col[i] * c[k + i];
      s *= tau[k];
      c[k] -= s;
      for (int i = 1; i < len; ++i) c[k + i] -= s * col[i];
    }
  }
  double thresh = DBL_EPSILON * (double)N * maxpivot;
  int rank = 0;
  for (int k = 0; k < N; ++k)
    if (fabs(AT(a, N, k, k)) > thresh) ++rank;
  /* with column pivoting |R_kk| is non-increasing up to rounding, so the first `rank`
   * diagonal entries are the accepted ones */

  double *y = (double *)calloc((size_t)N, sizeof(double));
  if (rank == N) {
    for (int i = N - 1; i >= 0; --i) {
      double s = c[i];
      for (int j = i + 1; j < N; ++j) s -= AT(a, N, i, j) * y[j];
      y[i] = s / AT(a, N, i, i);
    }
  } else if (rank > 0) {
    /* R1 = R[0:rank, 0:N] = [T 0] Z.  Take B = R1^T (N x rank) = Zt [Tt; 0] by Householder QR,
     * then R1 = [Tt^T 0] Zt^T and the min-norm solution is y = Zt [Tt^-T c1; 0]. */
    int r = rank;
    double *B = (double *)malloc(sizeof(double) * (size_t)N * (size_t)r);
    double *tz = (double *)calloc((size_t)r, sizeof(double));
#define BT(i, j) (B[(size_t)(j) * (size_t)N + (size_t)(i)])
    for (int j = 0; j < r; ++j)
      for (int i = 0; i < N; ++i) BT(i, j) = (i >= j) ? AT(a, N, j, i) : 0.0;
    for (int k = 0; k < r; ++k) {
      double *col = &BT(k, k);
      int len = N - k;
      tz[k] = make_householder(col, len);
      if (tz[k] != 0.0) {
        for (int j = k + 1; j < r; ++j) {
          double *cj = &BT(k, j);
          double s = cj[0];
          for (int i = 1; i < len; ++i) s += col[i] * cj[i];
          s *= tz[k];
          cj[0] -= s;
          for (int i = 1; i < len; ++i) cj[i] -= s * col[i];
        }
      }
    }
    /* Tt is upper r x r in B; solve Tt^T u = c1 (forward substitution) */
    for (int i = 0; i < N; ++i) w[i] = 0.0;
    for (int i = 0; i < r; ++i) {
      double s = c[i];
      for (int j = 0; j < i; ++j) s -= BT(j, i) * w[j];
      w[i] = s / BT(i, i);
    }
    /* y = Zt [u; 0] = H_0 H_1 ... H_{r-1} [u;0] */
    for (int k = r - 1; k >= 0; --k) {
      if (tz[k] == 0.0) continue;
      const double *col = &BT(k, k);
      int len = N - k;
      double s = w[k];
      for (int i = 1; i < len; ++i) s += col[i] * w[k + i];
      s *= tz[k];
      w[k] -= s;
      for (int i = 1; i < len; ++i) w[k + i] -= s * col[i];
    }
    memcpy(y, w, sizeof(double) * (size_t)N);
#undef BT
    free(B);
    free(tz);
  }
  for (int j = 0; j < N; ++j) x[perm[j]] = y[j];
  free(y); free(w); free(c); free(tau); free(perm);
  return rank;
}

/* ====================================================================================
 * class FCCQP   (fcc_qp.hpp:54-172, fcc_qp.cpp:24-207)
 * ==================================================================================== */
struct fccqp_oracle {
  /* fcc_qp.hpp:126-133 */
  int max_iter; double rho, eps_fcone, eps_bound;
  int n_vars, nc, lambda_c_start, n_eq;
  int warm_start;
  int N;
  /* workspace fcc_qp.hpp:135-152 (column-major N x N, like Eigen::MatrixXd) */
  double *q_rho;
  double *M_kkt, *M_kkt_pre, *b_kkt, *kkt_sol;
  double *x, *x_bar, *lambda_c_bar, *mu_x, *mu_lambda_c;
  int *transp, *transp_pre; double *ldlt_temp;
  double *cod_work;
  /* residuals / info fcc_qp.hpp:158-170 */
  double *x_res, *lambda_c_res;
  double x_res_norm, lambda_c_res_norm, bounds_viol, friction_cone_viol;
  int n_iter; double solve_time, factorization_time;
  int presolve_ran, presolve_info;
  /* trace */
  double *trace; int trace_cap, trace_n;
};

static double *zalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }

/* reference: fcc_qp.cpp:24-55 */
fccqp_oracle *fccqp_oracle_create(int num_vars, int num_equality_constraints, int nc,
                                  int lambda_c_start) {
  /* :29-33 asserts */
  if (num_vars < 0 || num_equality_constraints < 0 || nc < 0 || nc % 3 != 0 ||
      lambda_c_start < 0 || lambda_c_start + nc > num_vars)
    return NULL;
  fccqp_oracle *o = (fccqp_oracle *)calloc(1, sizeof(*o));
  /* fcc_qp.hpp:30-35 defaults */
  o->max_iter = 1000; o->rho = 1e-6; o->eps_fcone = 1e-3; o->eps_bound = 1e-6;
  o->n_vars = num_vars; o->n_eq = num_equality_constraints; o->nc = nc;
  o->lambda_c_start = lambda_c_start;
  int N = num_vars + num_equality_constraints;
  o->N = N;
  size_t NN = (size_t)N * (size_t)N;
  o->q_rho = zalloc((size_t)num_vars);
  o->M_kkt = zalloc(NN); o->M_kkt_pre = zalloc(NN);
  o->b_kkt = zalloc((size_t)N); o->kkt_sol = zalloc((size_t)N);
  o->x = zalloc((size_t)num_vars); o->x_bar = zalloc((size_t)num_vars);
  o->x_res = zalloc((size_t)num_vars); o->mu_x = zalloc((size_t)num_vars);
  o->lambda_c_bar = zalloc((size_t)nc); o->mu_lambda_c = zalloc((size_t)nc);
  o->lambda_c_res = zalloc((size_t)nc);
  o->transp = (int *)calloc((size_t)N + 1, sizeof(int));
  o->transp_pre = (int *)calloc((size_t)N + 1, sizeof(int));
  o->ldlt_temp = zalloc((size_t)N);
  o->cod_work = zalloc(NN);
  return o;
}

void fccqp_oracle_destroy(fccqp_oracle *o) {
  if (!o) return;
  free(o->q_rho); free(o->M_kkt); free(o->M_kkt_pre); free(o->b_kkt); free(o->kkt_sol);
  free(o->x); free(o->x_bar); free(o->x_res); free(o->mu_x); free(o->lambda_c_bar);
  free(o->mu_lambda_c); free(o->lambda_c_res); free(o->transp); free(o->transp_pre);
  free(o->ldlt_temp); free(o->cod_work);
  free(o);
}

/* reference: fcc_qp.hpp:75-91 */
void fccqp_oracle_set_options(fccqp_oracle *o, int max_iter, double rho, double eps_fcone,
                              double eps_bound) {
  o->max_iter = max_iter; o->rho = rho; o->eps_fcone = eps_fcone; o->eps_bound = eps_bound;
}
void fccqp_oracle_get_options(const fccqp_oracle *o, int *max_iter, double *rho,
                              double *eps_fcone, double *eps_bound) {
  *max_iter = o->max_iter; *rho = o->rho; *eps_fcone = o->eps_fcone; *eps_bound = o->eps_bound;
}
void fccqp_oracle_set_warm_start(fccqp_oracle *o, int warm_start) { o->warm_start = warm_start != 0; }

void fccqp_oracle_set_trace(fccqp_oracle *o, double *buf, int max_records) {
  o->trace = buf; o->trace_cap = buf ? max_records : 0; o->trace_n = 0;
}
int fccqp_oracle_trace_count(const fccqp_oracle *o) { return o->trace_n; }

/* Eigen maxCoeff/minCoeff based infinity norm, fcc_qp.cpp:97-98:
 *   abs(max(v.maxCoeff(), -v.minCoeff()));  empty vector defined as 0 (UB in the reference). */
static double inf_norm_ref(const double *v, int len) {
  if (len <= 0) return 0.0;
  double mx = v[0], mn = v[0];
  for (int i = 1; i < len; ++i) {
    if (v[i] > mx) mx = v[i];
    if (v[i] < mn) mn = v[i];
  }
  double t = (mx < -mn) ? -mn : mx; /* std::max(mx, -mn) */
  return fabs(t);
}

/* reference: FCCQP::DoADMM fcc_qp.cpp:57-112 */
static void do_admm(fccqp_oracle *o, const double *b, const double *mu, const double *lb,
                    const double *ub) {
  const int n = o->n_vars, N = o->N, nc = o->nc, s = o->lambda_c_start;
  /* :61-64  M_kkt = M_kkt_pre; top-left += rho * I */
  memcpy(o->M_kkt, o->M_kkt_pre, sizeof(double) * (size_t)N * (size_t)N);
  for (int i = 0; i < n; ++i) AT(o->M_kkt, N, i, i) += o->rho;
  /* :67-71 factorise (info() is never checked here) */
  double t0 = now_seconds();
  (void)fccqp_oracle_ldlt_compute(o->M_kkt, N, o->transp, o->ldlt_temp);
  o->factorization_time += now_seconds() - t0;
  /* :74-75 */
  memcpy(o->x_bar, o->x, sizeof(double) * (size_t)n);
  memcpy(o->lambda_c_bar, o->x + s, sizeof(double) * (size_t)nc);
  /* :78-79 */
  o->n_iter = o->max_iter;
  for (int iter = 0; iter < o->max_iter; ++iter) {
    /* :81-83 */
    for (int i = 0; i < n; ++i) o->q_rho[i] = -o->rho * (o->x_bar[i] - o->mu_x[i]);
    for (int i = 0; i < nc; ++i)
      o->q_rho[s + i] = -o->rho * (o->lambda_c_bar[i] - o->mu_lambda_c[i]);
    for (int i = 0; i < n; ++i) o->b_kkt[i] = -(b[i] + o->q_rho[i]);
    /* :86-87 */
    memcpy(o->kkt_sol, o->b_kkt, sizeof(double) * (size_t)N);
    fccqp_oracle_ldlt_solve(o->M_kkt, N, o->transp, o->kkt_sol);
    memcpy(o->x, o->kkt_sol, sizeof(double) * (size_t)n);
    /* :90 x_bar = project_to_bounds(x + mu_x) */
    for (int i = 0; i < n; ++i) o->x_res[i] = o->x[i] + o->mu_x[i]; /* scratch */
    fccqp_oracle_project_bounds(o->x_res, lb, ub, n, o->x_bar);
    /* :91-92 lambda_c_bar = project_to_friction_cone(x[lc] + mu_lc) */
    for (int i = 0; i < nc; ++i) o->lambda_c_res[i] = o->x[s + i] + o->mu_lambda_c[i];
    {
      double tmp[3];
      for (int i = 0; i < nc / 3; ++i) {
        fccqp_oracle_project_cone3(o->lambda_c_res + 3 * i, mu[i], tmp);
        o->lambda_c_bar[3 * i] = tmp[0];
        o->lambda_c_bar[3 * i + 1] = tmp[1];
        o->lambda_c_bar[3 * i + 2] = tmp[2];
      }
    }
    /* :95-98 */
    for (int i = 0; i < n; ++i) o->x_res[i] = o->x[i] - o->x_bar[i];
    for (int i = 0; i < nc; ++i) o->lambda_c_res[i] = o->x[s + i] - o->lambda_c_bar[i];
    o->x_res_norm = inf_norm_ref(o->x_res, n);
    o->lambda_c_res_norm = inf_norm_ref(o->lambda_c_res, nc);
    /* :101-102 dual update BEFORE the exit test */
    for (int i = 0; i < n; ++i) o->mu_x[i] += o->x_res[i];
    for (int i = 0; i < nc; ++i) o->mu_lambda_c[i] += o->lambda_c_res[i];

    if (o->trace && o->trace_n < o->trace_cap) {
      double *r = o->trace + (size_t)o->trace_n * (size_t)(3 * n + 2 * nc);
      memcpy(r, o->x, sizeof(double) * (size_t)n);
      memcpy(r + n, o->x_bar, sizeof(double) * (size_t)n);
      memcpy(r + 2 * n, o->mu_x, sizeof(double) * (size_t)n);
      memcpy(r + 3 * n, o->lambda_c_bar, sizeof(double) * (size_t)nc);
      memcpy(r + 3 * n + nc, o->mu_lambda_c, sizeof(double) * (size_t)nc);
      o->trace_n++;
    }
    /* :105-109 strict '<' on both, 0-based iteration index on exit */
    if (o->lambda_c_res_norm < o->eps_fcone && o->x_res_norm < o->eps_bound) {
      o->n_iter = iter;
      break;
    }
  }
}

/* reference: FCCQP::Solve fcc_qp.cpp:114-191 */
int fccqp_oracle_solve(fccqp_oracle *o, const double *Q, const double *b, const double *A_eq,
                       const double *b_eq, const double *friction_coeffs, int n_friction,
                       const double *lb, const double *ub) {
  const int n = o->n_vars, m = o->n_eq, N = o->N, nc = o->nc, s = o->lambda_c_start;
  double start = now_seconds();
  if (n_friction < nc / 3) return -1; /* .at() would throw (constraint_utils.cpp:32,55) */

  /* :132-133  isInf() is true for +inf and -inf alike */
  int equality_constrained = (nc == 0);
  if (equality_constrained)
    for (int i = 0; i < n; ++i)
      if (!isinf(lb[i]) || !isinf(ub[i])) { equality_constrained = 0; break; }

  /* :136-139 */
  if (!o->warm_start) {
    memset(o->mu_x, 0, sizeof(double) * (size_t)n);
    memset(o->mu_lambda_c, 0, sizeof(double) * (size_t)nc);
  }
  /* :141-143 */
  memset(o->M_kkt_pre, 0, sizeof(double) * (size_t)N * (size_t)N);
  memset(o->M_kkt, 0, sizeof(double) * (size_t)N * (size_t)N);
  memset(o->b_kkt, 0, sizeof(double) * (size_t)N);
  /* :146-150 assemble [[Q, A^T],[A, 0]] and [-b; b_eq]   (inputs are row-major here) */
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) AT(o->M_kkt_pre, N, i, j) = Q[(size_t)i * n + j];
  for (int r = 0; r < m; ++r)
    for (int j = 0; j < n; ++j) {
      double v = A_eq[(size_t)r * n + j];
      AT(o->M_kkt_pre, N, n + r, j) = v;
      AT(o->M_kkt_pre, N, j, n + r) = v;
    }
  for (int i = 0; i < n; ++i) o->b_kkt[i] = -b[i];
  for (int r = 0; r < m; ++r) o->b_kkt[n + r] = b_eq[r];
  /* :153-156 */
  o->factorization_time = 0; o->n_iter = 0; o->x_res_norm = 0; o->lambda_c_res_norm = 0;
  o->presolve_ran = 0; o->presolve_info = 0;

  /* :159-178 presolve */
  if (equality_constrained || !o->warm_start) {
    o->presolve_ran = 1;
    double t0 = now_seconds();
    /* LDLT::compute copies the matrix; M_kkt_pre itself must survive for DoADMM :62 */
    memcpy(o->M_kkt, o->M_kkt_pre, sizeof(double) * (size_t)N * (size_t)N);
    int info = fccqp_oracle_ldlt_compute(o->M_kkt, N, o->transp_pre, o->ldlt_temp);
    o->presolve_info = info;
    double *sol = o->kkt_sol;
    if (info != 0) {
      /* :164-166, :176 COD reads the full (symmetric) matrix */
      memcpy(o->cod_work, o->M_kkt_pre, sizeof(double) * (size_t)N * (size_t)N);
      o->factorization_time += now_seconds() - t0;
      fccqp_oracle_cod_solve(o->cod_work, N, o->b_kkt, sol);
    } else {
      o->factorization_time += now_seconds() - t0;
      memcpy(sol, o->b_kkt, sizeof(double) * (size_t)N);
      fccqp_oracle_ldlt_solve(o->M_kkt, N, o->transp_pre, sol);
    }
    memcpy(o->x, sol, sizeof(double) * (size_t)n);
  }
  /* :180-182 */
  if (!equality_constrained) do_admm(o, b, friction_coeffs, lb, ub);
  /* :184-186 */
  o->bounds_viol = fccqp_oracle_bound_violation(o->x, lb, ub, n);
  o->friction_cone_viol = 0.0;
  fccqp_oracle_cone_violation(o->x + s, nc, friction_coeffs, n_friction, &o->friction_cone_viol);
  /* :188-190 */
  o->solve_time = now_seconds() - start;
  return 0;
}

/* reference: FCCQP::GetSolution fcc_qp.cpp:194-207 */
void fccqp_oracle_get_solution(const fccqp_oracle *o, double *z, fccqp_oracle_details *d) {
  if (d) {
    d->admm_residual_bounds = o->x_res_norm;
    d->admm_residual_friction_cone = o->lambda_c_res_norm;
    d->bounds_viol = o->bounds_viol;
    d->friction_cone_viol = o->friction_cone_viol;
    d->solve_time = o->solve_time;
    d->factorization_time = o->factorization_time;
    d->n_iter = o->n_iter;
    d->solve_status =
        (o->n_iter == o->max_iter) ? FCCQP_ORACLE_MAX_ITERATIONS : FCCQP_ORACLE_SUCCESS;
    d->presolve_ran = o->presolve_ran;
    d->presolve_ldlt_info = o->presolve_info;
  }
  if (z) memcpy(z, o->x, sizeof(double) * (size_t)o->n_vars);
}

void fccqp_oracle_get_state(const fccqp_oracle *o, double *x, double *mu_x, double *mu_lc) {
  if (x) memcpy(x, o->x, sizeof(double) * (size_t)o->n_vars);
  if (mu_x) memcpy(mu_x, o->mu_x, sizeof(double) * (size_t)o->n_vars);
  if (mu_lc) memcpy(mu_lc, o->mu_lambda_c, sizeof(double) * (size_t)o->nc);
}
void fccqp_oracle_set_state(fccqp_oracle *o, const double *x, const double *mu_x,
                            const double *mu_lc) {
  if (x) memcpy(o->x, x, sizeof(double) * (size_t)o->n_vars);
  if (mu_x) memcpy(o->mu_x, mu_x, sizeof(double) * (size_t)o->n_vars);
  if (mu_lc) memcpy(o->mu_lambda_c, mu_lc, sizeof(double) * (size_t)o->nc);
}

/* ====================================================================================
 * Batch driver (CPU baseline): one solver instance per thread, static partition.
 * Mirrors the loop of fcc_qp_test.py:86-89 with set_warm_start(False) for every instance.
 * ==================================================================================== */
typedef struct {
  int n, m, nc, s, max_iter; double rho, eps_fcone, eps_bound;
  int begin, end;
  const double *Q, *b, *A, *b_eq, *mu, *lb, *ub;
  double *z_out; int *n_iter_out; fccqp_oracle_details *details_out;
} batch_job;

static void *batch_worker(void *arg) {
  batch_job *j = (batch_job *)arg;
  fccqp_oracle *o = fccqp_oracle_create(j->n, j->m, j->nc, j->s);
  if (!o) return NULL;
  fccqp_oracle_set_options(o, j->max_iter, j->rho, j->eps_fcone, j->eps_bound);
  fccqp_oracle_set_warm_start(o, 0);
  const size_t n = (size_t)j->n, m = (size_t)j->m, nf = (size_t)(j->nc / 3);
  for (int i = j->begin; i < j->end; ++i) {
    size_t k = (size_t)i;
    fccqp_oracle_solve(o, j->Q + k * n * n, j->b + k * n, j->A + k * m * n, j->b_eq + k * m,
                       j->mu + k * nf, (int)nf, j->lb + k * n, j->ub + k * n);
    fccqp_oracle_details d;
    fccqp_oracle_get_solution(o, j->z_out ? j->z_out + k * n : NULL, &d);
    if (j->n_iter_out) j->n_iter_out[i] = d.n_iter;
    if (j->details_out) j->details_out[i] = d;
  }
  fccqp_oracle_destroy(o);
  return NULL;
}

double fccqp_oracle_solve_batch(int n, int m, int nc, int s, int max_iter, double rho,
                                double eps_fcone, double eps_bound, int batch,
                                const double *Q, const double *b, const double *A,
                                const double *b_eq, const double *mu, const double *lb,
                                const double *ub, double *z_out, int *n_iter_out,
                                fccqp_oracle_details *details_out, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > batch) nthreads = batch > 0 ? batch : 1;
  batch_job *jobs = (batch_job *)calloc((size_t)nthreads, sizeof(batch_job));
  pthread_t *tids = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  double t0 = now_seconds();
  for (int t = 0; t < nthreads; ++t) {
    batch_job *j = &jobs[t];
    j->n = n; j->m = m; j->nc = nc; j->s = s; j->max_iter = max_iter; j->rho = rho;
    j->eps_fcone = eps_fcone; j->eps_bound = eps_bound;
    j->begin = (int)((long long)batch * t / nthreads);
    j->end = (int)((long long)batch * (t + 1) / nthreads);
    j->Q = Q; j->b = b; j->A = A; j->b_eq = b_eq; j->mu = mu; j->lb = lb; j->ub = ub;
    j->z_out = z_out; j->n_iter_out = n_iter_out; j->details_out = details_out;
    if (nthreads == 1) batch_worker(j);
    else pthread_create(&tids[t], NULL, batch_worker, j);
  }
  if (nthreads > 1)
    for (int t = 0; t < nthreads; ++t) pthread_join(tids[t], NULL);
  double dt = now_seconds() - t0;
  free(jobs); free(tids);
  return dt;
}
