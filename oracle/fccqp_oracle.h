/*
 * fccqp_oracle.h -- CPU restatement of the reference FCCQP solver (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X HIP path.  It restates, in plain C99 with no
 * third-party dependency, the algorithm of
 *     /root/reference/src/fcc_qp.cpp:24-207        (FCCQP ctor / Solve / DoADMM / GetSolution)
 *     /root/reference/src/constraint_utils.cpp:5-75 (projections and violation metrics)
 * plus the published algorithms of the Eigen 3.3.90 routines the reference calls
 * (Eigen is an empty, un-vendored submodule of the reference: .gitmodules:1-3,
 * CMakeLists.txt:8-10), namely LDLT<MatrixXd,Lower>::compute/solve/info and
 * CompleteOrthogonalDecomposition::compute/solve.
 *
 * PARITY UNPINNED: the reference ships no golden vectors, no assertions and its only data
 * file (test_data/id_qp_log_walking.npz) is missing from the tree, and
 * the reference itself cannot be compiled here (no Eigen).  The oracle is therefore pinned
 * only by analytic known answers, an independent numpy restatement and an independent
 * solver (see tests/ and DESIGN.md), not by outputs of the reference.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
 * The product (fcc_qp_amd / libfccqp_hip.so) never links, loads or calls it.
 *
 * Conventions: matrices handed in are row-major (C order, as numpy passes them);
 * Q is n x n, A_eq is m x n.  Only the lower triangle of the assembled KKT matrix is read
 * by the LDLT (Eigen default UpLo=Lower), i.e. Q[i][j] with i >= j and A_eq.
 */
#ifndef FCCQP_ORACLE_H
#define FCCQP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* reference: src/fcc_qp.hpp:14-17 */
enum { FCCQP_ORACLE_SUCCESS = 0, FCCQP_ORACLE_MAX_ITERATIONS = 1 };

/* reference: src/fcc_qp.hpp:19-28 (FCCQPDetails) */
typedef struct {
  int n_iter;
  double admm_residual_bounds;
  double admm_residual_friction_cone;
  double solve_time;
  double factorization_time;
  double bounds_viol;
  double friction_cone_viol;
  int solve_status;
  /* restatement-only diagnostics (not in the reference struct) */
  int presolve_ran;        /* 1 if the presolve branch fcc_qp.cpp:159 executed            */
  int presolve_ldlt_info;  /* 0 = Eigen::Success, 1 = NumericalIssue (COD fallback used)  */
} fccqp_oracle_details;

typedef struct fccqp_oracle fccqp_oracle;

/* reference: FCCQP::FCCQP fcc_qp.cpp:24-55.  Returns NULL on invalid sizes
 * (the reference only asserts: nc%3==0, lambda_c_start+nc<=num_vars, all >= 0). */
fccqp_oracle *fccqp_oracle_create(int num_vars, int num_equality_constraints, int nc,
                                  int lambda_c_start);
void fccqp_oracle_destroy(fccqp_oracle *o);

/* reference: fcc_qp.hpp:30-35 defaults max_iter=1000 rho=1e-6 eps_fcone=1e-3 eps_bound=1e-6;
 * setters fcc_qp.hpp:75-91 */
void fccqp_oracle_set_options(fccqp_oracle *o, int max_iter, double rho, double eps_fcone,
                              double eps_bound);
void fccqp_oracle_get_options(const fccqp_oracle *o, int *max_iter, double *rho,
                              double *eps_fcone, double *eps_bound);
void fccqp_oracle_set_warm_start(fccqp_oracle *o, int warm_start);

/* Optional iterate trace: record (x, x_bar, lambda_c_bar, mu_x, mu_lambda_c) after each of
 * the first `max_records` ADMM iterations of the next Solve calls.  Layout per record:
 * [x(n) | x_bar(n) | mu_x(n) | lambda_c_bar(nc) | mu_lambda_c(nc)].  Pass NULL to disable. */
void fccqp_oracle_set_trace(fccqp_oracle *o, double *buf, int max_records);
int fccqp_oracle_trace_count(const fccqp_oracle *o);

/* reference: FCCQP::Solve fcc_qp.cpp:114-191.  n_friction = number of friction coefficients
 * supplied; returns -1 (and solves nothing) if fewer than nc/3 are given -- the reference
 * throws std::out_of_range from .at() (constraint_utils.cpp:32,55).  Returns 0 otherwise. */
int fccqp_oracle_solve(fccqp_oracle *o, const double *Q, const double *b, const double *A_eq,
                       const double *b_eq, const double *friction_coeffs, int n_friction,
                       const double *lb, const double *ub);

/* reference: FCCQP::GetSolution fcc_qp.cpp:194-207 */
void fccqp_oracle_get_solution(const fccqp_oracle *o, double *z, fccqp_oracle_details *d);

/* Direct access to warm-start state (x, mu_x, mu_lambda_c), used by tests. */
void fccqp_oracle_get_state(const fccqp_oracle *o, double *x, double *mu_x, double *mu_lc);
void fccqp_oracle_set_state(fccqp_oracle *o, const double *x, const double *mu_x,
                            const double *mu_lc);

/* ---- constraint_utils.cpp restated ------------------------------------------------- */
/* :5-25 */
void fccqp_oracle_project_cone3(const double f[3], double mu, double out[3]);
/* :27-35 (returns -1 if n_mu < nf/3, mirroring .at()) */
int fccqp_oracle_project_cones(const double *f, int nf, const double *mu, int n_mu, double *out);
/* :37-46 */
void fccqp_oracle_project_bounds(const double *x, const double *lb, const double *ub, int n,
                                 double *out);
/* :48-59 */
int fccqp_oracle_cone_violation(const double *f, int nf, const double *mu, int n_mu,
                                double *violation);
/* :61-65 */
double fccqp_oracle_bound_violation(const double *x, const double *lb, const double *ub, int n);
/* :67-75 */
int fccqp_oracle_validate_bounds(const double *lb, const double *ub, int n);

/* ---- Eigen routines restated (column-major N x N, in place) ------------------------- */
/* LDLT<MatrixXd,Lower>::compute: a (col-major, lda=N) is overwritten with L (strict lower)
 * and D (diagonal); transp[N] receives the transpositions; returns info (0 Success,
 * 1 NumericalIssue). */
int fccqp_oracle_ldlt_compute(double *a, int N, int *transp, double *temp);
/* LDLT::solve: x := A^-1 x using the factorisation above (pseudo-inverse rule on D). */
void fccqp_oracle_ldlt_solve(const double *a, int N, const int *transp, double *x);
/* CompleteOrthogonalDecomposition(A).solve(rhs): minimum-norm least-squares solution of a
 * square N x N system (a is col-major and destroyed); returns the detected rank. */
int fccqp_oracle_cod_solve(double *a, int N, const double *rhs, double *x);

/* ---- batch driver for the CPU baseline ---------------------------------------------- */
/* Solves `batch` independent cold-start instances laid out instance-major
 * (Q[batch][n][n], A[batch][m][n], vectors [batch][len], mu[batch][nc/3]) with one
 * fccqp_oracle per thread and a static partition of the batch over `nthreads` pthreads.
 * z_out[batch][n]; n_iter_out[batch] may be NULL.  Returns wall seconds. */
double fccqp_oracle_solve_batch(int n, int m, int nc, int s, int max_iter, double rho,
                                double eps_fcone, double eps_bound, int batch,
                                const double *Q, const double *b, const double *A,
                                const double *b_eq, const double *mu, const double *lb,
                                const double *ub, double *z_out, int *n_iter_out,
                                fccqp_oracle_details *details_out, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
