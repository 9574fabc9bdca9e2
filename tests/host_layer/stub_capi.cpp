// Stand-in for libfccqp_hip.so in the sanitizer build of the host layer (tests/test_host_layer_sanitized.py):
// implements the C ABI of include/fccqp.h WITHOUT a GPU and WITHOUT solving anything.  Every entry
// point reads each input element it is given (so AddressSanitizer sees an out-of-range view or a
// bad shard offset in the C++ classes above it) and records what it was called with; "solutions" are
// a checksum pattern of the inputs, which lets the driver verify that FCCQPSharded hands every
// shard the right slice and gathers results into the right place.  Test infrastructure only.
#include "fccqp.h"

#include <cstring>
#include <string>
#include <vector>

struct fccqp_solver {
  int n, m, nc, s, batch, device;
  fccqp_options opt{1000, 1e-6, 1e-3, 1e-6};
  int warm = 0, kernel = 0, solves = 0;
  std::vector<double> z;
  std::vector<double> sum_in;   // per instance: sum of all its finite inputs
};

static double finite_sum(const double *p, size_t count) {
  double acc = 0.0;
  for (size_t i = 0; i < count; ++i) {
    const double v = p[i];
    if (v == v && v - v == 0.0) acc += v;   // skips NaN and +-inf (infinite bounds)
  }
  return acc;
}

extern "C" {
int fccqp_version(void) { return 1; }
const char *fccqp_strerror(int code) { return code ? "stub error" : "ok"; }
int fccqp_device_count(void) { return 4; }      // pretend node with four devices
int fccqp_create(int n, int m, int nc, int s, int batch, int device, fccqp_solver **out) {
  if (!out) return FCCQP_ERR_INVALID_ARG;
  *out = nullptr;
  if (n < 0 || m < 0 || nc < 0 || nc % 3 || s < 0 || s + nc > n || batch < 1) return FCCQP_ERR_SHAPE;
  if (device < 0 || device >= 4) return FCCQP_ERR_NO_DEVICE;
  *out = new fccqp_solver{n, m, nc, s, batch, device};
  (*out)->z.assign((size_t)batch * n, 0.0);
  (*out)->sum_in.assign((size_t)batch, 0.0);
  return FCCQP_OK;
}
int fccqp_destroy(fccqp_solver *s) { delete s; return FCCQP_OK; }
const char *fccqp_last_error(const fccqp_solver *) { return ""; }
int fccqp_default_options(fccqp_options *o) { *o = fccqp_options{1000, 1e-6, 1e-3, 1e-6}; return FCCQP_OK; }
int fccqp_set_options(fccqp_solver *s, const fccqp_options *o) { s->opt = *o; return FCCQP_OK; }
int fccqp_get_options(const fccqp_solver *s, fccqp_options *o) { *o = s->opt; return FCCQP_OK; }
int fccqp_set_warm_start(fccqp_solver *s, int w) { s->warm = w; return FCCQP_OK; }
int fccqp_set_kernel(fccqp_solver *s, int k) { s->kernel = k; return FCCQP_OK; }
int fccqp_set_presolve_fallback(fccqp_solver *, int) { return FCCQP_OK; }
int fccqp_set_kernel_timing(fccqp_solver *, int) { return FCCQP_OK; }
int fccqp_set_self_check(fccqp_solver *, int) { return FCCQP_OK; }
int fccqp_set_dispatch_order(fccqp_solver *, int) { return FCCQP_OK; }
int fccqp_get_kernel(const fccqp_solver *s, int *k) { *k = s->kernel; return FCCQP_OK; }
int fccqp_kernel_for_shape(int, int, int, int) { return FCCQP_KERNEL_GENERIC; }
int fccqp_get_shape(const fccqp_solver *s, int *n, int *m, int *nc, int *st, int *b, int *d) {
  if (n) *n = s->n; if (m) *m = s->m; if (nc) *nc = s->nc; if (st) *st = s->s; if (b) *b = s->batch; if (d) *d = s->device;
  return FCCQP_OK;
}
int fccqp_solve(fccqp_solver *s, const double *Q, const double *b, const double *A, const double *beq,
                const double *mu, int n_friction, const double *lb, const double *ub) {
  const size_t n = s->n, m = s->m, nm = s->nc / 3;
  if (n_friction < (int)nm) return FCCQP_ERR_FRICTION_COEFFS;
  for (int i = 0; i < s->batch; ++i) {
    double acc = finite_sum(Q + (size_t)i * n * n, n * n) + finite_sum(b + (size_t)i * n, n) +
                 finite_sum(lb + (size_t)i * n, n) + finite_sum(ub + (size_t)i * n, n);
    if (m) acc += finite_sum(A + (size_t)i * m * n, m * n) + finite_sum(beq + (size_t)i * m, m);
    if (nm) acc += finite_sum(mu + (size_t)i * n_friction, nm);
    s->sum_in[i] = acc;
    for (size_t j = 0; j < n; ++j) s->z[(size_t)i * n + j] = acc + (double)j + (s->warm ? 0.5 : 0.0);
  }
  s->solves++;
  return FCCQP_OK;
}
int fccqp_solve_async(fccqp_solver *s, const double *Q, const double *b, const double *A, const double *beq,
                      const double *mu, int nf, const double *lb, const double *ub) {
  return fccqp_solve(s, Q, b, A, beq, mu, nf, lb, ub);
}
int fccqp_host_register(void *, size_t) { return FCCQP_OK; }
int fccqp_host_unregister(void *) { return FCCQP_OK; }
// ("device" pointers are host pointers here: the stub reads them like fccqp_solve, nc/3 coefficients per instance)
int fccqp_solve_device(fccqp_solver *s, const double *Q, const double *b, const double *A, const double *beq, const double *mu,
                       const double *lb, const double *ub, void *) { return fccqp_solve(s, Q, b, A, beq, mu, s->nc / 3, lb, ub); }
int fccqp_solve_f32(fccqp_solver *, const float *, const float *, const float *, const float *, const float *, int,
                    const float *, const float *) { return FCCQP_ERR_NO_DEVICE; }
int fccqp_solve_device_f32(fccqp_solver *, const float *, const float *, const float *, const float *, const float *,
                           const float *, const float *, void *) { return FCCQP_ERR_NO_DEVICE; }
int fccqp_synchronize(fccqp_solver *) { return FCCQP_OK; }
int fccqp_get_solution(fccqp_solver *s, double *z, fccqp_details *d) {
  if (!s->solves) return FCCQP_ERR_NOT_SOLVED;
  if (z) std::memcpy(z, s->z.data(), sizeof(double) * s->z.size());
  if (d)
    for (int i = 0; i < s->batch; ++i) {
      d[i] = fccqp_details{};
      d[i].n_iter = s->device * 1000 + i;            // which handle / which local index produced it
      d[i].bounds_viol = s->sum_in[i];
      d[i].solve_status = (d[i].n_iter == s->opt.max_iter) ? FCCQP_STATUS_MAX_ITERATIONS : FCCQP_STATUS_SUCCESS;
    }
  return FCCQP_OK;
}
int fccqp_get_z_async(fccqp_solver *s, double *z) { return fccqp_get_solution(s, z, nullptr); }
int fccqp_get_solution_device(fccqp_solver *, const double **, const int **, const int **, const double **,
                              const long long **) { return FCCQP_ERR_NO_DEVICE; }
int fccqp_get_state(fccqp_solver *, double *, double *, double *) { return FCCQP_OK; }
int fccqp_set_state(fccqp_solver *, const double *, const double *, const double *) { return FCCQP_OK; }
int fccqp_last_kernel_ms(fccqp_solver *s, float *ms) { *ms = 1.0f + s->device; return FCCQP_OK; }
int fccqp_last_timing(fccqp_solver *s, float *ms, int *by) { *ms = 1.0f + s->device; *by = FCCQP_TIMING_DEVICE_EVENTS; return FCCQP_OK; }
int fccqp_phase_seconds(fccqp_solver *, double out[8]) { std::memset(out, 0, 64); return FCCQP_OK; }
int fccqp_algorithmic_bytes(const fccqp_solver *, long long *b) { *b = 0; return FCCQP_OK; }
int fccqp_debug_read_workspace(fccqp_solver *, double *, long long, long long *a) { if (a) *a = 0; return FCCQP_OK; }
}
