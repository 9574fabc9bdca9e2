// fcc_qp.cpp -- fcc_qp::FCCQP / fcc_qp::FCCQPBatch on top of the C ABI (include/fccqp.h).
// Mirrors the control flow of the reference class (src/fcc_qp.cpp:24-207); every numeric step is
// delegated to the HIP library.  Plain C++17, no HIP headers: this file only sees the C ABI.
#include "fcc_qp.hpp"

#include <cmath>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>

#include "fccqp.h"

namespace fcc_qp {
namespace {

[[noreturn]] void raise(int code, fccqp_solver *h, const char *where) {
  std::string msg = std::string(where) + ": " + fccqp_strerror(code);
  if (h && fccqp_last_error(h)[0]) msg += std::string(" [") + fccqp_last_error(h) + "]";
  switch (code) {
    case FCCQP_ERR_SHAPE:
    case FCCQP_ERR_INVALID_ARG:
      throw std::invalid_argument(msg);
    case FCCQP_ERR_FRICTION_COEFFS:
      // the reference throws std::out_of_range from friction_coeffs.at(i)
      // (src/constraint_utils.cpp:32,55)
      throw std::out_of_range(msg);
    default:
      throw FccqpError(code, msg);
  }
}

void check(int code, fccqp_solver *h, const char *where) {
  if (code != FCCQP_OK) raise(code, h, where);
}

fccqp_options to_c(const FCCQPOptions &o) { return fccqp_options{o.max_iter, o.rho, o.eps_fcone, o.eps_bound}; }

FCCQPDetails from_c(const fccqp_details &d) {
  FCCQPDetails o;
  o.n_iter = d.n_iter;
  o.admm_residual_bounds = d.admm_residual_bounds;
  o.admm_residual_friction_cone = d.admm_residual_friction_cone;
  o.solve_time = d.solve_time;
  o.factorization_time = d.factorization_time;
  o.bounds_viol = d.bounds_viol;
  o.friction_cone_viol = d.friction_cone_viol;
  o.solve_status = d.solve_status == FCCQP_STATUS_MAX_ITERATIONS ? kMaxIterations : kSuccess;
  o.flags = d.flags;
  return o;
}

}  // namespace

// ---------------------------------------------------------------------------- FCCQP
FCCQP::FCCQP(int num_vars, int num_equality_constraints, int nc, int lambda_c_start, int device)
    : n_vars_(num_vars), nc_(nc), lambda_c_start_(lambda_c_start), n_eq_(num_equality_constraints) {
  check(fccqp_create(num_vars, num_equality_constraints, nc, lambda_c_start, 1, device, &h_), nullptr,
        "FCCQP::FCCQP");
  const size_t n = (size_t)n_vars_, m = (size_t)n_eq_;
  pack_.resize(n * n + n + m * n + m + (size_t)(nc_ / 3) + n + n);
}

FCCQP::~FCCQP() { fccqp_destroy(h_); }

void FCCQP::set_rho(double rho) {
  options_.rho = rho;
  fccqp_options o = to_c(options_);
  check(fccqp_set_options(h_, &o), h_, "FCCQP::set_rho");
}
void FCCQP::set_max_iter(int n) {
  options_.max_iter = n;
  fccqp_options o = to_c(options_);
  check(fccqp_set_options(h_, &o), h_, "FCCQP::set_max_iter");
}
void FCCQP::set_options(FCCQPOptions opt) {
  options_ = opt;
  fccqp_options o = to_c(options_);
  check(fccqp_set_options(h_, &o), h_, "FCCQP::set_options");
}
void FCCQP::set_warm_start(bool warm_start) {
  check(fccqp_set_warm_start(h_, warm_start ? 1 : 0), h_, "FCCQP::set_warm_start");
}

void FCCQP::Solve(const ConstMatrixView &Q, const ConstVectorView &b, const ConstMatrixView &A_eq,
                  const ConstVectorView &b_eq, const std::vector<double> &friction_coeffs,
                  const ConstVectorView &lb, const ConstVectorView &ub) {
  const long n = n_vars_, m = n_eq_;
  // the reference only asserts these (src/fcc_qp.cpp:122-130); a mismatch is UB in its Release
  // build and a std::invalid_argument here
  if (Q.rows != n || Q.cols != n) throw std::invalid_argument("FCCQP::Solve: Q must be num_vars x num_vars");
  if (b.size != n) throw std::invalid_argument("FCCQP::Solve: b must have num_vars entries");
  if (A_eq.rows != m || (m > 0 && A_eq.cols != n))
    throw std::invalid_argument("FCCQP::Solve: A_eq must be num_equality_constraints x num_vars");
  if (b_eq.size != m) throw std::invalid_argument("FCCQP::Solve: b_eq must have num_equality_constraints entries");
  if (lb.size != n || ub.size != n) throw std::invalid_argument("FCCQP::Solve: lb/ub must have num_vars entries");
  if ((long)friction_coeffs.size() < nc_ / 3)
    throw std::out_of_range("FCCQP::Solve: friction_coeffs has fewer than nc/3 entries");

  // pack into contiguous row-major staging (the reference copies into its own workspace too,
  // src/fcc_qp.cpp:146-150)
  double *pQ = pack_.data();
  double *pb = pQ + n * n;
  double *pA = pb + n;
  double *pbe = pA + m * n;
  double *pmu = pbe + m;
  double *plb = pmu + nc_ / 3;
  double *pub = plb + n;
  for (long i = 0; i < n; ++i)
    for (long j = 0; j < n; ++j) pQ[i * n + j] = Q(i, j);
  for (long i = 0; i < n; ++i) { pb[i] = b(i); plb[i] = lb(i); pub[i] = ub(i); }
  for (long r = 0; r < m; ++r) {
    for (long j = 0; j < n; ++j) pA[r * n + j] = A_eq(r, j);
    pbe[r] = b_eq(r);
  }
  for (long i = 0; i < nc_ / 3; ++i) pmu[i] = friction_coeffs[(size_t)i];
  check(fccqp_solve(h_, pQ, pb, pA, pbe, pmu, nc_ / 3, plb, pub), h_, "FCCQP::Solve");
}

FCCQPSolution FCCQP::GetSolution() const {
  FCCQPSolution out;
  out.z.assign((size_t)n_vars_, 0.0);
  fccqp_details d{};
  int rc = fccqp_get_solution(h_, out.z.data(), &d);
  if (rc == FCCQP_ERR_NOT_SOLVED) {
    // the reference returns its zero-initialised members before any Solve (src/fcc_qp.cpp:48-53)
    out.details = FCCQPDetails{};
    out.details.solve_status = (0 == options_.max_iter) ? kMaxIterations : kSuccess;
    return out;
  }
  check(rc, h_, "FCCQP::GetSolution");
  out.details = from_c(d);
  return out;
}

// ---------------------------------------------------------------------------- FCCQPBatch
FCCQPBatch::FCCQPBatch(int num_vars, int num_equality_constraints, int nc, int lambda_c_start,
                       int batch, int device)
    : n_vars_(num_vars), n_eq_(num_equality_constraints), nc_(nc), lambda_c_start_(lambda_c_start),
      batch_(batch) {
  check(fccqp_create(num_vars, num_equality_constraints, nc, lambda_c_start, batch, device, &h_),
        nullptr, "FCCQPBatch::FCCQPBatch");
}
FCCQPBatch::~FCCQPBatch() { fccqp_destroy(h_); }

void FCCQPBatch::set_rho(double rho) {
  options_.rho = rho;
  fccqp_options o = to_c(options_);
  check(fccqp_set_options(h_, &o), h_, "FCCQPBatch::set_rho");
}
void FCCQPBatch::set_max_iter(int n) {
  options_.max_iter = n;
  fccqp_options o = to_c(options_);
  check(fccqp_set_options(h_, &o), h_, "FCCQPBatch::set_max_iter");
}
void FCCQPBatch::set_options(FCCQPOptions opt) {
  options_ = opt;
  fccqp_options o = to_c(options_);
  check(fccqp_set_options(h_, &o), h_, "FCCQPBatch::set_options");
}
void FCCQPBatch::set_warm_start(bool warm_start) {
  check(fccqp_set_warm_start(h_, warm_start ? 1 : 0), h_, "FCCQPBatch::set_warm_start");
}
void FCCQPBatch::set_kernel(int kernel) { check(fccqp_set_kernel(h_, kernel), h_, "FCCQPBatch::set_kernel"); }
void FCCQPBatch::set_kernel_timing(bool enabled) {
  check(fccqp_set_kernel_timing(h_, enabled ? 1 : 0), h_, "FCCQPBatch::set_kernel_timing");
}
int FCCQPBatch::kernel_in_use() const {
  int k = 0;
  check(fccqp_get_kernel(h_, &k), h_, "FCCQPBatch::kernel_in_use");
  return k;
}

void FCCQPBatch::Solve(const double *Q, const double *b, const double *A_eq, const double *b_eq,
                       const double *friction_coeffs, int n_friction, const double *lb,
                       const double *ub) {
  check(fccqp_solve(h_, Q, b, A_eq, b_eq, friction_coeffs, n_friction, lb, ub), h_, "FCCQPBatch::Solve");
}
void FCCQPBatch::SolveDevice(const double *dQ, const double *db, const double *dA_eq,
                             const double *db_eq, const double *dfriction, const double *dlb,
                             const double *dub, void *stream) {
  check(fccqp_solve_device(h_, dQ, db, dA_eq, db_eq, dfriction, dlb, dub, stream), h_,
        "FCCQPBatch::SolveDevice");
}
void FCCQPBatch::Solve(const float *Q, const float *b, const float *A_eq, const float *b_eq,
                       const float *friction_coeffs, int n_friction, const float *lb,
                       const float *ub) {
  check(fccqp_solve_f32(h_, Q, b, A_eq, b_eq, friction_coeffs, n_friction, lb, ub), h_, "FCCQPBatch::Solve");
}
void FCCQPBatch::SolveDevice(const float *dQ, const float *db, const float *dA_eq,
                             const float *db_eq, const float *dfriction, const float *dlb,
                             const float *dub, void *stream) {
  check(fccqp_solve_device_f32(h_, dQ, db, dA_eq, db_eq, dfriction, dlb, dub, stream), h_,
        "FCCQPBatch::SolveDevice");
}
void FCCQPBatch::Synchronize() { check(fccqp_synchronize(h_), h_, "FCCQPBatch::Synchronize"); }

FCCQPBatchSolution FCCQPBatch::GetSolution() const {
  FCCQPBatchSolution out;
  out.batch = batch_;
  out.num_vars = n_vars_;
  out.z.assign((size_t)batch_ * (size_t)n_vars_, 0.0);
  std::vector<fccqp_details> d((size_t)batch_);
  check(fccqp_get_solution(h_, out.z.data(), d.data()), h_, "FCCQPBatch::GetSolution");
  out.details.reserve((size_t)batch_);
  for (const auto &x : d) out.details.push_back(from_c(x));
  return out;
}
void FCCQPBatch::set_self_check(bool enabled) {
  check(fccqp_set_self_check(h_, enabled ? 1 : 0), h_, "FCCQPBatch::set_self_check");
}
void FCCQPBatch::set_dispatch_order_by_previous_iterations(bool enabled) {
  check(fccqp_set_dispatch_order(h_, enabled ? FCCQP_ORDER_PREVIOUS_ITERATIONS : FCCQP_ORDER_HASH), h_,
        "FCCQPBatch::set_dispatch_order_by_previous_iterations");
}
std::pair<float, int> FCCQPBatch::LastTiming() const {
  float ms = 0.f;
  int by = FCCQP_TIMING_NONE;
  check(fccqp_last_timing(h_, &ms, &by), h_, "FCCQPBatch::LastTiming");
  return {ms, by};
}
float FCCQPBatch::LastKernelMilliseconds() const {
  float ms = 0.f;
  check(fccqp_last_kernel_ms(h_, &ms), h_, "FCCQPBatch::LastKernelMilliseconds");
  return ms;
}
long long FCCQPBatch::AlgorithmicBytes() const {
  long long b = 0;
  check(fccqp_algorithmic_bytes(h_, &b), h_, "FCCQPBatch::AlgorithmicBytes");
  return b;
}

// ---------------------------------------------------------------------------- FCCQPSharded
// One persistent host thread per shard: created with the object, woken for every call that has per-shard work (a
// std::thread per shard and call, as through round 4, costs ~30 us each -- more than a small solve).  run() hands
// fn(g) to thread g for every g and returns when all are done; one run() at a time (the class is as non-reentrant as
// the reference's, src/fcc_qp.hpp:130-171).
class FCCQPSharded::Workers {
 public:
  explicit Workers(std::size_t count) : rc_(count, FCCQP_OK) {
    threads_.reserve(count);
    for (std::size_t g = 0; g < count; ++g) threads_.emplace_back([this, g] { loop(g); });
  }
  ~Workers() {
    {
      std::lock_guard<std::mutex> lock(m_);
      stop_ = true;
      ++generation_;
    }
    wake_.notify_all();
    for (auto &t : threads_) t.join();
  }
  // returns the status codes of fn(0) .. fn(count - 1)
  const std::vector<int> &run(const std::function<int(std::size_t)> &fn) {
    std::unique_lock<std::mutex> lock(m_);
    fn_ = &fn;
    pending_ = threads_.size();
    ++generation_;
    wake_.notify_all();
    done_.wait(lock, [this] { return pending_ == 0; });
    fn_ = nullptr;
    return rc_;
  }

 private:
  void loop(std::size_t g) {
    unsigned long long seen = 0;
    for (;;) {
      const std::function<int(std::size_t)> *fn = nullptr;
      {
        std::unique_lock<std::mutex> lock(m_);
        wake_.wait(lock, [&] { return generation_ != seen; });
        seen = generation_;
        if (stop_) return;
        fn = fn_;
      }
      const int rc = (*fn)(g);
      {
        std::lock_guard<std::mutex> lock(m_);
        rc_[g] = rc;
        if (--pending_ == 0) done_.notify_one();
      }
    }
  }
  std::mutex m_;
  std::condition_variable wake_, done_;
  std::vector<std::thread> threads_;
  std::vector<int> rc_;
  const std::function<int(std::size_t)> *fn_ = nullptr;
  std::size_t pending_ = 0;
  unsigned long long generation_ = 0;
  bool stop_ = false;
};

void FCCQPSharded::on_every_shard(const std::function<int(std::size_t)> &fn, const char *where) const {
  const std::vector<int> &rc = workers_->run(fn);
  for (std::size_t g = 0; g < h_.size(); ++g) check(rc[g], h_[g], where);
}

FCCQPSharded::FCCQPSharded(int num_vars, int num_equality_constraints, int nc, int lambda_c_start,
                           int batch, const std::vector<int> &devices)
    : n_vars_(num_vars), n_eq_(num_equality_constraints), nc_(nc), lambda_c_start_(lambda_c_start),
      batch_(batch), devices_(devices) {
  if (devices.empty()) throw FccqpError(FCCQP_ERR_INVALID_ARG, "FCCQPSharded: empty device list");
  if (batch < (int)devices.size()) throw FccqpError(FCCQP_ERR_SHAPE, "FCCQPSharded: fewer instances than shards");
  h_.assign(devices.size(), nullptr);
  for (size_t g = 0; g < devices.size(); ++g) {
    const int nb = shard_begin((int)g + 1) - shard_begin((int)g);
    int rc = fccqp_create(num_vars, num_equality_constraints, nc, lambda_c_start, nb, devices[g], &h_[g]);
    if (rc != FCCQP_OK) {
      for (auto *h : h_) fccqp_destroy(h);
      check(rc, nullptr, "FCCQPSharded::FCCQPSharded");
    }
  }
  workers_.reset(new Workers(devices.size()));
}
FCCQPSharded::~FCCQPSharded() {
  workers_.reset();                 // (joins the threads before the handles they use go away)
  for (auto *h : h_) fccqp_destroy(h);
}
void FCCQPSharded::set_options(FCCQPOptions opt) {
  options_ = opt;
  fccqp_options o = to_c(options_);
  for (auto *h : h_) check(fccqp_set_options(h, &o), h, "FCCQPSharded::set_options");
}
void FCCQPSharded::set_warm_start(bool warm_start) {
  for (auto *h : h_) check(fccqp_set_warm_start(h, warm_start ? 1 : 0), h, "FCCQPSharded::set_warm_start");
}
void FCCQPSharded::set_kernel(int kernel) {
  for (auto *h : h_) check(fccqp_set_kernel(h, kernel), h, "FCCQPSharded::set_kernel");
}
void FCCQPSharded::set_kernel_timing(bool enabled) {
  for (auto *h : h_) check(fccqp_set_kernel_timing(h, enabled ? 1 : 0), h, "FCCQPSharded::set_kernel_timing");
}

void FCCQPSharded::Solve(const double *Q, const double *b, const double *A_eq, const double *b_eq,
                         const double *friction_coeffs, int n_friction, const double *lb,
                         const double *ub) {
  const size_t n = (size_t)n_vars_, m = (size_t)n_eq_;
  on_every_shard([&](size_t g) {
    const size_t lo = (size_t)shard_begin((int)g);
    return fccqp_solve(h_[g], Q + lo * n * n, b + lo * n, A_eq ? A_eq + lo * m * n : nullptr,
                       b_eq ? b_eq + lo * m : nullptr,
                       friction_coeffs ? friction_coeffs + lo * (size_t)n_friction : nullptr, n_friction,
                       lb + lo * n, ub + lo * n);
  }, "FCCQPSharded::Solve");
}

void FCCQPSharded::SolveDevice(const std::vector<ShardDeviceInputs> &in) {
  if (in.size() != h_.size()) throw std::invalid_argument("FCCQPSharded::SolveDevice: one set of device pointers per shard");
  // (enqueue only -- microseconds per shard: no threads)
  for (size_t g = 0; g < h_.size(); ++g)
    check(fccqp_solve_device(h_[g], in[g].Q, in[g].b, in[g].A_eq, in[g].b_eq, in[g].friction_coeffs, in[g].lb, in[g].ub,
                             in[g].stream),
          h_[g], "FCCQPSharded::SolveDevice");
}

std::vector<FCCQPSharded::ShardDeviceInputs> FCCQPSharded::SliceDeviceBatch(const double *dQ, const double *db, const double *dA_eq,
                                                                            const double *db_eq, const double *dfriction,
                                                                            const double *dlb, const double *dub) const {
  const size_t n = (size_t)n_vars_, m = (size_t)n_eq_, nf = (size_t)(nc_ / 3);
  std::vector<ShardDeviceInputs> out(h_.size());
  for (size_t g = 0; g < h_.size(); ++g) {
    const size_t lo = (size_t)shard_begin((int)g);
    out[g].Q = dQ + lo * n * n;
    out[g].b = db + lo * n;
    out[g].A_eq = dA_eq ? dA_eq + lo * m * n : nullptr;
    out[g].b_eq = db_eq ? db_eq + lo * m : nullptr;
    out[g].friction_coeffs = dfriction ? dfriction + lo * nf : nullptr;
    out[g].lb = dlb + lo * n;
    out[g].ub = dub + lo * n;
  }
  return out;
}

void FCCQPSharded::SolveAsync(const double *Q, const double *b, const double *A_eq, const double *b_eq,
                              const double *friction_coeffs, int n_friction, const double *lb, const double *ub) {
  const size_t n = (size_t)n_vars_, m = (size_t)n_eq_;
  // (through the workers: hipMemcpyAsync out of pageable memory blocks, and then the shards' copies should overlap)
  on_every_shard([&](size_t g) {
    const size_t lo = (size_t)shard_begin((int)g);
    return fccqp_solve_async(h_[g], Q + lo * n * n, b + lo * n, A_eq ? A_eq + lo * m * n : nullptr,
                             b_eq ? b_eq + lo * m : nullptr,
                             friction_coeffs ? friction_coeffs + lo * (size_t)n_friction : nullptr, n_friction,
                             lb + lo * n, ub + lo * n);
  }, "FCCQPSharded::SolveAsync");
}

void FCCQPSharded::GetZAsync(double *z_pinned) {
  for (size_t g = 0; g < h_.size(); ++g)
    check(fccqp_get_z_async(h_[g], z_pinned + (size_t)shard_begin((int)g) * (size_t)n_vars_), h_[g], "FCCQPSharded::GetZAsync");
}

void FCCQPSharded::Synchronize() {
  for (auto *h : h_) check(fccqp_synchronize(h), h, "FCCQPSharded::Synchronize");
}

FCCQPBatchSolution FCCQPSharded::GetSolution() const {
  FCCQPBatchSolution out;
  out.batch = batch_;
  out.num_vars = n_vars_;
  out.z.assign((size_t)batch_ * (size_t)n_vars_, 0.0);
  std::vector<fccqp_details> d((size_t)batch_);
  on_every_shard([&](size_t g) {
    const size_t lo = (size_t)shard_begin((int)g);
    return fccqp_get_solution(h_[g], out.z.data() + lo * (size_t)n_vars_, d.data() + lo);
  }, "FCCQPSharded::GetSolution");
  out.details.reserve((size_t)batch_);
  for (const auto &x : d) out.details.push_back(from_c(x));
  return out;
}

float FCCQPSharded::LastKernelMilliseconds() const {
  float worst = 0.f;
  for (auto *h : h_) {
    float ms = 0.f;
    check(fccqp_last_kernel_ms(h, &ms), h, "FCCQPSharded::LastKernelMilliseconds");
    worst = ms > worst ? ms : worst;
  }
  return worst;
}

}  // namespace fcc_qp
