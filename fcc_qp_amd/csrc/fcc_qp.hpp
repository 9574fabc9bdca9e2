// fcc_qp.hpp -- host-side C++ surface of the MI355X solver: the reference's class fcc_qp::FCCQP
// (reference src/fcc_qp.hpp:14-172) re-expressed on top of the C ABI in include/fccqp.h, plus the
// batched class FCCQPBatch the reference does not have.
//
// Same type names, member names, defaults and call sequence as the reference
// (set_options / set_warm_start / Solve / GetSolution).  Eigen is not a dependency: the
// Eigen::Ref<const MatrixXd/VectorXd> arguments of the reference become the non-owning strided
// views below (any row-/column-major, contiguous or strided double array converts for free).
// All arithmetic happens on the GPU; there is no CPU path -- constructing a solver without a HIP
// device throws.
#pragma once
#include <cstddef>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>


struct fccqp_solver;  // include/fccqp.h

namespace fcc_qp {

// reference src/fcc_qp.hpp:14-17
enum FCCQPSolveStatus { kSuccess = 0, kMaxIterations = 1 };

// reference src/fcc_qp.hpp:19-28 (+ flags: FCCQP_FLAG_* diagnostics of include/fccqp.h)
struct FCCQPDetails {
  int n_iter{};
  double admm_residual_bounds{};
  double admm_residual_friction_cone{};
  double solve_time{};
  double factorization_time{};
  double bounds_viol{};
  double friction_cone_viol{};
  FCCQPSolveStatus solve_status{kSuccess};
  int flags{};
};

// reference src/fcc_qp.hpp:30-35
struct FCCQPOptions {
  int max_iter = 1000;
  double rho = 1e-6;
  double eps_fcone = 1e-3;
  double eps_bound = 1e-6;
};

// reference src/fcc_qp.hpp:37-40 (VectorXd z -> std::vector<double>)
struct FCCQPSolution {
  FCCQPDetails details{};
  std::vector<double> z;
};

// Non-owning strided views standing in for Eigen::Ref<const MatrixXd> / Ref<const VectorXd>.
struct ConstMatrixView {
  const double *data = nullptr;
  long rows = 0, cols = 0;
  long row_stride = 0, col_stride = 0;  // in elements
  ConstMatrixView() = default;
  ConstMatrixView(const double *d, long r, long c, long rs, long cs)
      : data(d), rows(r), cols(c), row_stride(rs), col_stride(cs) {}
  static ConstMatrixView RowMajor(const double *d, long r, long c) { return {d, r, c, c, 1}; }
  static ConstMatrixView ColMajor(const double *d, long r, long c) { return {d, r, c, 1, r}; }
  double operator()(long i, long j) const { return data[i * row_stride + j * col_stride]; }
};
struct ConstVectorView {
  const double *data = nullptr;
  long size = 0;
  long stride = 1;
  ConstVectorView() = default;
  ConstVectorView(const double *d, long n, long s = 1) : data(d), size(n), stride(s) {}
  ConstVectorView(const std::vector<double> &v) : data(v.data()), size((long)v.size()), stride(1) {}
  double operator()(long i) const { return data[i * stride]; }
};

class FccqpError : public std::runtime_error {
 public:
  FccqpError(int code, const std::string &what) : std::runtime_error(what), code_(code) {}
  int code() const { return code_; }
 private:
  int code_;
};

/*!
 * Drop-in for the reference class (src/fcc_qp.hpp:54-172): solves
 *   minimize (1/2) x'Qx + b'x   s.t.  A_eq x = b_eq,  lb <= x <= ub,  x[lambda_c] in friction cones
 * one instance per Solve call, on the GPU (one wavefront).
 */
class FCCQP {
 public:
  // reference ctor src/fcc_qp.cpp:24-55; `device` is new (HIP device index)
  FCCQP(int num_vars, int num_equality_constraints, int nc, int lambda_c_start, int device = 0);
  ~FCCQP();
  FCCQP(const FCCQP &) = delete;
  FCCQP &operator=(const FCCQP &) = delete;

  void set_rho(double rho);              // src/fcc_qp.hpp:75-78
  void set_max_iter(int n);              // :80-83
  void set_options(FCCQPOptions opt);    // :85-87
  void set_warm_start(bool warm_start);  // :89-91
  FCCQPOptions get_options() const { return options_; }

  // reference src/fcc_qp.hpp:114-117 / src/fcc_qp.cpp:114-191
  void Solve(const ConstMatrixView &Q, const ConstVectorView &b, const ConstMatrixView &A_eq,
             const ConstVectorView &b_eq, const std::vector<double> &friction_coeffs,
             const ConstVectorView &lb, const ConstVectorView &ub);

  FCCQPSolution GetSolution() const;     // src/fcc_qp.cpp:194-207
  int contact_vars_start() const { return lambda_c_start_; }  // src/fcc_qp.hpp:121
  fccqp_solver *handle() const { return h_; }

 private:
  FCCQPOptions options_;
  const int n_vars_, nc_, lambda_c_start_;
  int n_eq_;
  fccqp_solver *h_ = nullptr;
  std::vector<double> pack_;  // contiguous row-major staging of the seven inputs
};

struct FCCQPBatchSolution {
  int batch = 0, num_vars = 0;
  std::vector<double> z;               // [batch][num_vars]
  std::vector<FCCQPDetails> details;   // [batch]
};

/*!
 * `batch` independent QPs of one shape solved by one kernel launch, one wavefront per QP.
 * Inputs are instance-major contiguous arrays (include/fccqp.h).  Warm start keeps the
 * per-instance (x, mu_x, mu_lambda_c) resident on the device between calls.
 */
class FCCQPBatch {
 public:
  FCCQPBatch(int num_vars, int num_equality_constraints, int nc, int lambda_c_start, int batch,
             int device = 0);
  ~FCCQPBatch();
  FCCQPBatch(const FCCQPBatch &) = delete;
  FCCQPBatch &operator=(const FCCQPBatch &) = delete;

  void set_rho(double rho);
  void set_max_iter(int n);
  void set_options(FCCQPOptions opt);
  void set_warm_start(bool warm_start);
  void set_kernel(int kernel);  // FCCQP_KERNEL_*
  void set_kernel_timing(bool enabled);  // fccqp_set_kernel_timing: the start event of every solve (on by default)
  void set_self_check(bool enabled);     // fccqp_set_self_check: A_eq z = b_eq re-checked behind every solve (FLAG_EQ_RESIDUAL)
  // fccqp_set_dispatch_order: the next launch of a chain longest instance first, by the previous solve's iteration counts
  // (one-wave kernel of a built-in shape, batches of >= 8 instances per compute unit; never an effect on results)
  void set_dispatch_order_by_previous_iterations(bool enabled);
  int kernel_in_use() const;
  FCCQPOptions get_options() const { return options_; }

  // host pointers; n_friction = friction coefficients per instance (>= nc/3)
  void Solve(const double *Q, const double *b, const double *A_eq, const double *b_eq,
             const double *friction_coeffs, int n_friction, const double *lb, const double *ub);
  // device pointers, asynchronous on `stream` (hipStream_t; nullptr = HIP's default stream)
  void SolveDevice(const double *dQ, const double *db, const double *dA_eq, const double *db_eq,
                   const double *dfriction, const double *dlb, const double *dub, void *stream);
  // fp32-storage inputs (float arrays of the same shapes; widened on load, arithmetic stays fp64:
  // bit-identical to Solve on the widened values, half the input traffic; include/fccqp.h)
  void Solve(const float *Q, const float *b, const float *A_eq, const float *b_eq,
             const float *friction_coeffs, int n_friction, const float *lb, const float *ub);
  void SolveDevice(const float *dQ, const float *db, const float *dA_eq, const float *db_eq,
                   const float *dfriction, const float *dlb, const float *dub, void *stream);
  void Synchronize();
  FCCQPBatchSolution GetSolution() const;
  float LastKernelMilliseconds() const;
  // the same with how it was measured (FCCQP_TIMING_*: device events, host clock of the low-latency path, or none)
  std::pair<float, int> LastTiming() const;
  long long AlgorithmicBytes() const;

  int batch() const { return batch_; }
  int num_vars() const { return n_vars_; }
  int num_equality_constraints() const { return n_eq_; }
  int nc() const { return nc_; }
  int contact_vars_start() const { return lambda_c_start_; }
  fccqp_solver *handle() const { return h_; }

 private:
  FCCQPOptions options_;
  const int n_vars_, n_eq_, nc_, lambda_c_start_, batch_;
  fccqp_solver *h_ = nullptr;
};

/*!
 * One instance-major host batch split over several solver handles -- normally one per GPU of the
 * node: shard g takes the instances [g B / G, (g + 1) B / G) and solves them
 * on device `devices[g]`, one host thread per shard; results are gathered back into one
 * instance-major array.  No data-path collective: every QP is independent (all reference state is per
 * instance, src/fcc_qp.hpp:130-171), so nothing crosses between the devices; the warm-start state
 * of a shard stays resident on its own device between solves.  The same device may be listed more
 * than once (the 1-GPU test runs two shards on device 0).
 */
class FCCQPSharded {
 public:
  FCCQPSharded(int num_vars, int num_equality_constraints, int nc, int lambda_c_start, int batch,
               const std::vector<int> &devices);
  ~FCCQPSharded();
  FCCQPSharded(const FCCQPSharded &) = delete;
  FCCQPSharded &operator=(const FCCQPSharded &) = delete;

  void set_options(FCCQPOptions opt);
  void set_warm_start(bool warm_start);
  void set_kernel(int kernel);
  void set_kernel_timing(bool enabled);
  FCCQPOptions get_options() const { return options_; }

  // host pointers to the whole batch; n_friction = friction coefficients per instance (>= nc/3).  Blocking: every
  // shard's persistent worker thread copies its slice out of (pageable) host memory, solves and waits.
  void Solve(const double *Q, const double *b, const double *A_eq, const double *b_eq,
             const double *friction_coeffs, int n_friction, const double *lb, const double *ub);
  // Device-resident inputs, asynchronous: shard g's instances [shard_begin(g), shard_begin(g + 1)) lie in the seven
  // instance-major arrays of in[g], resident on device_of(g); the solve is enqueued on in[g].stream (nullptr: HIP's
  // default stream; the default argument: the shard's own stream) and the call returns.  This is the path that keeps
  // several GPUs at their device-resident rate from ONE process (Solve above is bound by the PCIe copies).
  struct ShardDeviceInputs {
    const double *Q = nullptr, *b = nullptr, *A_eq = nullptr, *b_eq = nullptr, *friction_coeffs = nullptr, *lb = nullptr,
                 *ub = nullptr;
    void *stream = reinterpret_cast<void *>(static_cast<std::ptrdiff_t>(-1));      // FCCQP_STREAM_OWN
  };
  void SolveDevice(const std::vector<ShardDeviceInputs> &in);
  // One instance-major batch resident on ONE device (every shard on that device: tests, or several handles sharing a
  // GPU): the shard pointers are the base pointers advanced to shard_begin(g).
  std::vector<ShardDeviceInputs> SliceDeviceBatch(const double *dQ, const double *db, const double *dA_eq, const double *db_eq,
                                                  const double *dfriction, const double *dlb, const double *dub) const;
  // Page-locked host memory, asynchronous: every shard enqueues the copies of its slice, its kernel and (GetZAsync)
  // the copy of its slice of z into `z_pinned` [batch][num_vars] on its own stream and returns; the arrays stay
  // borrowed until Synchronize().  n_friction must be nc/3 here.  (Pageable arrays work too, but their copies block.)
  void SolveAsync(const double *Q, const double *b, const double *A_eq, const double *b_eq,
                  const double *friction_coeffs, int n_friction, const double *lb, const double *ub);
  void GetZAsync(double *z_pinned);
  void Synchronize();
  FCCQPBatchSolution GetSolution() const;
  // slowest shard's device time for the last Solve (ms)
  float LastKernelMilliseconds() const;

  int batch() const { return batch_; }
  int shards() const { return (int)h_.size(); }
  int shard_begin(int g) const { return (int)(((long long)batch_ * g) / (long long)h_.size()); }
  int device_of(int g) const { return devices_[(size_t)g]; }
  int num_vars() const { return n_vars_; }
  int num_equality_constraints() const { return n_eq_; }
  int nc() const { return nc_; }

 private:
  FCCQPOptions options_;
  const int n_vars_, n_eq_, nc_, lambda_c_start_, batch_;
  std::vector<int> devices_;
  std::vector<fccqp_solver *> h_;
  // one persistent host thread per shard (created once: a solve call hands every shard's work to its thread and waits)
  class Workers;
  std::unique_ptr<Workers> workers_;
  void on_every_shard(const std::function<int(std::size_t)> &fn, const char *where) const;
};

}  // namespace fcc_qp
