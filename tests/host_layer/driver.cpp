// Driver of the sanitizer build: exercises the host-side C++ classes (fcc_qp_amd/csrc/fcc_qp.{hpp,cpp}) on
// top of the stub C ABI -- view packing of FCCQP::Solve (strided, column-major, row-major), the
// error mapping, and the shard arithmetic / threads / gather of FCCQPSharded with 1..4 shards over
// batches that do not divide evenly.  Exit code 0 = all checks passed (ASan/UBSan abort otherwise).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

#include "fcc_qp.hpp"

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); std::exit(1); } } while (0)

using namespace fcc_qp;

static double fsum(const std::vector<double> &v, size_t lo, size_t count) {
  double a = 0; for (size_t i = lo; i < lo + count; ++i) if (std::isfinite(v[i])) a += v[i]; return a;
}

int main() {
  const int n = 7, m = 3, nc = 3, s0 = 2;
  // ---- FCCQP::Solve: views with strides -------------------------------------------------------
  {
    FCCQP one(n, m, nc, s0);
    std::vector<double> Qc(n * n), Ar(m * n), b(2 * n), beq(m), lb(n, -std::numeric_limits<double>::infinity()), ub(n, 1.0);
    for (int i = 0; i < n * n; ++i) Qc[i] = i + 1;
    for (int i = 0; i < m * n; ++i) Ar[i] = 0.5 * i;
    for (int i = 0; i < 2 * n; ++i) b[i] = (i % 2 == 0) ? i : 1e9;     // stride-2 view must skip the 1e9s
    for (int i = 0; i < m; ++i) beq[i] = -i;
    one.Solve(ConstMatrixView::ColMajor(Qc.data(), n, n), ConstVectorView(b.data(), n, 2),
              ConstMatrixView(Ar.data(), m, n, n, 1), ConstVectorView(beq.data(), m), {0.6}, ConstVectorView(lb.data(), n),
              ConstVectorView(ub.data(), n));
    FCCQPSolution sol = one.GetSolution();
    double expect = fsum(Qc, 0, n * n) + fsum(Ar, 0, m * n) + fsum(beq, 0, m) + 0.6 + n * 1.0;
    for (int i = 0; i < n; ++i) expect += b[2 * i];
    CHECK(std::fabs(sol.z[0] - expect) < 1e-9);
    CHECK((int)sol.z.size() == n);
    bool threw = false;
    try { one.Solve(ConstMatrixView::ColMajor(Qc.data(), n, n), ConstVectorView(b.data(), n, 2),
                    ConstMatrixView(Ar.data(), m, n, n, 1), ConstVectorView(beq.data(), m), {}, ConstVectorView(lb.data(), n),
                    ConstVectorView(ub.data(), n)); } catch (const std::out_of_range &) { threw = true; }
    CHECK(threw);                                   // .at() of the reference (constraint_utils.cpp:32)
    threw = false;
    try { FCCQP bad(n, m, 4, s0); } catch (const std::exception &) { threw = true; }
    CHECK(threw);                                   // nc % 3 != 0 (fcc_qp.cpp:32)
  }
  // ---- FCCQPSharded: slices, threads, gather --------------------------------------------------------
  for (int shards = 1; shards <= 4; ++shards) {
    for (int batch : {shards, 5, 9, 64, 101}) {
      if (batch < shards) continue;
      const size_t B = batch;
      std::vector<double> Q(B * n * n), b(B * n), A(B * m * n), beq(B * m), mu(B * 2), lb(B * n, -1.0), ub(B * n, 2.0);
      for (size_t i = 0; i < Q.size(); ++i) Q[i] = 1e-3 * (double)(i % 977);
      for (size_t i = 0; i < b.size(); ++i) b[i] = (double)(i % 13);
      for (size_t i = 0; i < A.size(); ++i) A[i] = (double)(i % 7) - 3.0;
      for (size_t i = 0; i < beq.size(); ++i) beq[i] = 0.25 * (double)i;
      for (size_t i = 0; i < mu.size(); ++i) mu[i] = (i % 2 == 0) ? 0.5 : 1e6;   // second coefficient per instance is surplus
      std::vector<int> devs;
      for (int g = 0; g < shards; ++g) devs.push_back(g);
      FCCQPSharded sh(n, m, nc, s0, batch, devs);
      FCCQPOptions o; o.max_iter = 12345; sh.set_options(o);
      sh.Solve(Q.data(), b.data(), A.data(), beq.data(), mu.data(), 2, lb.data(), ub.data());
      FCCQPBatchSolution sol = sh.GetSolution();
      CHECK(sol.batch == batch && (int)sol.z.size() == batch * n && (int)sol.details.size() == batch);
      for (int i = 0; i < batch; ++i) {
        const double expect = fsum(Q, (size_t)i * n * n, n * n) + fsum(b, (size_t)i * n, n) + fsum(A, (size_t)i * m * n, m * n) +
                              fsum(beq, (size_t)i * m, m) + 0.5 + n * (-1.0) + n * 2.0;
        CHECK(std::fabs(sol.z[(size_t)i * n] - expect) < 1e-9);        // instance i solved from ITS slice ...
        CHECK(std::fabs(sol.z[(size_t)i * n + n - 1] - (expect + n - 1)) < 1e-9);
        int g = 0;
        while (g + 1 < shards && sh.shard_begin(g + 1) <= i) ++g;
        CHECK(sol.details[(size_t)i].n_iter == g * 1000 + (i - sh.shard_begin(g)));   // ... by the right handle
      }
      CHECK(sh.shard_begin(0) == 0 && sh.shard_begin(shards) == batch);
      CHECK(std::fabs(sh.LastKernelMilliseconds() - (float)shards) < 1e-6);          // slowest shard
      sh.set_warm_start(true);
      sh.Solve(Q.data(), b.data(), A.data(), beq.data(), mu.data(), 2, lb.data(), ub.data());
      CHECK(std::fabs(sh.GetSolution().z[0] - sol.z[0] - 0.5) < 1e-9);               // the flag reached the handles
      sh.set_warm_start(false);
      // device-resident form: one set of pointers per shard, sliced out of one instance-major batch (compact friction
      // coefficients: nc / 3 per instance); asynchronous page-locked form: SolveAsync + GetZAsync + Synchronize
      std::vector<double> mu1(B);
      for (size_t i = 0; i < B; ++i) mu1[i] = 0.5;
      const auto slices = sh.SliceDeviceBatch(Q.data(), b.data(), A.data(), beq.data(), mu1.data(), lb.data(), ub.data());
      CHECK((int)slices.size() == shards);
      for (int g = 0; g < shards; ++g) CHECK(slices[(size_t)g].Q == Q.data() + (size_t)sh.shard_begin(g) * n * n && slices[(size_t)g].ub == ub.data() + (size_t)sh.shard_begin(g) * n);
      sh.SolveDevice(slices);
      sh.Synchronize();
      const FCCQPBatchSolution dev = sh.GetSolution();
      CHECK(dev.z == sol.z);                                                         // same instances, same handles
      std::vector<double> zpin(B * n, -7.0);
      for (int rep = 0; rep < 3; ++rep) {                                            // (the worker threads are reused)
        sh.SolveAsync(Q.data(), b.data(), A.data(), beq.data(), mu1.data(), 1, lb.data(), ub.data());
        sh.GetZAsync(zpin.data());
        sh.Synchronize();
        CHECK(zpin == sol.z);
      }
      bool refused = false;
      try { sh.SolveDevice(std::vector<FCCQPSharded::ShardDeviceInputs>((size_t)shards + 1)); } catch (const std::invalid_argument &) { refused = true; }
      CHECK(refused);
    }
  }
  bool threw = false;
  try { FCCQPSharded bad(n, m, nc, s0, 2, {0, 1, 2}); } catch (const std::exception &) { threw = true; }
  CHECK(threw);
  threw = false;
  try { FCCQPSharded bad(n, m, nc, s0, 8, {0, 9}); } catch (const std::exception &) { threw = true; }
  CHECK(threw);                                    // no such device: handles created so far are released
  std::puts("host layer: all checks passed");
  return 0;
}
