// fccqp_check.hip -- the small kernels around a solve.  (1) Self-check of a finished solve: does the returned z satisfy
// the equality constraints?  (2) order_by_iterations_kernel, further down: the block -> instance order of a chain's next solve.
//
// Every primal iterate of the path is the head of a KKT solve (reference /root/reference/src/fcc_qp.cpp:86-87 and, for
// the presolve, :174-178), so A_eq z = b_eq holds at exit to rounding whatever the ADMM iteration count -- unless the
// solve itself went wrong.  The kernels for 256 < N <= 512 (generic kernel, KKT matrix in a global workspace) once
// returned wrong iterates on some boxes for reasons of code generation; the cause was found
// and is guarded by code-object tests, but a silent wrong z is the one failure a drop-in must not have, so for those
// shapes (and on request for any shape: fccqp_set_self_check) this kernel runs behind every solve and raises
// FCCQP_FLAG_EQ_RESIDUAL on an instance whose z is not finite or leaves
//     |A_eq z - b_eq|_i  >  tol * (|b_eq,i| + sum_j |A_ij z_j|)      for some row i      (tol = 1e-6).
// One wavefront per instance, rows one after the other, lanes along a row (coalesced); HBM-bound: m n words per
// instance, a few per cent of what the solve itself reads and writes for these shapes.
#include "fccqp_device.h"

namespace fccqp {
namespace {

constexpr int kWave = 64;

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

template <class TIN>
__global__ __launch_bounds__(kWave) void eq_residual_kernel(const KernelArgs a, const double tol) {
  const int inst = blockIdx.x, lane = threadIdx.x;
  if (inst >= a.batch) return;
  const int n = a.n, m = a.m;
  const TIN *__restrict__ A = reinterpret_cast<const TIN *>(a.A) + (size_t)inst * m * n;
  const double *__restrict__ beq = a.b_eq + (size_t)inst * m;
  const double *__restrict__ z = a.x + (size_t)inst * n;
  bool bad = false;
  for (int j = lane; j < n; j += kWave) bad = bad || !(fabs(z[j]) <= 1.79769313486231570815e+308);   // NaN or infinity
  for (int i = 0; i < m; ++i) {
    double acc = 0.0, mag = 0.0;
    for (int j = lane; j < n; j += kWave) {
      const double t = (double)A[(size_t)i * n + j] * z[j];
      acc += t;
      mag += fabs(t);
    }
    acc = wave_sum_f64(acc);
    mag = wave_sum_f64(mag);
    const double b = beq[i];
    // (written so that a NaN on either side counts as a failure)
    bad = bad || !(fabs(acc - b) <= tol * (fabs(b) + mag));
  }
  const bool any = __builtin_amdgcn_ballot_w64(bad) != 0;
  if (lane == 0 && any) a.flags[inst] |= kFlagEqResidual;
}

// ---- block -> instance order for the next solve of a chain: longest first --------------------------------------------
// order[0 .. batch): the instances sorted by the iteration count of the PREVIOUS solve, descending (counting sort on
// min(n_iter, 1023); ties in the order the atomics fall, which results never depend on).  One workgroup, ~5 us for 4096
// instances.  Why: the workgroups of a launch are handed to the SIMDs in block order as slots fall free -- list
// scheduling -- and an instance is one indivisible job of 38 us (leaves after its first iteration) to 61 us (all 100):
// in a random order a four-round launch ends (longest job - mean job) after the work runs out, longest-first it ends
// with it (tools/schedule_sim.py).  Before a solve nobody knows n_iter -- except in a chain
// (fcc_qp_test.py:77-89: the same robot a tick later, warm-started), where the last count is a good guess at the next.
__global__ __launch_bounds__(1024) void order_by_iterations_kernel(const int *__restrict__ n_iter, const int batch,
                                                                    int *__restrict__ order) {
  __shared__ int hist[1024], scan[1024];
  const int tid = threadIdx.x;
  auto key = [](int it) { return it < 0 ? 0 : (it > 1023 ? 1023 : it); };
  hist[tid] = 0;
  __syncthreads();
  for (int i = tid; i < batch; i += 1024) atomicAdd(&hist[key(n_iter[i])], 1);
  __syncthreads();
  // start[k] = number of instances with a larger key: exclusive scan over the keys in descending order
  const int mine = hist[1023 - tid];
  scan[tid] = mine;
  for (int off = 1; off < 1024; off <<= 1) {
    __syncthreads();
    const int t = (tid >= off) ? scan[tid - off] : 0;
    __syncthreads();
    scan[tid] += t;
  }
  __syncthreads();
  hist[1023 - tid] = scan[tid] - mine;
  __syncthreads();
  for (int i = tid; i < batch; i += 1024) order[atomicAdd(&hist[key(n_iter[i])], 1)] = i;
}

}  // namespace

hipError_t launch_order_by_iterations(const int *n_iter, int batch, int *order, hipStream_t stream) {
  if (batch <= 0) return hipSuccess;
  hipLaunchKernelGGL(order_by_iterations_kernel, dim3(1), dim3(1024), 0, stream, n_iter, batch, order);
  return hipGetLastError();
}

hipError_t launch_eq_residual_check(const KernelArgs &a, hipStream_t stream, bool qa_f32, double tol) {
  if (a.batch <= 0) return hipSuccess;
  if (qa_f32) hipLaunchKernelGGL((eq_residual_kernel<float>), dim3(a.batch), dim3(kWave), 0, stream, a, tol);
  else hipLaunchKernelGGL((eq_residual_kernel<double>), dim3(a.batch), dim3(kWave), 0, stream, a, tol);
  return hipGetLastError();
}

}  // namespace fccqp
