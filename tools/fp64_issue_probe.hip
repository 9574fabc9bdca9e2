// fp64_issue_probe.hip -- cycles per instruction of the fp64 building blocks of the fused kernel,
// ONE wave per SIMD (the fused kernel's occupancy): v_fma_f64, v_fmac_f64_dpp row_newbcast,
// v_mfma_f64_16x16x4, v_mfma_f64_4x4x4, v_readlane pair + fma, v_rsq_f64 / v_rcp_f64, each as
// independent and as dependent chains.  clock64() = s_memtime (shader clock).
//   hipcc --offload-arch=gfx950 -O3 -o fp64_issue_probe fp64_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = double __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, long long *cyc, int iters) {
  const int lane = threadIdx.x;
  double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-6;
  double acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = lane * 0.1 + i;
  d4 t[4];
  for (int i = 0; i < 4; ++i) t[i] = d4{a, b, a, b};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {          // 8 independent fma chains, 16 fma
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 7] = fma(acc[u & 7], b, a);
    } else if constexpr (MODE == 1) {   // 1 dependent chain, 16 fma
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[0] = fma(acc[0], b, a);
    } else if constexpr (MODE == 2) {   // 16 dpp fmac, 8 accumulators
#pragma unroll
      for (int u = 0; u < 16; ++u)
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[u & 7]) : "v"(a), "v"(b));
    } else if constexpr (MODE == 3) {   // 16 mfma 16x16x4, 4 independent accumulators
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, t[u & 3], 0, 0, 0);
    } else if constexpr (MODE == 4) {   // 16 mfma 16x16x4, dependent
#pragma unroll
      for (int u = 0; u < 16; ++u) t[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, t[0], 0, 0, 0);
    } else if constexpr (MODE == 5) {   // 16 mfma 4x4x4 (4 blocks), 8 independent
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 7] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[u & 7], 0, 0, 0);
    } else if constexpr (MODE == 6) {   // 16 x (readlane pair + fma with the scalar)
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        int lo = __builtin_amdgcn_readlane(__double2loint(acc[7]), u), hi = __builtin_amdgcn_readlane(__double2hiint(acc[7]), u);
        acc[u & 3] = fma(__hiloint2double(hi, lo), b, acc[u & 3]);
      }
    } else if constexpr (MODE == 7) {   // 16 dependent rsq
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[0] = __builtin_amdgcn_rsq(acc[0] + 2.0);
    } else if constexpr (MODE == 8) {   // 16 v_mul_f64 independent
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 7] = acc[u & 7] * b;
    } else if constexpr (MODE == 9) {   // 16 v_cndmask_b32 pairs (select of a double)
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 7] = (acc[(u + 1) & 7] > 0.5) ? acc[u & 7] : b;
    } else if constexpr (MODE == 10) {  // mfma 16x16x4 interleaved 1:1 with independent v_fma_f64
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        t[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, t[u & 3], 0, 0, 0);
        acc[u & 7] = fma(acc[u & 7], b, a);
      }
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  for (int i = 0; i < 4; ++i) s += t[i][0] + t[i][1] + t[i][2] + t[i][3];
  out[blockIdx.x * 64 + lane] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, double *out, long long *cyc) {
  const int blocks = 1024, iters = 2000;   // 1024 one-wave blocks, 64 KB LDS not needed: force 1 wave/SIMD by grid size
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
  }
  static long long h[1024];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
  printf("%-58s %7.2f clock64 ticks per instruction (group)\n", name, s / blocks / iters / 16);
}

// accuracy of v_rsq_f64 refined by 0, 1 and 2 Newton steps (the cone projection's reciprocal square root)
__global__ void rsq_accuracy(const double *a, double *r0, double *r1, double *r2, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const double x = a[i];
  double r = __builtin_amdgcn_rsq(x);
  r0[i] = r;
  double e = fma(-x * r, r, 1.0);
  r = fma(0.5 * r, e, r);
  r1[i] = r;
  e = fma(-x * r, r, 1.0);
  r = fma(0.5 * r, e, r);
  r2[i] = r;
}
#include <cmath>
#include <vector>
#include <random>
static void rsq_report() {
  const int count = 1 << 20;
  std::vector<double> a(count), h0(count), h1(count), h2(count);
  std::mt19937_64 gen(7);
  std::uniform_real_distribution<double> ex(-40.0, 40.0);
  for (auto &v : a) v = std::exp2(ex(gen));
  double *da, *d0, *d1, *d2;
  (void)hipMalloc(&da, count * 8); (void)hipMalloc(&d0, count * 8); (void)hipMalloc(&d1, count * 8); (void)hipMalloc(&d2, count * 8);
  (void)hipMemcpy(da, a.data(), count * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(rsq_accuracy, dim3(count / 256), dim3(256), 0, 0, da, d0, d1, d2, count);
  (void)hipMemcpy(h0.data(), d0, count * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(h1.data(), d1, count * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(h2.data(), d2, count * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < count; ++i) {
    const long double ref = 1.0L / sqrtl((long double)a[i]);
    m0 = std::fmax(m0, (double)fabsl((h0[i] - ref) / ref));
    m1 = std::fmax(m1, (double)fabsl((h1[i] - ref) / ref));
    m2 = std::fmax(m2, (double)fabsl((h2[i] - ref) / ref));
  }
  printf("v_rsq_f64 max relative error over 2^20 samples: raw %.3e, one Newton step %.3e, two %.3e (eps = %.3e)\n", m0, m1, m2,
         2.220446049250313e-16);
}

int main() {
  rsq_report();
  double *out; long long *cyc;
  (void)hipMalloc(&out, 1024 * 64 * 8); (void)hipMalloc(&cyc, 1024 * 8);
  run<0>("v_fma_f64, 8 chains", out, cyc);
  run<1>("v_fma_f64, 1 chain", out, cyc);
  run<2>("v_fmac_f64_dpp row_newbcast, 8 accumulators", out, cyc);
  run<3>("v_mfma_f64_16x16x4, 4 accumulators", out, cyc);
  run<4>("v_mfma_f64_16x16x4, dependent", out, cyc);
  run<5>("v_mfma_f64_4x4x4_4b, 8 accumulators", out, cyc);
  run<6>("2 v_readlane + v_fma_f64(sgpr)", out, cyc);
  run<7>("v_add + v_rsq_f64, dependent", out, cyc);
  run<8>("v_mul_f64, 8 chains", out, cyc);
  run<9>("v_cmp_f64 + 2 v_cndmask", out, cyc);
  run<10>("mfma 16x16x4 + v_fma_f64 pair (per pair)", out, cyc);
  return 0;
}
