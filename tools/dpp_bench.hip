// Microbenchmark: 60 x 60 mat-vec per lane = row, the vector operand delivered three ways:
//  (a) LDS broadcast reads (ds_read_b128 of two doubles every lane reads), as the fused kernel's P4 did
//  (b) v_fmac_f64_dpp row_newbcast: the vector sits 16 entries per lane row, no LDS traffic
//  (c) plain register operands (no broadcast at all: the FMA-issue floor)
// One wave per workgroup, 4 workgroups per CU (39 KB of LDS each), like the fused kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/dpp_bench.hip -o tools/dpp_bench && tools/dpp_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using d2 = double __attribute__((ext_vector_type(2)));
constexpr int n = 60, ITERS = 2000;

template <int MODE>
__global__ __launch_bounds__(64, 1) void k(double *out, const double *in) {
  __shared__ double lds[4992];
  const int lane = threadIdx.x;
  double M[n];
#pragma unroll
  for (int j = 0; j < n; ++j) M[j] = in[(lane * n + j) % 4096] * 1e-3;
  double x = in[lane];
  for (int it = 0; it < ITERS; ++it) {
    double a0 = 1.0, a1 = 0, a2 = 0, a3 = 0;
    if (MODE == 0) {
      lds[lane] = x;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
      const d2 *w2 = reinterpret_cast<const d2 *>(lds);
#pragma unroll
      for (int jp = 0; jp < n / 2; ++jp) {
        const d2 w = w2[jp];
        if (jp & 1) { a2 = fma(M[2 * jp], w.x, a2); a3 = fma(M[2 * jp + 1], w.y, a3); }
        else { a0 = fma(M[2 * jp], w.x, a0); a1 = fma(M[2 * jp + 1], w.y, a1); }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    } else if (MODE == 1) {
      lds[lane] = x;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
      const int l = lane & 15;
      double W[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) W[q] = lds[16 * q + l];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#define FM(acc, q, l_) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #l_ " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(W[q]), "v"(M[16 * q + l_]))
#define ROW(q) FM(a0, q, 0); FM(a1, q, 1); FM(a2, q, 2); FM(a3, q, 3); FM(a0, q, 4); FM(a1, q, 5); FM(a2, q, 6); FM(a3, q, 7); \
               FM(a0, q, 8); FM(a1, q, 9); FM(a2, q, 10); FM(a3, q, 11);
      ROW(0) FM(a0, 0, 12); FM(a1, 0, 13); FM(a2, 0, 14); FM(a3, 0, 15);
      ROW(1) FM(a0, 1, 12); FM(a1, 1, 13); FM(a2, 1, 14); FM(a3, 1, 15);
      ROW(2) FM(a0, 2, 12); FM(a1, 2, 13); FM(a2, 2, 14); FM(a3, 2, 15);
      ROW(3)
    } else {
#pragma unroll
      for (int jp = 0; jp < n / 2; ++jp) {
        if (jp & 1) { a2 = fma(M[2 * jp], x, a2); a3 = fma(M[2 * jp + 1], x, a3); }
        else { a0 = fma(M[2 * jp], x, a0); a1 = fma(M[2 * jp + 1], x, a1); }
      }
    }
    x = (a0 + a1) + (a2 + a3);
    x = (x > 4.0) ? x - 4.0 : x;
  }
  out[blockIdx.x * 64 + lane] = x;
}

int main() {
  const int blocks = 4096;
  double *out, *in;
  hipMalloc(&out, blocks * 64 * 8); hipMalloc(&in, 4096 * 8);
  std::vector<double> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = 0.001 * (i % 97) + 0.01;
  hipMemcpy(in, h.data(), 4096 * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, const char *name) {
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
      hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, in); hipEventRecord(e1);
      hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    std::vector<double> o(64); hipMemcpy(o.data(), out, 64 * 8, hipMemcpyDeviceToHost);
    // 4096 blocks on 1024 wave slots: 4 rounds; per-iteration time of one wave
    printf("%-28s %8.3f ms  -> %7.1f ns per 60x60 mat-vec per wave (4 waves/CU)  check %.6f %.6f\n", name, best,
           1e6 * best / (4.0 * ITERS), o[3], o[40]);
  };
  run(k<0>, "LDS broadcast (ds_read_b128)");
  run(k<1>, "v_fmac_f64_dpp row_newbcast");
  run(k<2>, "register operand (floor)");
  return 0;
}
