// microbench.hip -- gfx950 instruction-rate probes that decide the fused kernel's design:
// how fast one wave / several waves per SIMD can (a) issue v_fma_f64, (b) broadcast an operand
// through LDS (same-address ds_read), (c) broadcast through v_readlane, (d) do both together.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2000;

template <int MODE>
__global__ __launch_bounds__(256) void probe(double *out, long long *cycles, const double *gsrc) {
  __shared__ double lds[2048];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2048; i += 256) lds[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  double acc0 = tid, acc1 = 1, acc2 = 2, acc3 = 3, acc4 = 4, acc5 = 5, acc6 = 6, acc7 = 7;
  double a = 1.0000001, b = 1e-9;
  unsigned base = (unsigned)((tid >> 6) * 1024);  // per-wave region (bytes), same address for all lanes
  long long t0 = clock64();
  for (int it = 0; it < ITERS; ++it) {
    unsigned addr = base + (unsigned)((it & 7) * 128);
    if constexpr (MODE == 0) {  // 16 independent-ish FMAs (8 chains)
      asm volatile(
          "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
          "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
          "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
          "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
          : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(acc4), "+v"(acc5), "+v"(acc6), "+v"(acc7)
          : "v"(a), "v"(b));
    } else if constexpr (MODE == 1) {  // 16 FMAs, 2 chains
      asm volatile(
          "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
          "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
          "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
          "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
          : "+v"(acc0), "+v"(acc1) : "v"(a), "v"(b));
    } else if constexpr (MODE == 2) {  // 16 FMAs, 1 chain (latency)
      asm volatile(
          "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
          "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
          "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
          "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
          : "+v"(acc0) : "v"(a), "v"(b));
    } else if constexpr (MODE == 3) {  // 16 same-address ds_read_b64 (broadcast), then wait
      double r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile(
          "ds_read_b64 %0, %8 offset:0\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:16\n ds_read_b64 %3, %8 offset:24\n"
          "ds_read_b64 %4, %8 offset:32\n ds_read_b64 %5, %8 offset:40\n ds_read_b64 %6, %8 offset:48\n ds_read_b64 %7, %8 offset:56\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(addr));
      acc0 += r0; acc1 += r1; acc2 += r2; acc3 += r3; acc4 += r4; acc5 += r5; acc6 += r6; acc7 += r7;
      asm volatile(
          "ds_read_b64 %0, %8 offset:64\n ds_read_b64 %1, %8 offset:72\n ds_read_b64 %2, %8 offset:80\n ds_read_b64 %3, %8 offset:88\n"
          "ds_read_b64 %4, %8 offset:96\n ds_read_b64 %5, %8 offset:104\n ds_read_b64 %6, %8 offset:112\n ds_read_b64 %7, %8 offset:120\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(addr));
      acc0 += r0; acc1 += r1; acc2 += r2; acc3 += r3; acc4 += r4; acc5 += r5; acc6 += r6; acc7 += r7;
    } else if constexpr (MODE == 4) {  // 8 same-address ds_read_b128 = 16 doubles
      using d2 = __attribute__((ext_vector_type(2))) double;
      d2 r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile(
          "ds_read_b128 %0, %8 offset:0\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
          "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n ds_read_b128 %7, %8 offset:112\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(addr));
      acc0 += r0.x + r0.y; acc1 += r1.x + r1.y; acc2 += r2.x + r2.y; acc3 += r3.x + r3.y;
      acc4 += r4.x + r4.y; acc5 += r5.x + r5.y; acc6 += r6.x + r6.y; acc7 += r7.x + r7.y;
    } else if constexpr (MODE == 5) {  // the target pattern: 8 x b128 broadcast feeding 16 FMAs
      using d2 = __attribute__((ext_vector_type(2))) double;
      d2 r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile(
          "ds_read_b128 %0, %8 offset:0\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
          "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n ds_read_b128 %7, %8 offset:112\n"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(addr));
      // FMAs of the PREVIOUS group's data would overlap in a real kernel; here: wait then 16 FMAs
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc0 = fma(r0.x, a, acc0); acc1 = fma(r0.y, a, acc1); acc2 = fma(r1.x, a, acc2); acc3 = fma(r1.y, a, acc3);
      acc4 = fma(r2.x, a, acc4); acc5 = fma(r2.y, a, acc5); acc6 = fma(r3.x, a, acc6); acc7 = fma(r3.y, a, acc7);
      acc0 = fma(r4.x, a, acc0); acc1 = fma(r4.y, a, acc1); acc2 = fma(r5.x, a, acc2); acc3 = fma(r5.y, a, acc3);
      acc4 = fma(r6.x, a, acc4); acc5 = fma(r6.y, a, acc5); acc6 = fma(r7.x, a, acc6); acc7 = fma(r7.y, a, acc7);
    } else if constexpr (MODE == 6) {  // 16 v_readlane_b32 (= 8 doubles broadcast)
      int s0, s1, s2, s3, s4, s5, s6, s7;
      asm volatile(
          "v_readlane_b32 %0, %8, 3\n v_readlane_b32 %1, %9, 3\n v_readlane_b32 %2, %8, 7\n v_readlane_b32 %3, %9, 7\n"
          "v_readlane_b32 %4, %8, 11\n v_readlane_b32 %5, %9, 11\n v_readlane_b32 %6, %8, 15\n v_readlane_b32 %7, %9, 15\n"
          : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7)
          : "v"(__double2loint(acc0)), "v"(__double2hiint(acc0)));
      asm volatile(
          "v_readlane_b32 %0, %8, 19\n v_readlane_b32 %1, %9, 19\n v_readlane_b32 %2, %8, 23\n v_readlane_b32 %3, %9, 23\n"
          "v_readlane_b32 %4, %8, 27\n v_readlane_b32 %5, %9, 27\n v_readlane_b32 %6, %8, 31\n v_readlane_b32 %7, %9, 31\n"
          : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7)
          : "v"(__double2loint(acc1)), "v"(__double2hiint(acc1)));
      acc2 += (double)(s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7) * 1e-300;
    } else if constexpr (MODE == 7) {  // 8 x (2 readlane + fma with SGPR operand)
      asm volatile(
          "v_readlane_b32 s20, %8, 3\n v_readlane_b32 s21, %9, 3\n v_readlane_b32 s22, %8, 7\n v_readlane_b32 s23, %9, 7\n"
          "v_readlane_b32 s24, %8, 11\n v_readlane_b32 s25, %9, 11\n v_readlane_b32 s26, %8, 15\n v_readlane_b32 s27, %9, 15\n"
          "v_fma_f64 %0, %10, s[20:21], %0\n v_fma_f64 %1, %10, s[22:23], %1\n v_fma_f64 %2, %10, s[24:25], %2\n v_fma_f64 %3, %10, s[26:27], %3\n"
          "v_readlane_b32 s20, %8, 19\n v_readlane_b32 s21, %9, 19\n v_readlane_b32 s22, %8, 23\n v_readlane_b32 s23, %9, 23\n"
          "v_readlane_b32 s24, %8, 27\n v_readlane_b32 s25, %9, 27\n v_readlane_b32 s26, %8, 31\n v_readlane_b32 s27, %9, 31\n"
          "v_fma_f64 %4, %10, s[20:21], %4\n v_fma_f64 %5, %10, s[22:23], %5\n v_fma_f64 %6, %10, s[24:25], %6\n v_fma_f64 %7, %10, s[26:27], %7\n"
          : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(acc4), "+v"(acc5), "+v"(acc6), "+v"(acc7)
          : "v"(__double2loint(a)), "v"(__double2hiint(a)), "v"(b)
          : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    } else if constexpr (MODE == 8) {  // s_load_dwordx16 (8 doubles) from global + 8 FMAs with SGPR operands
      const double *p = gsrc + ((it & 63) * 8);
      asm volatile(
          "s_load_dwordx16 s[36:51], %8, 0x0\n s_waitcnt lgkmcnt(0)\n"
          "v_fma_f64 %0, %9, s[36:37], %0\n v_fma_f64 %1, %9, s[38:39], %1\n v_fma_f64 %2, %9, s[40:41], %2\n v_fma_f64 %3, %9, s[42:43], %3\n"
          "v_fma_f64 %4, %9, s[44:45], %4\n v_fma_f64 %5, %9, s[46:47], %5\n v_fma_f64 %6, %9, s[48:49], %6\n v_fma_f64 %7, %9, s[50:51], %7\n"
          : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(acc4), "+v"(acc5), "+v"(acc6), "+v"(acc7)
          : "s"(p), "v"(b)
          : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51");
    } else if constexpr (MODE == 9) {  // v_mov_b32 DPP row broadcast-ish: 16 dpp movs (quad_perm) = 8 doubles
      int lo = __double2loint(acc0), hi = __double2hiint(acc0);
      int r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile(
          "v_mov_b32_dpp %0, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %9 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n"
          "v_mov_b32_dpp %2, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
          "v_mov_b32_dpp %4, %8 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %9 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n"
          "v_mov_b32_dpp %6, %8 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %9 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(lo), "v"(hi));
      acc1 += (double)(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) * 1e-300;
      asm volatile(
          "v_mov_b32_dpp %0, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %9 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n"
          "v_mov_b32_dpp %2, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
          "v_mov_b32_dpp %4, %8 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %9 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n"
          "v_mov_b32_dpp %6, %8 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %9 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(lo), "v"(hi));
      acc2 += (double)(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) * 1e-300;
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * 256 + tid] = acc0 + acc1 + acc2 + acc3 + acc4 + acc5 + acc6 + acc7;
  if ((tid & 63) == 0) cycles[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int MODE>
int run(const char *name, int units_per_iter, double *dout, long long *dcyc, const double *gsrc, int ncu) {
  for (int wps : {1, 2, 4, 8}) {
    int grid = ncu * wps;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, dout, dcyc, gsrc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, dout, dcyc, gsrc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> cyc(grid * 4);
    CK(hipMemcpy(cyc.data(), dcyc, sizeof(long long) * grid * 4, hipMemcpyDeviceToHost));
    std::sort(cyc.begin(), cyc.end());
    double med = (double)cyc[cyc.size() / 2];
    double per_unit_wave = med / ((double)ITERS * units_per_iter);
    // units per cycle per SIMD, assuming waves-per-SIMD = wps all concurrently resident
    printf("%-34s waves/SIMD=%d  wave-cycles/unit=%7.2f  SIMD-cycles/unit=%6.2f  kernel=%.3f ms\n", name, wps,
           per_unit_wave, per_unit_wave / wps, ms);
  }
  return 0;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int ncu = prop.multiProcessorCount;
  printf("device %s  CUs=%d  clock=%d kHz  LDS/block=%zu  regs/block=%d\n", prop.name, ncu, prop.clockRate,
         prop.sharedMemPerBlock, prop.regsPerBlock);
  double *dout; long long *dcyc; double *gsrc;
  CK(hipMalloc(&dout, sizeof(double) * ncu * 8 * 256));
  CK(hipMalloc(&dcyc, sizeof(long long) * ncu * 8 * 4));
  CK(hipMalloc(&gsrc, sizeof(double) * 1024));
  std::vector<double> h(1024, 1.0);
  CK(hipMemcpy(gsrc, h.data(), sizeof(double) * 1024, hipMemcpyHostToDevice));
  run<0>("fma_f64 x16, 8 chains", 16, dout, dcyc, gsrc, ncu);
  run<1>("fma_f64 x16, 2 chains", 16, dout, dcyc, gsrc, ncu);
  run<2>("fma_f64 x16, 1 chain (latency)", 16, dout, dcyc, gsrc, ncu);
  run<3>("ds_read_b64 bcast x16 (doubles)", 16, dout, dcyc, gsrc, ncu);
  run<4>("ds_read_b128 bcast x8 (16 doubles)", 16, dout, dcyc, gsrc, ncu);
  run<5>("b128 bcast x8 + 16 fma (per fma)", 16, dout, dcyc, gsrc, ncu);
  run<6>("v_readlane_b32 x16 (8 doubles)", 8, dout, dcyc, gsrc, ncu);
  run<7>("2 readlane + fma(sgpr) x8 (per fma)", 8, dout, dcyc, gsrc, ncu);
  run<8>("s_load_x16 + 8 fma(sgpr) (per fma)", 8, dout, dcyc, gsrc, ncu);
  run<9>("v_mov_dpp quad bcast x16 (8 dbl)", 8, dout, dcyc, gsrc, ncu);
  return 0;
}
