// mfma_f64_probe.hip -- cycles per v_mfma_f64_16x16x4_f64 at one wave per SIMD by where its operands live
// (accumulator in AGPRs or VGPRs, B operand in a VGPR or an AGPR) and by how many accumulators rotate.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define M_AA(acc, b) "v_mfma_f64_16x16x4_f64 a[" #acc ":" #acc "+7], v[10:11], " b ", a[" #acc ":" #acc "+7]\n\t"
#define M_VV(acc, b) "v_mfma_f64_16x16x4_f64 v[" #acc ":" #acc "+7], v[10:11], " b ", v[" #acc ":" #acc "+7]\n\t"

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, long long *cyc, int iters) {
  asm volatile("v_mov_b32 v10, 0\n\tv_mov_b32 v11, 0x3ff00000\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0x3ff00000\n\t"
               "v_accvgpr_write_b32 a100, v12\n\tv_accvgpr_write_b32 a101, v13" ::: "v10", "v11", "v12", "v13", "a100", "a101");
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0)   // one accumulator in AGPRs, B in a VGPR
      asm volatile(M_AA(0, "v[12:13]") M_AA(0, "v[12:13]") M_AA(0, "v[12:13]") M_AA(0, "v[12:13]") M_AA(0, "v[12:13]") M_AA(0, "v[12:13]") M_AA(0, "v[12:13]") M_AA(0, "v[12:13]")
                   ::: "a0","a1","a2","a3","a4","a5","a6","a7");
    if constexpr (MODE == 1)   // four accumulators in AGPRs
      asm volatile(M_AA(0, "v[12:13]") M_AA(8, "v[12:13]") M_AA(16, "v[12:13]") M_AA(24, "v[12:13]") M_AA(0, "v[12:13]") M_AA(8, "v[12:13]") M_AA(16, "v[12:13]") M_AA(24, "v[12:13]")
                   ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
    if constexpr (MODE == 2)   // one accumulator in VGPRs
      asm volatile(M_VV(20, "v[12:13]") M_VV(20, "v[12:13]") M_VV(20, "v[12:13]") M_VV(20, "v[12:13]") M_VV(20, "v[12:13]") M_VV(20, "v[12:13]") M_VV(20, "v[12:13]") M_VV(20, "v[12:13]")
                   ::: "v20","v21","v22","v23","v24","v25","v26","v27");
    if constexpr (MODE == 3)   // four accumulators in VGPRs
      asm volatile(M_VV(20, "v[12:13]") M_VV(28, "v[12:13]") M_VV(36, "v[12:13]") M_VV(44, "v[12:13]") M_VV(20, "v[12:13]") M_VV(28, "v[12:13]") M_VV(36, "v[12:13]") M_VV(44, "v[12:13]")
                   ::: "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51");
    if constexpr (MODE == 4)   // four accumulators in AGPRs, B in an AGPR
      asm volatile(M_AA(0, "a[100:101]") M_AA(8, "a[100:101]") M_AA(16, "a[100:101]") M_AA(24, "a[100:101]") M_AA(0, "a[100:101]") M_AA(8, "a[100:101]") M_AA(16, "a[100:101]") M_AA(24, "a[100:101]")
                   ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
    if constexpr (MODE == 5)   // two accumulators in AGPRs
      asm volatile(M_AA(0, "v[12:13]") M_AA(8, "v[12:13]") M_AA(0, "v[12:13]") M_AA(8, "v[12:13]") M_AA(0, "v[12:13]") M_AA(8, "v[12:13]") M_AA(0, "v[12:13]") M_AA(8, "v[12:13]")
                   ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15");
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + threadIdx.x] = 0.0;
}
template <int MODE>
void run(const char *name, double *out, long long *cyc) {
  const int blocks = 1024, iters = 1000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
  }
  static long long h[1024];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
  printf("%-44s %6.2f ticks per MFMA\n", name, s / blocks / iters / 8);
}
int main() {
  double *out; long long *cyc;
  (void)hipMalloc(&out, 1024 * 64 * 8); (void)hipMalloc(&cyc, 1024 * 8);
  run<0>("1 accumulator (AGPR), B in VGPR", out, cyc);
  run<5>("2 accumulators (AGPR), B in VGPR", out, cyc);
  run<1>("4 accumulators (AGPR), B in VGPR", out, cyc);
  run<4>("4 accumulators (AGPR), B in AGPR", out, cyc);
  run<2>("1 accumulator (VGPR), B in VGPR", out, cyc);
  run<3>("4 accumulators (VGPR), B in VGPR", out, cyc);
  return 0;
}
