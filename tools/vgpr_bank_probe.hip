// vgpr_bank_probe.hip -- does the VGPR bank (register index mod 4) of the operands of v_fmac_f64_dpp
// matter?  16 FMAs per block on four accumulators, one wave per SIMD, every combination of bank pair
// ({0,1} = "A": index = 0 mod 4, {2,3} = "B": index = 2 mod 4) for accumulator / DPP operand / plain operand.
//   hipcc --offload-arch=gfx950 -O3 -o vgpr_bank_probe vgpr_bank_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define F(acc, w, m) "v_fmac_f64_dpp v[" #acc ":" #acc "+1], v[" #w ":" #w "+1], v[" #m ":" #m "+1] row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
// accumulators at A: 100,104,108,112  B: 102,106,110,114 ; w at A: 120 B: 122 ; m at A: 128.. step 4, B: 130.. step 4
#define BLOCK(a0, a1, a2, a3, w, m0)                                                                          \
  F(a0, w, m0) F(a1, w, m0 + 4) F(a2, w, m0 + 8) F(a3, w, m0 + 12) F(a0, w, m0 + 16) F(a1, w, m0 + 20)       \
  F(a2, w, m0 + 24) F(a3, w, m0 + 28) F(a0, w, m0 + 32) F(a1, w, m0 + 36) F(a2, w, m0 + 40) F(a3, w, m0 + 44) \
  F(a0, w, m0 + 48) F(a1, w, m0 + 52) F(a2, w, m0 + 56) F(a3, w, m0 + 60)

#define CLOBBERS "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115"

// the same 16 FMAs on 1, 2, 4, 8 and 16 accumulators (dependent-issue distance of the DPP form)
#define BLOCKN(w, m0, a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15)                            \
  F(a0, w, m0) F(a1, w, m0 + 4) F(a2, w, m0 + 8) F(a3, w, m0 + 12) F(a4, w, m0 + 16) F(a5, w, m0 + 20)               \
  F(a6, w, m0 + 24) F(a7, w, m0 + 28) F(a8, w, m0 + 32) F(a9, w, m0 + 36) F(a10, w, m0 + 40) F(a11, w, m0 + 44)     \
  F(a12, w, m0 + 48) F(a13, w, m0 + 52) F(a14, w, m0 + 56) F(a15, w, m0 + 60)
#define CLOBBERS2 CLOBBERS, "v116","v117","v118","v119","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75"

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, long long *cyc, int iters) {
  // initialise the registers the blocks read
  asm volatile("v_mov_b32 v120, 0\n\tv_mov_b32 v121, 0x3ff00000\n\tv_mov_b32 v122, 0\n\tv_mov_b32 v123, 0x3ff00000" ::: "v120", "v121", "v122", "v123");
  asm volatile(
      "s_mov_b32 s20, 0\n\t" ::: "s20");
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) asm volatile(BLOCK(100, 104, 108, 112, 120, 128) ::: CLOBBERS);   // acc A, w A, m A
    if constexpr (MODE == 1) asm volatile(BLOCK(100, 104, 108, 112, 120, 130) ::: CLOBBERS);   // acc A, w A, m B
    if constexpr (MODE == 2) asm volatile(BLOCK(100, 104, 108, 112, 122, 128) ::: CLOBBERS);   // acc A, w B, m A
    if constexpr (MODE == 3) asm volatile(BLOCK(100, 104, 108, 112, 122, 130) ::: CLOBBERS);   // acc A, w B, m B
    if constexpr (MODE == 4) asm volatile(BLOCK(102, 106, 110, 114, 120, 128) ::: CLOBBERS);   // acc B, w A, m A
    if constexpr (MODE == 5) asm volatile(BLOCK(102, 106, 110, 114, 120, 130) ::: CLOBBERS);   // acc B, w A, m B
    if constexpr (MODE == 6) asm volatile(BLOCK(102, 106, 110, 114, 122, 128) ::: CLOBBERS);   // acc B, w B, m A
    if constexpr (MODE == 7) asm volatile(BLOCK(102, 106, 110, 114, 122, 130) ::: CLOBBERS);   // acc B, w B, m B
    if constexpr (MODE == 10) asm volatile(BLOCKN(120, 130, 100, 100, 100, 100, 100, 100, 100, 100, 100, 100, 100, 100, 100, 100, 100, 100) ::: CLOBBERS2);
    if constexpr (MODE == 11) asm volatile(BLOCKN(120, 130, 100, 104, 100, 104, 100, 104, 100, 104, 100, 104, 100, 104, 100, 104, 100, 104) ::: CLOBBERS2);
    if constexpr (MODE == 12) asm volatile(BLOCKN(120, 130, 100, 104, 108, 112, 100, 104, 108, 112, 100, 104, 108, 112, 100, 104, 108, 112) ::: CLOBBERS2);
    if constexpr (MODE == 13) asm volatile(BLOCKN(120, 130, 100, 104, 108, 112, 116, 102, 100, 104, 108, 112, 116, 102, 100, 104, 108, 112) ::: CLOBBERS2);
    if constexpr (MODE == 14) asm volatile(BLOCKN(120, 130, 100, 104, 108, 112, 116, 102, 106, 110, 100, 104, 108, 112, 116, 102, 106, 110) ::: CLOBBERS2);
    if constexpr (MODE == 15) asm volatile(BLOCKN(120, 130, 100, 104, 108, 112, 116, 102, 106, 110, 114, 118, 60, 64, 68, 72, 62, 66) ::: CLOBBERS2);
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + threadIdx.x] = 0.0;
}

template <int MODE>
void run(const char *name, double *out, long long *cyc) {
  const int blocks = 1024, iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
  }
  static long long h[1024];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
  printf("%-28s %6.2f ticks per v_fmac_f64_dpp\n", name, s / blocks / iters / 16);
}
int main() {
  double *out; long long *cyc;
  (void)hipMalloc(&out, 1024 * 64 * 8); (void)hipMalloc(&cyc, 1024 * 8);
  run<0>("acc A, dpp A, plain A", out, cyc); run<1>("acc A, dpp A, plain B", out, cyc);
  run<2>("acc A, dpp B, plain A", out, cyc); run<3>("acc A, dpp B, plain B", out, cyc);
  run<4>("acc B, dpp A, plain A", out, cyc); run<5>("acc B, dpp A, plain B", out, cyc);
  run<6>("acc B, dpp B, plain A", out, cyc); run<7>("acc B, dpp B, plain B", out, cyc);
  run<10>("1 accumulator", out, cyc); run<11>("2 accumulators", out, cyc); run<12>("4 accumulators", out, cyc);
  run<13>("6 accumulators", out, cyc); run<14>("8 accumulators", out, cyc); run<15>("16 accumulators", out, cyc);
  return 0;
}
