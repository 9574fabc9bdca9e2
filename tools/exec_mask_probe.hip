// exec_mask_probe.hip -- does a fp64 VALU instruction get cheaper when only part of the wave is active?
// 32 v_fmac_f64_dpp (fresh operands, as in tile_ldl / the ADMM mat-vec) per loop trip with EXEC = all 64 lanes,
// the low 32, the low 16, and lanes 0..15 + 32..47; one wave per SIMD, 1024 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o exec_mask_probe exec_mask_probe.hip && ./exec_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define F(acc, m) "v_fmac_f64_dpp v[" #acc ":" #acc "+1], v[120:121], v[" #m ":" #m "+1] row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
#define BLOCK F(100, 128) F(102, 130) F(104, 132) F(106, 134) F(100, 136) F(102, 138) F(104, 140) F(106, 142) \
              F(100, 144) F(102, 146) F(104, 148) F(106, 150) F(100, 152) F(102, 154) F(104, 156) F(106, 158) \
              F(100, 160) F(102, 162) F(104, 164) F(106, 166) F(100, 168) F(102, 170) F(104, 172) F(106, 174) \
              F(100, 176) F(102, 178) F(104, 180) F(106, 182) F(100, 184) F(102, 186) F(104, 188) F(106, 190)

#define G(acc, m) F(acc, m)
#define BLOCK1 G(100, 128) G(100, 130) G(100, 132) G(100, 134) G(100, 136) G(100, 138) G(100, 140) G(100, 142) \
               G(100, 144) G(100, 146) G(100, 148) G(100, 150) G(100, 152) G(100, 154) G(100, 156) G(100, 158) \
               G(100, 160) G(100, 162) G(100, 164) G(100, 166) G(100, 168) G(100, 170) G(100, 172) G(100, 174) \
               G(100, 176) G(100, 178) G(100, 180) G(100, 182) G(100, 184) G(100, 186) G(100, 188) G(100, 190)
#define BLOCK2 G(100, 128) G(102, 130) G(100, 132) G(102, 134) G(100, 136) G(102, 138) G(100, 140) G(102, 142) \
               G(100, 144) G(102, 146) G(100, 148) G(102, 150) G(100, 152) G(102, 154) G(100, 156) G(102, 158) \
               G(100, 160) G(102, 162) G(100, 164) G(102, 166) G(100, 168) G(102, 170) G(100, 172) G(102, 174) \
               G(100, 176) G(102, 178) G(100, 180) G(102, 182) G(100, 184) G(102, 186) G(100, 188) G(102, 190)
// all matrix operands in one VGPR bank (register index = 0 mod 4), four accumulators: round 2's probe layout
#define BLOCK4S G(100, 128) G(104, 132) G(108, 136) G(112, 140) G(100, 144) G(104, 148) G(108, 152) G(112, 156) \
                G(100, 160) G(104, 164) G(108, 168) G(112, 172) G(100, 176) G(104, 180) G(108, 184) G(112, 188) \
                G(100, 192) G(104, 196) G(108, 200) G(112, 204) G(100, 208) G(104, 212) G(108, 216) G(112, 220) \
                G(100, 224) G(104, 228) G(108, 232) G(112, 236) G(100, 240) G(104, 244) G(108, 248) G(112, 252)

template <int MODE>
__global__ __launch_bounds__(64) void k(long long *cyc, int iters, unsigned long long mask) {
  asm volatile("v_mov_b32 v120, 0\n\tv_mov_b32 v121, 0x3ff00000" ::: "v120", "v121");
  const long long t0 = clock64();
  asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, %0" :: "s"(mask) : "s20", "s21");
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) asm volatile(BLOCK ::: "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
    else if constexpr (MODE == 2) asm volatile(BLOCK1 ::: "v100", "v101");
    else if constexpr (MODE == 3) asm volatile(BLOCK2 ::: "v100", "v101", "v102", "v103");
    else if constexpr (MODE == 4) asm volatile(BLOCK4S ::: "v100", "v101", "v104", "v105", "v108", "v109", "v112", "v113");
    else asm volatile("v_fma_f64 v[100:101], v[128:129], v[130:131], v[100:101]\n\tv_fma_f64 v[102:103], v[132:133], v[134:135], v[102:103]\n\t"
                      "v_fma_f64 v[104:105], v[136:137], v[138:139], v[104:105]\n\tv_fma_f64 v[106:107], v[140:141], v[142:143], v[106:107]\n\t"
                      "v_fma_f64 v[100:101], v[144:145], v[146:147], v[100:101]\n\tv_fma_f64 v[102:103], v[148:149], v[150:151], v[102:103]\n\t"
                      "v_fma_f64 v[104:105], v[152:153], v[154:155], v[104:105]\n\tv_fma_f64 v[106:107], v[156:157], v[158:159], v[106:107]\n\t"
                      ::: "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
  }
  asm volatile("s_mov_b64 exec, s[20:21]" ::: "memory");
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, long long *cyc, unsigned long long mask, int per_trip) {
  const int blocks = 1024, iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, cyc, iters, mask);
    (void)hipDeviceSynchronize();
  }
  static long long h[1024];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < blocks; ++i) s += h[i];
  printf("%-44s %6.2f ticks per instruction\n", name, s / blocks / iters / per_trip);
}
int main() {
  long long *cyc;
  (void)hipMalloc(&cyc, 1024 * 8);
  run<0>("dpp fmac, EXEC = 64 lanes", cyc, ~0ull, 32);
  run<0>("dpp fmac, EXEC = lanes 0..31", cyc, 0xffffffffull, 32);
  run<0>("dpp fmac, EXEC = lanes 0..15", cyc, 0xffffull, 32);
  run<0>("dpp fmac, EXEC = lanes 0..15 and 32..47", cyc, 0x0000ffff0000ffffull, 32);
  run<2>("dpp fmac, ONE accumulator (dependent chain)", cyc, ~0ull, 32);
  run<3>("dpp fmac, two accumulators", cyc, ~0ull, 32);
  run<4>("dpp fmac, 4 acc, operands all in one bank", cyc, ~0ull, 32);
  run<1>("v_fma_f64, EXEC = 64 lanes", cyc, ~0ull, 8);
  run<1>("v_fma_f64, EXEC = lanes 0..31", cyc, 0xffffffffull, 8);
  run<1>("v_fma_f64, EXEC = lanes 0..15", cyc, 0xffffull, 8);
  return 0;
}
