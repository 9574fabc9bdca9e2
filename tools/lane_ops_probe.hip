// lane_ops_probe.hip -- what v_permlane32_swap / v_permlane16_swap and the DPP wave shifts do on
// gfx950, and what a "copy of w in every 16-lane row" costs through them against the LDS round trip
// the fused kernel's ADMM loop used.   hipcc --offload-arch=gfx950 -O3 -o lane_ops_probe lane_ops_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__global__ void probe(int *o) {
  const int lane = threadIdx.x;
  v2u r = __builtin_amdgcn_permlane32_swap((unsigned)lane, (unsigned)(100 + lane), false, false);
  v2u q = __builtin_amdgcn_permlane16_swap((unsigned)lane, (unsigned)(100 + lane), false, false);
  o[0 * 64 + lane] = r.x; o[1 * 64 + lane] = r.y; o[2 * 64 + lane] = q.x; o[3 * 64 + lane] = q.y;
  o[4 * 64 + lane] = __builtin_amdgcn_update_dpp(999, lane, 0x138, 0xf, 0xf, false);  // wave_shr:1
  o[5 * 64 + lane] = __builtin_amdgcn_update_dpp(999, lane, 0x130, 0xf, 0xf, false);  // wave_shl:1
}

__device__ __forceinline__ double swap_copy(double w, int q_sel, double (&Wq)[4]) {
  unsigned lo = __double2loint(w), hi = __double2hiint(w);
  v2u a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);   // (r0 r1 r0 r1), (r2 r3 r2 r3)
  v2u b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  v2u a0 = __builtin_amdgcn_permlane16_swap(a.x, a.x, false, false);  // (r0 x4), (r1 x4)
  v2u a1 = __builtin_amdgcn_permlane16_swap(a.y, a.y, false, false);  // (r2 x4), (r3 x4)
  v2u b0 = __builtin_amdgcn_permlane16_swap(b.x, b.x, false, false);
  v2u b1 = __builtin_amdgcn_permlane16_swap(b.y, b.y, false, false);
  Wq[0] = __hiloint2double(b0.x, a0.x); Wq[1] = __hiloint2double(b0.y, a0.y);
  Wq[2] = __hiloint2double(b1.x, a1.x); Wq[3] = __hiloint2double(b1.y, a1.y);
  return 0.0;
}

// mode 0: LDS round trip, mode 1: swaps.  A dependent chain: the next w is a sum of the copies.
__global__ void timing(double *out, long long *cycles, int iters, int mode) {
  __shared__ double buf[64];
  const int lane = threadIdx.x;
  double w = lane * 0.5 + 1.0;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    double Wq[4];
    if (mode == 0) {
      buf[lane] = w;
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      for (int q = 0; q < 4; ++q) Wq[q] = buf[16 * q + (lane & 15)];
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
      swap_copy(w, 0, Wq);
    }
    w = (Wq[0] + Wq[1]) * 0.25 + (Wq[2] + Wq[3]) * 0.125 + lane;
  }
  long long t1 = clock64();
  out[blockIdx.x * 64 + lane] = w;
  if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  int *o; hipMalloc(&o, 6 * 64 * sizeof(int));
  probe<<<1, 64>>>(o);
  int h[6 * 64]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[6] = {"permlane32_swap.vdst(lane,100+lane)", "permlane32_swap.src0", "permlane16_swap.vdst", "permlane16_swap.src0",
                          "wave_shr:1 (old=999)", "wave_shl:1 (old=999)"};
  for (int k = 0; k < 6; ++k) {
    printf("%s:\n", names[k]);
    for (int l = 0; l < 64; ++l) printf("%4d%s", h[k * 64 + l], (l % 16 == 15) ? "\n" : "");
  }
  double *out; long long *cyc; hipMalloc(&out, 1024 * 64 * 8); hipMalloc(&cyc, 1024 * 8);
  for (int mode = 0; mode < 2; ++mode) {
    double res[64];
    for (int rep = 0; rep < 2; ++rep) {
      timing<<<1024, 64>>>(out, cyc, 1000, mode);
      hipDeviceSynchronize();
    }
    long long hc[1024]; hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    hipMemcpy(res, out, sizeof(res), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += hc[i];
    printf("mode %d (%s): %.1f clock64 ticks per trip (1024 one-wave blocks), w[5]=%.6f w[37]=%.6f\n", mode,
           mode ? "permlane swaps" : "LDS round trip", s / 1024 / 1000, res[5], res[37]);
  }
  return 0;
}
