// Platform check: a long-running (tens of ms) one-wave-per-block kernel that does nothing but
// deterministic per-lane arithmetic through registers, scratch (a dynamically indexed private
// array), LDS and its own slice of global memory -- no communication between lanes at all -- and
// compares every lane's result with the host's.  Written to tell a wave save/restore (pre-emption)
// problem of a shared GPU from a bug in the solver's global-workspace path.
//   hipcc --offload-arch=gfx950 -O3 tools/longrun_stress.hip -o /tmp/longrun_stress && /tmp/longrun_stress
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__host__ __device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 29; return x; }

__global__ __launch_bounds__(64, 1) void stress(uint64_t *out, uint64_t *work, int iters) {
  __shared__ uint64_t lds[64 * 8];
  const int lane = threadIdx.x;
  for (int i = 0; i < 8; ++i) lds[lane * 8 + i] = 0;
  uint64_t priv[64];                                   // dynamically indexed -> scratch
  for (int i = 0; i < 64; ++i) priv[i] = mix(blockIdx.x * 64 + lane + i);
  uint64_t *mine = work + ((size_t)blockIdx.x * 64 + lane) * 64;
  for (int i = 0; i < 64; ++i) mine[i] = priv[i] + 1;
  uint64_t acc = lane;
  for (int it = 0; it < iters; ++it) {
    const int k = (int)(acc & 63);
    acc = mix(acc + priv[k]);
    priv[(k + 7) & 63] ^= acc;
    lds[lane * 8 + (it & 7)] = acc;
    acc += lds[lane * 8 + ((it + 3) & 7)] & 0xffff;
    mine[k] += acc;                                    // global read-modify-write, own slice
    acc ^= mine[(k + 5) & 63];
  }
  out[blockIdx.x * 64 + lane] = acc;
}

static uint64_t host_ref(int block, int lane, int iters) {
  uint64_t priv[64], mine[64], lds[8] = {0};
  for (int i = 0; i < 64; ++i) { priv[i] = mix(block * 64 + lane + i); mine[i] = priv[i] + 1; }
  uint64_t acc = lane;
  for (int it = 0; it < iters; ++it) {
    const int k = (int)(acc & 63);
    acc = mix(acc + priv[k]);
    priv[(k + 7) & 63] ^= acc;
    lds[it & 7] = acc;
    acc += lds[(it + 3) & 7] & 0xffff;
    mine[k] += acc;
    acc ^= mine[(k + 5) & 63];
  }
  return acc;
}

int main(int argc, char **argv) {
  const int blocks = 8, iters = argc > 1 ? atoi(argv[1]) : 40000, reps = argc > 2 ? atoi(argv[2]) : 20;
  uint64_t *out, *work;
  if (hipMalloc(&out, blocks * 64 * 8) != hipSuccess || hipMalloc(&work, (size_t)blocks * 64 * 64 * 8) != hipSuccess) return 2;
  std::vector<uint64_t> ref(blocks * 64), got(blocks * 64);
  for (int b = 0; b < blocks; ++b) for (int l = 0; l < 64; ++l) ref[b * 64 + l] = host_ref(b, l, iters);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int bad_runs = 0; float ms = 0;
  for (int r = 0; r < reps; ++r) {
    (void)hipEventRecord(e0); hipLaunchKernelGGL(stress, dim3(blocks), dim3(64), 0, 0, out, work, iters); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(got.data(), out, got.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0; for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
    if (bad) { ++bad_runs; printf("run %d: %d of %zu lanes differ\n", r, bad, got.size()); }
  }
  printf("longrun_stress: %d iterations, %.1f ms per launch, %d of %d launches wrong\n", iters, ms, bad_runs, reps);
  return bad_runs ? 1 : 0;
}
