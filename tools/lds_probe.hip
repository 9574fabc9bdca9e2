// prints the LDS limits the runtime reports (decides whether the generic kernel keeps the KKT matrix in LDS)
#include <hip/hip_runtime.h>
#include <cstdio>
int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
  int optin = -1;
  hipError_t e = hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, 0);
  printf("sharedMemPerBlock %zu  optin rc=%d value=%d  maxSharedMemoryPerMultiProcessor %zu\n",
         p.sharedMemPerBlock, (int)e, optin, p.maxSharedMemoryPerMultiProcessor);
  return 0;
}
