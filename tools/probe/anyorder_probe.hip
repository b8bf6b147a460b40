// Does hipExtAnyOrderLaunch let two independent kernels of ONE stream overlap on gfx950?  (hip_ext.h says the flag is "not
// supported on AMD GFX9xx boards" for hipExtModuleLaunchKernel.)  Two single-workgroup kernels that each spin ~50 us:
// back to back they take ~100 us + boundaries; overlapped ~50 us.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin_kernel(long long cycles, int* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (threadIdx.x == 0) out[blockIdx.x] = 1;
}
int main() {
  int* d; hipMalloc(&d, 1024);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const long long cyc = 5000;   // wall_clock64 ticks at 100 MHz: 50 us
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a, s);
      for (int k = 0; k < 8; ++k) {
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, cyc, d);
        if (mode == 0) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, cyc, d + 1);
        else hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d + 1);
      }
      hipEventRecord(b, s); hipEventSynchronize(b);
      float ms = 0; hipEventElapsedTime(&ms, a, b);
      printf("%s: 8 pairs of 50 us kernels in %.1f us (%.1f us per pair)\n", mode ? "second kernel of a pair with hipExtAnyOrderLaunch" : "plain launches", ms * 1e3, ms * 1e3 / 8);
    }
  }
  return 0;
}
