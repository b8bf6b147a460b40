// How much does a grid-wide barrier cost on this chip, next to a kernel boundary?  (a persistent
// R-iteration ECC kernel would replace 10 + 11 launches by as many grid barriers.)  250 workgroups x 256 threads (the ECC
// step kernels' grid at one scene; all co-resident: one per CU), K barriers in a loop -- agent-scope release, one atomic,
// bounded spin on a relaxed load, acquire (cdna guide G16) -- against K empty launches of the same grid.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier_probe grid_barrier_probe.hip && ./grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void barrier_loop(int* counter, int* err, int nwg, int iters, float* sink) {
  __shared__ int ok;
  float v = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    v = v * 1.0001f + 1.f;                                   // a token of work between barriers
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int target = (it + 1) * nwg;
      int spins = 0, good = 1;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 2000000) { good = 0; atomicExch(err, 1); break; }      // never hang the box
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      ok = good;
    }
    __syncthreads();
    if (!ok) break;
  }
  if (v == 123.456f) sink[0] = v;
}

__global__ void empty_kernel(float* sink) {
  float v = threadIdx.x * 1.0001f + 1.f;
  if (v == 123.456f) sink[0] = v;
}

int main() {
  const int nwg = 250, iters = 200;
  int *counter, *err;
  float* sink;
  hipMalloc(&counter, 4); hipMalloc(&err, 4); hipMalloc(&sink, 4);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(counter, 0, 4); hipMemset(err, 0, 4);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(barrier_loop, dim3(nwg), dim3(256), 0, 0, counter, err, nwg, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    int e = 0; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
    printf("grid barrier (250 WGs): %.2f us per barrier%s\n", ms * 1e3 / iters, e ? "  [SPIN LIMIT HIT]" : "");
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(empty_kernel, dim3(nwg), dim3(256), 0, 0, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("kernel boundary (250 WGs, empty kernels back to back): %.2f us per launch\n", ms * 1e3 / iters);
  }
  return 0;
}
