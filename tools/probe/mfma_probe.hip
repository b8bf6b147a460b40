// Ground-truth probe: what limits v_mfma_f32_32x32x2_f32 streams on gfx950?
// Variants: pure MFMA; + ds_read_b128 fragment traffic; + VALU filler; with 1 or 2 waves per SIMD and 1..2 blocks/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters, const float* in) {
  extern __shared__ f32x4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 256) lds[i] = f32x4{(float)i, 1.f, 2.f, 3.f};
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  f32x4 a[2] = {lds[lane], lds[lane + 64]}, b[2] = {lds[lane + 128], lds[lane + 192]};
  float filler = in[tid];
  f32x4 g = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
      if (MODE & 1) {   // fragment reads from LDS like the GEMM main loop
        a[0] = lds[(it * 4 + grp) % 8 * 129 + lane]; a[1] = lds[(it * 4 + grp) % 8 * 129 + lane + 32];
        b[0] = lds[1100 + (it * 4 + grp) % 8 * 129 + lane]; b[1] = lds[1100 + (it * 4 + grp) % 8 * 129 + lane + 32];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        if (MODE & 2) {   // VALU filler in the MFMA shadow (16 dependent-free FMAs)
#pragma unroll
          for (int v = 0; v < 16; ++v) filler = fmaf(filler, 1.0001f, 0.5f);
        }
        if (MODE & 4) {   // global loads in the shadow
          if (s == 0) g += *reinterpret_cast<const f32x4*>(in + ((size_t)(it * 4 + grp) * 256 + tid) * 4 % (1 << 22));
        }
        if (MODE & 8) {   // LDS writes in the shadow
          if (s == 1) lds[2300 + tid] = g;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE & 16) __syncthreads();
  }
  float s = filler + g[0] + g[1];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, int blocks, float* out, const float* in) {
  const int iters = 200;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 65536, 0, out, iters, in);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 65536, 0, out, iters, in);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)blocks * 4 * iters * 64;   // per wave: 64 MFMAs per iteration
  const double tf = mfma * 4096 / (ms * 1e-3) / 1e12;
  printf("%-44s blocks=%4d  %.3f ms  %.1f TF (%.0f%% of 157.3)\n", name, blocks, ms, tf, tf / 157.3 * 100);
}

int main() {
  float *out, *in;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, (1 << 22) * 4 + 65536); hipMemset(in, 0, (1 << 22) * 4 + 65536);
  for (int blocks : {256, 512, 2048}) {
    run<0>("pure MFMA", blocks, out, in);
    run<1>("+ LDS fragment reads", blocks, out, in);
    run<3>("+ LDS reads + VALU filler", blocks, out, in);
    run<5>("+ LDS reads + global loads", blocks, out, in);
    run<13>("+ LDS reads + global loads + LDS writes", blocks, out, in);
    run<29>("+ ... + barrier per 64 MFMAs", blocks, out, in);
    run<31>("+ ... + barrier + VALU filler", blocks, out, in);
  }
  return 0;
}
