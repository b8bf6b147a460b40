// Which ingredient of the row-GEMM main loop costs MFMA issue time on gfx950?  A stream shaped like one reduction chunk
// of spg_rowgemm_kernel<128,128,2,2> (64 x v_mfma_f32_32x32x2_f32 on 4 accumulators, 16 k-steps) with the other
// instruction classes of the real loop switched on one at a time:
//   bit 0  16 ds_read_b128 fragment reads per chunk (register double-buffered, one group ahead)
//   bit 1  8 ds_write_b128 staging writes per chunk (slots 3..10, real out-major addresses)
//   bit 2  8 global_load_dwordx4 per chunk (slots 0..2), L2-resident source
//   bit 3  32 independent VALU (v_fma + v_max on the loaded data) per chunk, next to the LDS writes
//   bit 4  s_barrier per chunk
//   bit 5  9 x 64-bit address adds per chunk (v_lshl_add_u64), as the compiler emits for the loads
//   bit 6  the global loads come from a 1 GiB buffer (HBM) instead of L2
// build: hipcc --offload-arch=gfx950 -O3 mainloop_probe.hip -o mainloop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters, const float* in, size_t span) {
  extern __shared__ f32x4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, wi = wave >> 1, wj = wave & 1;
  for (int i = tid; i < 2 * (8 * 129 + 8 * 129); i += 256) lds[i] = f32x4{(float)i, 1.f, 2.f, 3.f};
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  f32x4 raw[8];
  for (int i = 0; i < 8; ++i) raw[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  const f32x4 sc = {1.0001f, 0.9999f, 1.0002f, 0.9998f}, sh = {0.1f, 0.2f, 0.3f, 0.4f};
  const f32x4* As = lds;
  const f32x4* Bs = lds + 8 * 129;
  const int rowA = wi * 64 + r, rowB = wj * 64 + r;
  unsigned voff[8];
  for (int i = 0; i < 8; ++i) voff[i] = (((tid >> 3) + 32u * (i & 3)) * 128u + 4u * (tid & 7)) * 4u;
  const char* base = reinterpret_cast<const char*>(in) + (size_t)blockIdx.x * 65536 % span;
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    const f32x4* Ac = As + buf * 2064;
    const f32x4* Bc = Bs + buf * 2064;
    f32x4* An = const_cast<f32x4*>(As) + (buf ^ 1) * 2064;
    const char* cb = base + ((size_t)it * 16384) % (span > 65536 ? span - 65536 : 1);
    f32x4 a[2][2], b[2][2];
    if (MODE & 1) {
      for (int i = 0; i < 2; ++i) a[0][i] = Ac[h * 129 + rowA + 32 * i];
      for (int j = 0; j < 2; ++j) b[0][j] = Bc[h * 129 + rowB + 32 * j];
    } else {
      for (int i = 0; i < 2; ++i) { a[0][i] = raw[i]; a[1][i] = raw[i + 2]; }
      for (int j = 0; j < 2; ++j) { b[0][j] = raw[4 + j]; b[1][j] = raw[6 + j]; }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
      if ((MODE & 1) && g + 1 < 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[nxt][i] = Ac[(2 * (g + 1) + h) * 129 + rowA + 32 * i];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[nxt][j] = Bc[(2 * (g + 1) + h) * 129 + rowB + 32 * j];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][s], b[cur][j][s], acc[i][j], 0, 0, 0);
        const int slot = 4 * g + s;
        if (MODE & 4) {
          if (slot < 3) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              const int k = 3 * slot + i;
              if (k < 8) {
                if (MODE & 32) raw[k] = *reinterpret_cast<const f32x4*>(cb + (size_t)voff[k]);            // 64-bit vector address
                else raw[k] = *reinterpret_cast<const f32x4*>(cb + voff[k]);
              }
            }
          }
        }
        if (slot >= 3 && slot < 11) {
          const int k = slot - 3;
          f32x4 v = raw[k];
          if (MODE & 8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);
          }
          if (MODE & 2) An[(tid & 7) * 129 + (tid >> 3) + 32 * (k & 3) + (k >> 2) * 1032] = v;
          else if (MODE & 8) raw[k] = v;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE & 16) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += raw[i][0] + raw[i][3];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, int blocks, float* out, const float* in, size_t span) {
  const int iters = 400;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 66048, 0, out, iters, in, span);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 66048, 0, out, iters, in, span);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)blocks * 4 * iters * 64;
  const double tf = mfma * 4096 / (ms * 1e-3) / 1e12;
  printf("%-62s blocks=%4d  %.3f ms  %6.1f TF (%.0f%% of 157.3)\n", name, blocks, ms, tf, tf / 157.3 * 100);
  fflush(stdout);
}

int main() {
  float *out, *in;
  const size_t big = (size_t)1 << 30;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, big + (1 << 20)); hipMemset(in, 0, big + (1 << 20));
  const size_t l2 = (size_t)1 << 22;
  for (int blocks : {512}) {
    run<0>("pure MFMA (operands in registers)", blocks, out, in, l2);
    run<1>("+ 16 ds_read_b128", blocks, out, in, l2);
    run<1 | 2>("+ 16 ds_read_b128 + 8 ds_write_b128", blocks, out, in, l2);
    run<1 | 4>("+ 16 ds_read_b128 + 8 global loads (L2)", blocks, out, in, l2);
    run<1 | 8>("+ 16 ds_read_b128 + 32 VALU", blocks, out, in, l2);
    run<1 | 2 | 4>("+ reads + writes + loads", blocks, out, in, l2);
    run<1 | 2 | 4 | 8>("+ reads + writes + loads + VALU", blocks, out, in, l2);
    run<1 | 2 | 4 | 8 | 16>("+ reads + writes + loads + VALU + barrier", blocks, out, in, l2);
    run<1 | 2 | 4 | 8 | 16 | 32>("+ ... + 64-bit vector addresses", blocks, out, in, l2);
    run<1 | 2 | 4 | 8 | 16 | 64>("+ reads + writes + loads (HBM) + VALU + barrier", blocks, out, in, big);
    run<2>("8 ds_write_b128 only", blocks, out, in, l2);
    run<4>("8 global loads only", blocks, out, in, l2);
    run<8>("32 VALU only", blocks, out, in, l2);
    run<16>("barrier only", blocks, out, in, l2);
  }
  for (int blocks : {256, 1024}) {
    run<0>("pure MFMA", blocks, out, in, l2);
    run<1 | 2 | 4 | 8 | 16>("+ reads + writes + loads + VALU + barrier", blocks, out, in, l2);
  }
  return 0;
}
