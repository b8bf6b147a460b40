// Do v_mfma_f32_32x32x2_f32 (fp32-input MFMA, 64 FLOP/clk/SIMD = the fp32 VECTOR rate) and plain fp32 VALU work of ANOTHER wave
// on the same SIMD run next to each other, or do they take turns on the same lanes?  (Round 6: every attempt to hide the row-GEMM
// epilogues behind the co-resident workgroup's MFMAs had moved nothing; T = T_mfma + T_other fitted every variant.)
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run role A, waves 4-7 (their SIMD partners) role B.
//   A: a stream of NA x 64 independent-accumulator MFMAs (fp32 32x32x2, or bf16 32x32x16 for contrast)
//   B: idle | a stream of NB x 64 independent v_fma_f32 | ds_read_b128 stream | global_load stream
// T(A alone), T(B alone), T(A and B): separate pipes -> max, shared lanes -> sum.
// Second part: ONE wave per SIMD (or two doing the same) with k plain VALU operations between consecutive MFMAs: price per VALU.
// build: hipcc --offload-arch=gfx950 -O3 coissue_probe.hip -o coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int AKIND, int BKIND>     // AKIND 0 none, 1 fp32 MFMA, 2 bf16 MFMA; BKIND 0 none, 1 VALU fma, 2 LDS reads, 3 global loads, 4 fp32 MFMA
__global__ __launch_bounds__(512, 1) void roles(float* out, int na, int nb, const float* in) {
  __shared__ f32x4 lds[4096];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 4096; i += 512) lds[i] = f32x4{(float)i, 1.f, 2.f, 3.f};
  __syncthreads();
  float res = 0.f;
  auto mfma32 = [&](int n) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    const float a = in[lane], b = in[lane + 64];
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) res += acc[i][q];
  };
  auto mfma16 = [&](int n) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)in[lane + e]; b[e] = (__bf16)in[lane + 64 + e]; }
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) res += acc[i][q];
  };
  if (wave < 4) {
    if (AKIND == 1) mfma32(na);
    if (AKIND == 2) mfma16(na);
  } else {
    if (BKIND == 1) {
      float x[8];
      for (int i = 0; i < 8; ++i) x[i] = in[lane + i];
      const float m = in[200 + lane], c = in[300 + lane];
      for (int it = 0; it < nb; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));   // (plain C gets packed into v_pk_fma_f32)
      }
      for (int i = 0; i < 8; ++i) res += x[i];
    } else if (BKIND == 2) {
      f32x4 s = {0, 0, 0, 0};
      for (int it = 0; it < nb; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { const f32x4 v = lds[(it * 16 + u) * 64 % 4032 + lane]; s += v; }
      }
      res += s[0] + s[1] + s[2] + s[3];
    } else if (BKIND == 3) {
      f32x4 s = {0, 0, 0, 0};
      for (int it = 0; it < nb; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s += *reinterpret_cast<const f32x4*>(in + (((size_t)(it * 8 + u) * 4096 + blockIdx.x * 512 + tid) * 4) % (1 << 20));
      }
      res += s[0] + s[1] + s[2] + s[3];
    } else if (BKIND == 4) {
      mfma32(nb);
    } else if (BKIND >= 5) {      // 5 / 6 / 7: ONE / TWO / FOUR dependent chains of v_fma_f32 (the statistics of a GEMM epilogue are such chains)
      constexpr int NC = BKIND == 5 ? 1 : (BKIND == 6 ? 2 : 4);
      float x[NC];
      for (int i = 0; i < NC; ++i) x[i] = in[lane + i];
      const float m = in[200 + lane], c = in[300 + lane];
      for (int it = 0; it < nb; ++it) {
#pragma unroll
        for (int u = 0; u < 64 / NC; ++u)
#pragma unroll
          for (int i = 0; i < NC; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));
      }
      for (int i = 0; i < NC; ++i) res += x[i];
    }
  }
  out[blockIdx.x * 512 + tid] = res;
}

// W waves per SIMD all running: per MFMA, KV plain VALU operations (independent chains) placed behind it
template <int KV, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void mixed(float* out, int n, const float* in) {
  const int tid = threadIdx.x, lane = tid & 63;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  const float a = in[lane], b = in[lane + 64];
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = in[lane + i];
  const float m = in[200 + lane], c = in[300 + lane];
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // one volatile statement per MFMA and per VALU operation: volatile asm statements keep their program order
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
        for (int v = 0; v < KV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v]) : "v"(m), "v"(c));
      }
  }
  float res = 0.f;
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) res += acc[i][q];
  for (int i = 0; i < 16; ++i) res += x[i];
  out[blockIdx.x * THREADS + tid] = res;
}

// the same with OTHER VALU operations behind each MFMA: OP 1 = v_pk_fma_f32 (two fp32 FMAs per lane), 2 = v_mov_b64, 3 = v_mov_b32,
// 4 = v_max_f32, 5 = v_cndmask_b32 (round 6: is a packed operation charged like one plain operation?)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KV, int THREADS, int OP>
__global__ __launch_bounds__(THREADS, 1) void mixed_op(float* out, int n, const float* in) {
  const int tid = threadIdx.x, lane = tid & 63;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  const float a = in[lane], b = in[lane + 64];
  f32x2 x[16];
  for (int i = 0; i < 16; ++i) { x[i][0] = in[lane + i]; x[i][1] = in[lane + 32 + i]; }
  const f32x2 m = {in[200 + lane], in[201 + lane]}, c = {in[300 + lane], in[301 + lane]};
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
        for (int v = 0; v < KV; ++v) {
          if constexpr (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[v]) : "v"(m), "v"(c));
          else if constexpr (OP == 2) asm volatile("v_mov_b64 %0, %1" : "+v"(x[v]) : "v"(m));
          else if constexpr (OP == 3) asm volatile("v_mov_b32 %0, %1" : "+v"(x[v][0]) : "v"(m[0]));
          else if constexpr (OP == 4) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[v][0]) : "v"(m[0]));
          else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[v][0]) : "v"(m[0]));
        }
      }
  }
  float res = 0.f;
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) res += acc[i][q];
  for (int i = 0; i < 16; ++i) res += x[i][0] + x[i][1];
  out[blockIdx.x * THREADS + tid] = res;
}

template <class F>
static float timeit(F&& launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float *out, *in;
  hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&in, (1 << 22) + 4096);
  hipMemset(in, 0, (1 << 22) + 4096);
  const int blocks = 256, na = 2000, nbv = 16000, nbl = 6000, nbg = 1500;
#define ROLES(A, B, NA, NB, name) { const float ms = timeit([&] { hipLaunchKernelGGL((roles<A, B>), dim3(blocks), dim3(512), 0, 0, out, NA, NB, in); }); printf("%-58s %8.3f ms\n", name, ms); fflush(stdout); }
  printf("== roles: waves 0-3 = A, waves 4-7 (same SIMDs) = B; 256 workgroups (one per CU) ==\n");
  ROLES(1, 0, na, 0, "A fp32 MFMA alone (2000 x 64 MFMAs per wave)");
  ROLES(0, 1, 0, nbv, "B VALU fma alone (16000 x 64 v_fma per wave)");
  ROLES(1, 1, na, nbv, "A fp32 MFMA + B VALU fma");
  ROLES(0, 2, 0, nbl, "B ds_read_b128 alone");
  ROLES(1, 2, na, nbl, "A fp32 MFMA + B ds_read_b128");
  ROLES(0, 3, 0, nbg, "B global loads alone");
  ROLES(1, 3, na, nbg, "A fp32 MFMA + B global loads");
  ROLES(1, 4, na, na, "A fp32 MFMA + B fp32 MFMA (expected: sum)");
  ROLES(0, 5, 0, nbv / 4, "B ONE dependent chain of v_fma alone (4000 x 64)");
  ROLES(1, 5, na, nbv / 4, "A fp32 MFMA + B one dependent chain");
  ROLES(0, 6, 0, nbv / 4, "B TWO dependent chains alone (4000 x 64)");
  ROLES(1, 6, na, nbv / 4, "A fp32 MFMA + B two dependent chains");
  ROLES(0, 7, 0, nbv / 4, "B FOUR dependent chains alone (4000 x 64)");
  ROLES(1, 7, na, nbv / 4, "A fp32 MFMA + B four dependent chains");
  ROLES(2, 0, 4 * na, 0, "A bf16 MFMA alone (8000 x 64 MFMAs per wave)");
  ROLES(2, 1, 4 * na, nbv, "A bf16 MFMA + B VALU fma");
  printf("== mixed: every wave runs MFMA + k VALU behind each MFMA; 1000 x 64 MFMAs per wave ==\n");
#define MIXED(KV, T, name) { const float ms = timeit([&] { hipLaunchKernelGGL((mixed<KV, T>), dim3(blocks), dim3(T), 0, 0, out, 1000, in); }); \
    const double cyc = ms * 1e-3 * 2.4e9 / (1000.0 * 64 * (T / 256)); printf("%-58s %8.3f ms  %6.1f cycles per MFMA per SIMD (at 2.4 GHz)\n", name, ms, cyc); fflush(stdout); }
  MIXED(0, 256, "1 wave/SIMD, k = 0");
  MIXED(2, 256, "1 wave/SIMD, k = 2");
  MIXED(4, 256, "1 wave/SIMD, k = 4");
  MIXED(8, 256, "1 wave/SIMD, k = 8");
  MIXED(12, 256, "1 wave/SIMD, k = 12");
  MIXED(16, 256, "1 wave/SIMD, k = 16");
  MIXED(0, 512, "2 waves/SIMD, k = 0");
  MIXED(2, 512, "2 waves/SIMD, k = 2");
  MIXED(4, 512, "2 waves/SIMD, k = 4");
  MIXED(8, 512, "2 waves/SIMD, k = 8");
  MIXED(16, 512, "2 waves/SIMD, k = 16");
  printf("== mixed_op: 8 operations of another kind behind each MFMA, 1 wave/SIMD ==\n");
#define MIXOP(OP, name) { const float ms = timeit([&] { hipLaunchKernelGGL((mixed_op<8, 256, OP>), dim3(blocks), dim3(256), 0, 0, out, 1000, in); }); \
    const double cyc = ms * 1e-3 * 2.4e9 / (1000.0 * 64); printf("%-58s %8.3f ms  %6.1f cycles per MFMA per SIMD (at 2.4 GHz)\n", name, ms, cyc); fflush(stdout); }
  MIXOP(1, "k = 8 v_pk_fma_f32");
  MIXOP(2, "k = 8 v_mov_b64");
  MIXOP(3, "k = 8 v_mov_b32");
  MIXOP(4, "k = 8 v_max_f32");
  MIXOP(5, "k = 8 v_cndmask_b32");
  return 0;
}
