// VERDICT r4 item 3 / r5 item 5a as a TIMED experiment: a PointNet FC head (learning/pointnet.py:39-49: Linear 128 -> 128, BatchNorm,
// ReLU, Linear 128 -> 64, BatchNorm, ReLU, Linear 64 -> 4 over ~1000 superpoints) is a chain of three few-row links whose train-mode
// BatchNorm needs the statistics of ALL rows between two links.  Does ONE launch with in-kernel hand-overs beat three dependent launches?
//   A  three launches (what the library does: 32 workgroups of 32 rows each; a link = fold the previous link's per-workgroup statistics
//      partials -> scale / shift, stage 32 rows, multiply, write the raw output + its statistics partial)
//   B  one launch of the same 32 workgroups: behind each link a device-scope hand-over -- release the partial, arrive on a counter, spin
//      until all 32 have arrived, acquire -- then the next link (MI355X_MICROARCH.md price list: fan-in 3.2-4.5 us, hand-off with a flag
//      2-6 us under load)
//   C  B with the workgroups' partials published as tagged 8-byte granules and swept by every workgroup (no counter, no fence)
// The link bodies are the same code in all variants (fp32 FMAs from LDS; the real links use MFMA tiles of the same size -- the body's
// time cancels in the comparison), sized like the real ones.  Prints the time per chain, median of 200.
// build: hipcc --offload-arch=gfx950 -O3 head_chain_probe.hip -o head_chain_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define ROWS 32
#define NWG 32
struct Chain {
  const float* x0;        // [M][128] input of the first link (already normalised)
  const float* W[3];      // [N][K]
  float* y[3];            // raw outputs [M][N]
  float* part[3];         // [NWG][N][2] statistics partials of y[l] (sum, sum of squares)
  unsigned long long* gran[3];   // [NWG][N] tagged granules {tag, float sum | float sumsq packed as two halves -> two arrays}
  unsigned long long* gran2[3];
  unsigned* counter;      // arrival counters [3]
  unsigned tag;           // epoch of this run (granule variant)
  int M;
};
__device__ constexpr int KK[3] = {128, 128, 64}, NN[3] = {128, 64, 4};

// one link for this workgroup's 32 rows: x (LDS, [32][K], already BatchNorm + ReLU) -> y rows + statistics partial
template <int K, int N>
__device__ void link_body(const float* xs, const float* __restrict__ W, float* __restrict__ y, float* psum, float* psq, int m0, int M) {
  // W is [K][N] (coalesced along the outputs); a thread owns output column n and every (256 / N)-th row
  const int tid = threadIdx.x, n = tid % N, r0 = tid / N;
  constexpr int RSTEP = 256 / N > 0 ? 256 / N : 1, NR = (ROWS + RSTEP - 1) / RSTEP;
  if (tid >= N * RSTEP) return;
  float acc[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) acc[i] = 0.f;
  for (int k = 0; k < K; ++k) {
    const float w = W[k * N + n];
#pragma unroll
    for (int i = 0; i < NR; ++i) { const int row = r0 + RSTEP * i; if (row < ROWS) acc[i] = fmaf(xs[row * (K + 1) + k], w, acc[i]); }
  }
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int row = r0 + RSTEP * i;
    if (row < ROWS && m0 + row < M) { y[(long)(m0 + row) * N + n] = acc[i]; s += acc[i]; q += acc[i] * acc[i]; }
  }
  atomicAdd(&psum[n], s);
  atomicAdd(&psq[n], q);
}

// stage this workgroup's rows of the previous output with its BatchNorm (from the partials of ALL workgroups) + ReLU
template <int K>
__device__ void stage_bn(const float* __restrict__ yprev, const float* __restrict__ part, float* xs, float* sc, float* sh, int m0, int M) {
  const int tid = threadIdx.x;
  for (int c = tid; c < K; c += 256) {
    double s = 0, q = 0;
    for (int w = 0; w < NWG; ++w) { s += part[(w * K + c) * 2]; q += part[(w * K + c) * 2 + 1]; }
    const double mean = s / M, var = q / M - mean * mean;
    const float rstd = (float)(1.0 / sqrt(var + 1e-5));
    sc[c] = rstd; sh[c] = (float)(-mean) * rstd;
  }
  __syncthreads();
  for (int i = tid; i < ROWS * K; i += 256) {
    const int row = i / K, k = i % K;
    const float v = m0 + row < M ? yprev[(long)(m0 + row) * K + k] : 0.f;
    xs[row * (K + 1) + k] = fmaxf(fmaf(v, sc[k], sh[k]), 0.f);
  }
  __syncthreads();
}

template <int L>
__device__ void run_link(const Chain& c, float* xs, float* sc, float* sh, float* psum, float* psq, const float* part_prev) {
  constexpr int K = L == 0 ? 128 : (L == 1 ? 128 : 64), N = L == 0 ? 128 : (L == 1 ? 64 : 4);
  const int m0 = blockIdx.x * ROWS, tid = threadIdx.x;
  if (L == 0) {
    for (int i = tid; i < ROWS * K; i += 256) { const int row = i / K, k = i % K; xs[row * (K + 1) + k] = m0 + row < c.M ? c.x0[(long)(m0 + row) * K + k] : 0.f; }
    __syncthreads();
  } else {
    stage_bn<K>(c.y[L - 1], part_prev, xs, sc, sh, m0, c.M);
  }
  for (int n = tid; n < N; n += 256) { psum[n] = 0.f; psq[n] = 0.f; }
  __syncthreads();
  link_body<K, N>(xs, c.W[L], c.y[L], psum, psq, m0, c.M);
  __syncthreads();
}

template <int L>
__global__ __launch_bounds__(256) void link_kernel(const Chain c) {      // variant A: one launch per link
  __shared__ float xs[ROWS * 129], sc[128], sh[128], psum[128], psq[128];
  constexpr int N = L == 0 ? 128 : (L == 1 ? 64 : 4);
  run_link<L>(c, xs, sc, sh, psum, psq, L > 0 ? c.part[L - 1] : nullptr);
  for (int n = threadIdx.x; n < N; n += 256) { c.part[L][(blockIdx.x * N + n) * 2] = psum[n]; c.part[L][(blockIdx.x * N + n) * 2 + 1] = psq[n]; }
}

template <int L>
__device__ void handover_counter(const Chain& c, const float* psum, const float* psq, unsigned round) {
  constexpr int N = L == 0 ? 128 : 64;
  for (int n = threadIdx.x; n < N; n += 256) { c.part[L][(blockIdx.x * N + n) * 2] = psum[n]; c.part[L][(blockIdx.x * N + n) * 2 + 1] = psq[n]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(c.counter + L, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(c.counter + L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round * NWG) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void chain_counter_kernel(const Chain c, unsigned round) {      // variant B
  __shared__ float xs[ROWS * 129], sc[128], sh[128], psum[128], psq[128];
  run_link<0>(c, xs, sc, sh, psum, psq, nullptr);
  handover_counter<0>(c, psum, psq, round);
  run_link<1>(c, xs, sc, sh, psum, psq, c.part[0]);
  handover_counter<1>(c, psum, psq, round);
  run_link<2>(c, xs, sc, sh, psum, psq, c.part[1]);
}

// variant C: partials as tagged granules {tag, value}: every workgroup publishes N x 2 granules and sweeps all NWG x N x 2
template <int L>
__device__ void handover_granules(const Chain& c, const float* psum, const float* psq, float* part_lds) {
  constexpr int N = L == 0 ? 128 : 64;
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  for (int n = threadIdx.x; n < N; n += 256) {
    __hip_atomic_store((gu64*)(c.gran[L] + blockIdx.x * N + n), ((unsigned long long)c.tag << 32) | __float_as_uint(psum[n]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((gu64*)(c.gran2[L] + blockIdx.x * N + n), ((unsigned long long)c.tag << 32) | __float_as_uint(psq[n]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (int i = threadIdx.x; i < NWG * N; i += 256) {
    unsigned long long a, b;
    do {
      a = __hip_atomic_load((gu64*)(c.gran[L] + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      b = __hip_atomic_load((gu64*)(c.gran2[L] + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } while ((unsigned)(a >> 32) != c.tag || (unsigned)(b >> 32) != c.tag);
    part_lds[i * 2] = __uint_as_float((unsigned)a); part_lds[i * 2 + 1] = __uint_as_float((unsigned)b);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void chain_granule_kernel(const Chain c) {      // variant C
  __shared__ float xs[ROWS * 129], sc[128], sh[128], psum[128], psq[128], pl[NWG * 128 * 2];
  run_link<0>(c, xs, sc, sh, psum, psq, nullptr);
  handover_granules<0>(c, psum, psq, pl);
  run_link<1>(c, xs, sc, sh, psum, psq, pl);
  handover_granules<1>(c, psum, psq, pl);
  run_link<2>(c, xs, sc, sh, psum, psq, pl);
}

static float median_us(std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2] * 1e3f; }

int main() {
  const int M = 998;
  Chain c{};
  float* x0; hipMalloc(&x0, M * 128 * 4); hipMemset(x0, 0, M * 128 * 4); c.x0 = x0; c.M = M;
  const int Ks[3] = {128, 128, 64}, Ns[3] = {128, 64, 4};
  std::vector<float> hw(128 * 128, 0.01f);
  for (int l = 0; l < 3; ++l) {
    float* w; hipMalloc(&w, Ns[l] * Ks[l] * 4); hipMemcpy(w, hw.data(), Ns[l] * Ks[l] * 4, hipMemcpyHostToDevice); c.W[l] = w;
    hipMalloc(&c.y[l], M * Ns[l] * 4); hipMalloc(&c.part[l], NWG * 128 * 2 * 4);
    hipMalloc(&c.gran[l], NWG * 128 * 8); hipMalloc(&c.gran2[l], NWG * 128 * 8);
    hipMemset(c.gran[l], 0, NWG * 128 * 8); hipMemset(c.gran2[l], 0, NWG * 128 * 8);
  }
  hipMalloc(&c.counter, 16); hipMemset(c.counter, 0, 16);
  std::vector<float> hx(M * 128);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  hipMemcpy(x0, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  // a large streaming kernel's worth of other work is NOT running: idle chip, as between two latency-bound launches of the step
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 200, chains = 20;
  std::vector<float> ta, tb, tc;
  unsigned round = 0;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0);
    for (int k = 0; k < chains; ++k) {
      hipLaunchKernelGGL(link_kernel<0>, dim3(NWG), dim3(256), 0, 0, c);
      hipLaunchKernelGGL(link_kernel<1>, dim3(NWG), dim3(256), 0, 0, c);
      hipLaunchKernelGGL(link_kernel<2>, dim3(NWG), dim3(256), 0, 0, c);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ta.push_back(ms / chains);
    hipEventRecord(e0);
    for (int k = 0; k < chains; ++k) { ++round; hipLaunchKernelGGL(chain_counter_kernel, dim3(NWG), dim3(256), 0, 0, c, round); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); tb.push_back(ms / chains);
    hipEventRecord(e0);
    for (int k = 0; k < chains; ++k) { ++c.tag; hipLaunchKernelGGL(chain_granule_kernel, dim3(NWG), dim3(256), 0, 0, c); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); tc.push_back(ms / chains);
  }
  // the body alone: the three links without any statistics hand-over (wrong results, timing only) = lower bound of every variant
  printf("FC head chain 128 -> 128 -> 64 -> 4 over %d rows, 32 workgroups of 32 rows, idle chip, back-to-back chains (median of %d x %d):\n", M, reps, chains);
  printf("  A  three dependent launches (partials through memory, fold in the next launch's prologue): %7.2f us per chain\n", median_us(ta));
  printf("  B  one launch, counter hand-over (release fence + arrive + spin + acquire fence) x 2:       %7.2f us per chain\n", median_us(tb));
  printf("  C  one launch, tagged 8-byte granules swept by every workgroup x 2:                        %7.2f us per chain\n", median_us(tc));
  return 0;
}
