// Inference-mode PointNet convolution stack in ONE kernel (reference: the `convs` Sequential + max-pool of
// learning/pointnet.py:27-36,55-58 (STN) and :83-93,126-128 (PointNet) with BatchNorm in eval mode).
//
// With running statistics the BatchNorm of every layer is a per-channel affine map known before the launch, so the
// layers of one superpoint no longer need a grid-wide synchronisation between them.  One WAVEFRONT takes 32 points
// through all 1x1 convolutions (14 -> 64 -> 64 -> 128 -> 128 -> 256 for S3DIS): its activations live in a wave-private LDS
// tile (re-written in place after each layer: the outputs are produced only when the whole reduction has been read), the
// weights stream from L2 straight into MFMA B-operand registers (16-byte loads, one group of 8 reduction steps ahead),
// and there is no workgroup barrier anywhere.  The four waves of a workgroup cover the <=128 points of a superpoint and
// write per-wave max / min partials of the last layer; the usual pool-select kernel combines them.
// HBM traffic per superpoint: the cloud (7 KB) in, 4 partial rows out -- the per-layer path writes and re-reads 0.4 MB of
// activations.  Bound: MFMA (fp32 32x32x2), 15.96 MFLOP per superpoint for the S3DIS stack.
#include "spg_gemm.h"
#include <float.h>

namespace {

constexpr int CS_KMAX = 128;                      // widest layer INPUT

// one <=128-column pass of a layer for the wave's 32*RT rows: acc[i][j] += A_i[32 x K] * W[n0 + 32 j .. +31][K]^T.
// Weights are pre-packed per launch (spg_pack_w_kernel) in MFMA B-operand order: [K/8 groups][cout/32 tiles][64 lanes]
// float4, lane (r, h) of tile t and group g holds W[32 t + r][8 g + 4 h .. +3] (zero beyond cin) -- one fully coalesced
// 1 KiB load per operand instead of 32 strided 32-byte pieces.  Every operand feeds RT row tiles: with RT = 2 the weight
// stream from L2 (every wave reads the whole layer) is halved per FLOP.
template <int RT, int TJ>
__device__ __forceinline__ void cs_gemm_pass(const f32x4* __restrict__ A, const f32x4* __restrict__ Wp, int ntile, int K,
                                             int n0, int r, int h, f32x16 (&acc)[RT][4]) {
  constexpr int APLANE = 32 * RT + 1;
  const int lane = r + 32 * h, t0 = n0 >> 5;
  auto load_b = [&](int k0, f32x4 (&b)[TJ]) {
    const f32x4* src = Wp + ((long)(k0 >> 3) * ntile + t0) * 64 + lane;
#pragma unroll
    for (int j = 0; j < TJ; ++j) b[j] = src[j * 64];
  };
  auto load_a = [&](int k0, f32x4 (&a)[RT]) {
#pragma unroll
    for (int i = 0; i < RT; ++i) a[i] = A[((k0 >> 2) + h) * APLANE + 32 * i + r];
  };
  auto mfma_group = [&](const f32x4 (&a)[RT], const f32x4 (&b)[TJ]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
  };
  // two statically named fragment sets: the loads of the next group of 8 reduction steps fly during the MFMAs of this one
  f32x4 b0[TJ], b1[TJ], a0[RT], a1[RT];
  load_b(0, b0);
  load_a(0, a0);
  for (int k0 = 0; k0 < K; k0 += 16) {
    if (k0 + 8 < K) { load_b(k0 + 8, b1); load_a(k0 + 8, a1); }
    mfma_group(a0, b0);
    if (k0 + 8 < K) {
      if (k0 + 16 < K) { load_b(k0 + 16, b0); load_a(k0 + 16, a0); }
      mfma_group(a1, b1);
    }
  }
}

// RT = 1: a workgroup (4 waves x 32 points) is one superpoint, two workgroups per CU.
// RT = 2: a wave owns 64 points, a workgroup two superpoints, one workgroup per CU.
template <int RT>
__global__ __launch_bounds__(256, 2) void spg_conv_stack_eval_kernel(const SpgConvStackParams p) {
  extern __shared__ f32x4 smem[];
  constexpr int APLANE = 32 * RT + 1, WAVE_F4 = (CS_KMAX / 4) * APLANE, ROWS = 32 * RT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int g = RT == 1 ? (int)blockIdx.x : 2 * (int)blockIdx.x + (wave >> 1);      // superpoint of this wave
  if (g >= p.B) return;                                                               // (no workgroup barriers below)
  const int P = p.P, row0 = RT == 1 ? 32 * wave : 64 * (wave & 1);                     // first point of this wave
  const int part0 = row0 / 32;                                                         // first pooling partial of this wave
  f32x4* A4 = smem + wave * WAVE_F4;                                     // wave-private activation tile [K/4][ROWS+1][4]
  float* A = reinterpret_cast<float*>(A4);

  // ---- the cloud: channel-major [F][P] -> reduction planes, 2x2 spatial transform on (x, y) ----
  {
    const float* cl = p.clouds + (long)g * p.Ctot * P;
    const int K0 = (p.cin[0] + 7) & ~7;
    float T0 = 1.f, T1 = 0.f, T2 = 0.f, T3 = 1.f;
    if (p.stnT != nullptr) {
      const float* T = p.stnT + (long)g * 4;
      T0 = T[0] + 1.f; T1 = T[1]; T2 = T[2]; T3 = T[3] + 1.f;
    }
    for (int idx = lane; idx < K0 * ROWS; idx += 64) {
      const int row = idx % ROWS, k = idx / ROWS;
      const int pt = row0 + row;
      float v = 0.f;
      if (pt < P && k < p.cin[0]) {
        v = cl[(long)k * P + pt];
        if (p.stnT != nullptr && k < 2) {        // learning/pointnet.py:123  [x y] @ (proj.view(2,2) + I)
          const float x = cl[pt], y = cl[(long)P + pt];
          v = k == 0 ? fmaf(x, T0, y * T2) : fmaf(x, T1, y * T3);
        }
      }
      A[((k >> 2) * APLANE + row) * 4 + (k & 3)] = v;
    }
  }
  for (int l = 0; l < p.nlayers; ++l) {
    const int cin = p.cin[l], cout = p.cout[l];
    const int K = (cin + 7) & ~7;
    const bool last = l + 1 == p.nlayers;
    const f32x4* __restrict__ Wp = p.Wp[l];
    const int ntile = cout >> 5;
    const float* __restrict__ cs = p.s[l];
    const float* __restrict__ ct = p.t[l];
    const float* __restrict__ bias = p.bias[l];
    for (int n0 = 0; n0 < cout; n0 += 128) {
      const int NP = min(128, cout - n0);                                // 32, 64 or 128 columns in this pass
      f32x16 acc[RT][4];
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
      if (NP == 128) cs_gemm_pass<RT, 4>(A4, Wp, ntile, K, n0, r, h, acc);
      else if (NP == 64) cs_gemm_pass<RT, 2>(A4, Wp, ntile, K, n0, r, h, acc);
      else cs_gemm_pass<RT, 1>(A4, Wp, ntile, K, n0, r, h, acc);
      const int TJ = NP / 32;
      if (!last) {
        // the whole reduction of this layer has been read: overwrite the tile in place with relu(s * y + t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < TJ) {
            const int col = n0 + 32 * j + r;
            const float b = bias ? bias[col] : 0.f, s = cs[col], t = ct[col];
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
              for (int q = 0; q < 16; ++q) {
                const int row = 32 * i + spg_acc_row(q, h);
                A[((col >> 2) * APLANE + row) * 4 + (col & 3)] = fmaxf(fmaf(acc[i][j][q] + b, s, t), 0.f);
              }
          }
      } else {
        // per-32-point max / min of the raw output over the valid points (spg_pool_select_kernel combines the 4 partials)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < TJ) {
            const int col = n0 + 32 * j + r;
            const float b = bias ? bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
              float vmx = -FLT_MAX, vmn = FLT_MAX;
#pragma unroll
              for (int q = 0; q < 16; ++q) {
                const float v = acc[i][j][q] + b;
                if (row0 + 32 * i + spg_acc_row(q, h) < P) { vmx = fmaxf(vmx, v); vmn = fminf(vmn, v); }
              }
              vmx = fmaxf(vmx, __shfl_xor(vmx, 32, 64));
              vmn = fminf(vmn, __shfl_xor(vmn, 32, 64));
              if (h == 0) {
                const long o = ((long)g * 4 + part0 + i) * cout + col;
                p.pmax[o] = vmx; p.pmin[o] = vmn;
              }
            }
          }
      }
    }
  }
}

// W [cout, cin] row-major -> packed operand order (all layers of the stack in one launch: blockIdx.y = layer)
__global__ void spg_pack_w_kernel(const SpgConvStackParams p) {
  const int l = blockIdx.y;
  if (l >= p.nlayers) return;
  const int cin = p.cin[l], cout = p.cout[l], ntile = cout >> 5, K = (cin + 7) & ~7;
  const long n = (long)(K >> 3) * ntile * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long gt = i >> 6;
    const int t = (int)(gt % ntile), g = (int)(gt / ntile);
    const int row = 32 * t + (lane & 31), k = 8 * g + 4 * (lane >> 5);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = k + e < cin ? p.W[l][(long)row * cin + k + e] : 0.f;
    p.Wp[l][i] = v;
  }
}

}  // namespace

size_t spg_conv_stack_packed_floats(int cin, int cout) { return (size_t)((cin + 7) & ~7) * cout; }

bool spg_conv_stack_eval_supported(const SpgConvStackParams& p) {
  if (p.nlayers < 1 || p.nlayers > SPG_CONVSTACK_MAX_LAYERS || p.P < 1 || p.P > 128) return false;
  if (p.cin[0] < 1 || p.cin[0] > CS_KMAX) return false;
  for (int l = 0; l < p.nlayers; ++l) {
    if (p.cout[l] < 32 || p.cout[l] > 256 || p.W[l] == nullptr || p.Wp[l] == nullptr) return false;
    for (int n0 = 0; n0 < p.cout[l]; n0 += 128) {
      const int np = p.cout[l] - n0 < 128 ? p.cout[l] - n0 : 128;
      if (np != 32 && np != 64 && np != 128) return false;
    }
    if (l > 0 && (p.cin[l] != p.cout[l - 1] || p.cin[l] > CS_KMAX)) return false;
    if (l + 1 < p.nlayers && p.cout[l] > 128) return false;      // an intermediate layer is one pass (in-place tile)
  }
  return true;
}

int spg_launch_conv_stack_eval(const SpgConvStackParams& p, hipStream_t stream) {
  SPG_CHECK_ARG(spg_conv_stack_eval_supported(p), "unsupported layer stack for the fused inference kernel");
  hipLaunchKernelGGL(spg_pack_w_kernel, dim3(16, p.nlayers), dim3(256), 0, stream, p);
  SPG_LAUNCH_CHECK();
  // RT = 2 (64 points per wave, half the weight stream, one workgroup per CU) measured 3 % slower than RT = 1: two
  // waves per SIMD hide more than the L2 traffic costs
  const size_t lds = (size_t)4 * (CS_KMAX / 4) * 33 * sizeof(f32x4);
  hipLaunchKernelGGL(spg_conv_stack_eval_kernel<1>, dim3(p.B), dim3(256), lds, stream, p);
  SPG_LAUNCH_CHECK();
  return 0;
}
