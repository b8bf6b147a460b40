// Operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950, pinned by experiment (used by spg_cloud_gram16_kernel, spg_narrow.hip):
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_f64_probe.hip -o tools/probe/mfma_f64_probe && tools/probe/mfma_f64_probe
// A[i][k] = 1 + i + 100 k, B[k][j] = 1 + j + 100 k under the ASSUMED operand map (lane l: i = j = l % 16, k = l / 16); prints, for every
// (lane, register), which (i, j) of the exact product A B it holds.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
  const int l = threadIdx.x, i = l % 16, k = l / 16;
  const double a = 1.0 + i + 100.0 * k, b = 1.0 + i + 100.0 * k + 0.5;      // B[k][j] = 1.5 + j + 100 k: not symmetric with A
  f64x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int v = 0; v < 4; ++v) out[l * 4 + v] = acc[v];
}
int main() {
  double* d; hipMalloc(&d, 256 * 8);
  probe<<<1, 64>>>(d);
  double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double ref[16][16];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += (1.0 + i + 100.0 * k) * (1.5 + j + 100.0 * k); ref[i][j] = s; }
  int ok_a = 1, ok_b = 1;
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
    int fi = -1, fj = -1;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (ref[i][j] == h[l * 4 + v]) { fi = i; fj = j; }
    if (l < 20 || l % 16 == 0) printf("lane %2d reg %d -> (i %2d, j %2d)\n", l, v, fi, fj);
    if (!(fi == 4 * (l / 16) + v && fj == l % 16)) ok_a = 0;
    if (!(fi == (l / 16) + 4 * v && fj == l % 16)) ok_b = 0;
  }
  printf("layout i = 4 (l / 16) + v, j = l %% 16: %s;  layout i = l / 16 + 4 v, j = l %% 16: %s\n", ok_a ? "YES" : "no", ok_b ? "YES" : "no");
  return 0;
}
