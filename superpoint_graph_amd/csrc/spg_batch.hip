// Device-side batch construction:
//   * spg_set_batch      = GraphConvInfo.set_batch (learning/ecc/GraphConvInfo.py:33-69): edges of the batched graph
//                          ordered by target, idxn = source per edge, degs = in-degree per node, plus the permutation
//                          that brings per-edge data (the edge features) into the same order;
//   * spg_gather_rows    = the edge-feature reordering of :53-56;
//   * spg_edge_features  = spg_edge_features + the StandardScaler transform of scaler01 (learning/spg.py:23-64).
// The order by target is STABLE (ties keep the original edge order) and deterministic: a counting sort whose buckets are
// filled with integer atomics in arbitrary order and then sorted by edge index, one wavefront per target node.  The
// reference orders with numpy's default argsort, whose tie order is unspecified; every target segment holds the same
// edges either way (tests/test_gpu_batch.py compares segment-wise and through the model outputs).
#include "../../include/spg_hip.h"
#include "spg_common.h"

namespace {

__global__ void count_targets_kernel(const int64_t* __restrict__ edges, int E, int N, int* __restrict__ deg, int* __restrict__ err) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t s = edges[2 * (long)e], t = edges[2 * (long)e + 1];
  if (t < 0 || t >= N || s < 0 || s >= N) { *err = 1; return; }
  atomicAdd(&deg[(int)t], 1);
}

// exclusive scan of deg[0..N) into rowptr[0..N]; one workgroup (N is a few thousand superpoints per batch)
__global__ __launch_bounds__(1024) void scan_kernel(const int* __restrict__ deg, int N, int* __restrict__ rowptr,
                                                    int64_t* __restrict__ degs64) {
  __shared__ int part[1024];
  const int tid = threadIdx.x, per = (N + 1023) / 1024;
  const int b = tid * per, e = min(N, b + per);
  int s = 0;
  for (int i = b; i < e; ++i) s += deg[i];
  part[tid] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = tid ? part[tid - 1] : 0;
  for (int i = b; i < e; ++i) {
    rowptr[i] = run;
    degs64[i] = deg[i];
    run += deg[i];
  }
  if (tid == 1023) rowptr[N] = part[1023];
}

__global__ void fill_buckets_kernel(const int64_t* __restrict__ edges, int E, int N, const int* __restrict__ rowptr,
                                    int* __restrict__ cursor, int* __restrict__ bucket) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t t = edges[2 * (long)e + 1];
  if (t < 0 || t >= N) return;
  const int slot = atomicAdd(&cursor[(int)t], 1);          // arrival order is arbitrary: the next kernel sorts the bucket
  bucket[rowptr[(int)t] + slot] = e;
}

// one wavefront per target node: its bucket (edge indices) in ascending order = original edge order (stable)
__global__ __launch_bounds__(256) void sort_buckets_kernel(const int64_t* __restrict__ edges, int N, const int* __restrict__ rowptr,
                                                           const int* __restrict__ bucket, int64_t* __restrict__ idxn,
                                                           int64_t* __restrict__ perm) {
  const int node = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (node >= N) return;
  const int b = rowptr[node], d = rowptr[node + 1] - b;
  for (int i = lane; i < d; i += 64) {                     // rank of element i = number of smaller edge indices in the bucket
    const int v = bucket[b + i];
    int rank = 0;
    for (int j = 0; j < d; ++j) rank += bucket[b + j] < v;
    perm[b + rank] = v;
    idxn[b + rank] = edges[2 * (long)v];
  }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, long lds_, const int64_t* __restrict__ perm, long rows, int cols,
                                   float* __restrict__ dst, long ldd) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const long r = i / cols;
  const int c = (int)(i - r * cols);
  const long s = perm[r];
  dst[r * ldd + c] = s >= 0 ? src[s * lds_ + c] : 0.f;      // a negative index = a zero row (embedding scatter)
}

__global__ void edge_features_kernel(const spg_edge_feature_spec* __restrict__ specs_unused, spg_edge_feature_specs S,
                                     const int64_t* __restrict__ edges, long E, const double* __restrict__ mean,
                                     const double* __restrict__ scale, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E * S.ncols) return;
  const long e = i / S.ncols;
  const int c = (int)(i - e * S.ncols);
  const spg_edge_feature_spec sp = S.col[c];
  const long a = edges[2 * e], b = edges[2 * e + 1];
  float v;
  if (sp.kind == SPG_EF_COPY) {      // numpy's .astype(float32) of the attribute column, whatever its storage type
    v = sp.is_f64 ? (float)((const double*)sp.data)[e * sp.ld + sp.column] : ((const float*)sp.data)[e * sp.ld + sp.column];
  } else if (sp.kind == SPG_EF_CONST) {
    v = 1.f;
  } else if (sp.is_f64) {                                   // e.g. the point count (u64 in the file): float64 arithmetic, one final rounding
    const double* d = (const double*)sp.data;
    const double x = d[a * sp.ld + sp.column], y = d[b * sp.ld + sp.column];
    double r;
    if (sp.kind == SPG_EF_DIFF) r = x - y;
    else if (sp.kind == SPG_EF_LOGDIFF) r = log(x + 1e-10) - log(y + 1e-10);
    else r = x / (y + 1e-10);
    v = (float)r;
  } else {                                                  // float32 attributes: float32 arithmetic like numpy
    const float* d = (const float*)sp.data;
    const float x = d[a * sp.ld + sp.column], y = d[b * sp.ld + sp.column];
    if (sp.kind == SPG_EF_DIFF) v = x - y;
    else if (sp.kind == SPG_EF_LOGDIFF) v = logf(x + 1e-10f) - logf(y + 1e-10f);
    else v = x / (y + 1e-10f);
  }
  if (mean != nullptr) {                                    // StandardScaler.transform on a float32 array (in place: two roundings)
    v = (float)((double)v - mean[c]);
    v = (float)((double)v / scale[c]);
  }
  out[i] = v;
}

}  // namespace

extern "C" size_t spg_set_batch_workspace_bytes(int N, int E) {
  // deg[N] | cursor[N] | rowptr[N+1] | bucket[E] | err
  return ((size_t)3 * (N + 1) + (size_t)E + 8) * sizeof(int) + 256;
}

extern "C" int spg_set_batch(const int64_t* edges, int N, int E, int64_t* idxn, int64_t* degs, int64_t* perm, void* workspace,
                             int32_t* error_flag, void* stream) {
  SPG_CHECK_ARG(N > 0 && E >= 0 && degs && workspace && (E == 0 || (edges && idxn && perm)), "bad argument");
  hipStream_t st = (hipStream_t)stream;
  int* deg = (int*)workspace;
  int* cursor = deg + (N + 1);
  int* rowptr = cursor + (N + 1);
  int* bucket = rowptr + (N + 1);
  int* err = bucket + E;
  hipError_t rc = hipMemsetAsync(workspace, 0, ((size_t)2 * (N + 1)) * sizeof(int), st);
  if (rc == hipSuccess) rc = hipMemsetAsync(err, 0, sizeof(int), st);
  if (rc != hipSuccess) { spg_set_error("hipMemsetAsync: %s", hipGetErrorString(rc)); return (int)rc; }
  if (E > 0) {
    hipLaunchKernelGGL(count_targets_kernel, dim3(spg_cdiv(E, 256)), dim3(256), 0, st, edges, E, N, deg, err);
    SPG_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, deg, N, rowptr, degs);
  SPG_LAUNCH_CHECK();
  if (E > 0) {
    hipLaunchKernelGGL(fill_buckets_kernel, dim3(spg_cdiv(E, 256)), dim3(256), 0, st, edges, E, N, rowptr, cursor, bucket);
    SPG_LAUNCH_CHECK();
    hipLaunchKernelGGL(sort_buckets_kernel, dim3(spg_cdiv(N, 4)), dim3(256), 0, st, edges, N, rowptr, bucket, idxn, perm);
    SPG_LAUNCH_CHECK();
  }
  if (error_flag != nullptr) {
    rc = hipMemcpyAsync(error_flag, err, sizeof(int), hipMemcpyDeviceToDevice, st);
    if (rc != hipSuccess) { spg_set_error("hipMemcpyAsync: %s", hipGetErrorString(rc)); return (int)rc; }
  }
  return 0;
}

extern "C" int spg_gather_rows(const float* src, long ld_src, const int64_t* perm, long rows, int cols, float* dst, long ld_dst,
                               void* stream) {
  SPG_CHECK_ARG(rows == 0 || (src && perm && dst && cols > 0), "bad argument");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(spg_cdiv(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, src, ld_src, perm,
                     rows, cols, dst, ld_dst);
  SPG_LAUNCH_CHECK();
  return 0;
}

extern "C" int spg_edge_features(const spg_edge_feature_specs* specs, const int64_t* edges, long E, const double* mean,
                                 const double* scale, float* out, void* stream) {
  SPG_CHECK_ARG(specs && out && (E == 0 || edges) && specs->ncols > 0 && specs->ncols <= SPG_EF_MAX_COLS, "bad argument");
  SPG_CHECK_ARG((mean == nullptr) == (scale == nullptr), "mean and scale go together");
  if (E == 0) return 0;
  hipLaunchKernelGGL(edge_features_kernel, dim3(spg_cdiv(E * specs->ncols, 256)), dim3(256), 0, (hipStream_t)stream, nullptr, *specs,
                     edges, E, mean, scale, out);
  SPG_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// spg_upload: page-locked staging ring for the small per-batch host -> device copies
// ------------------------------------------------------------------------------------------------------------------
#include <cstring>
#include <mutex>

namespace {

struct StagingSlot {
  void* buf = nullptr;
  size_t cap = 0;
  hipEvent_t done = nullptr;
  bool pending = false;
  bool own = false;          // buf is this slot's own allocation (not a piece of the device's shared block)
};
// two rings per device: 32 slots for the small per-batch vectors (<= 1 MiB: edge lists, index vectors, edge features) and 4
// slots for large buffers (clouds of a host-side loader), so that page-locked memory stays bounded by 4 x the largest upload
constexpr int kSmallSlots = 32, kLargeSlots = 4;
constexpr size_t kSmallBytes = 1u << 20;
// the small slots start as 512 KiB pieces of ONE page-locked block per device (a hipHostMalloc costs ~0.5 ms: 32 of them would
// spread over the first steps of a run); a slot that meets a larger buffer gets its own allocation
constexpr size_t kSmallInitial = 512u << 10;
struct StagingRing {
  std::mutex mu;
  StagingSlot slot[SPG_MAX_DEVICES][kSmallSlots + kLargeSlots];
  int next_small[SPG_MAX_DEVICES] = {}, next_large[SPG_MAX_DEVICES] = {};
  void* block[SPG_MAX_DEVICES] = {};
};
StagingRing g_staging;

}  // namespace

// a staging slot of the current device with room for `bytes`, free for the host to write (its last copy has left the buffer);
// the ring's mutex is held only while the slot is chosen -- the caller owns the slot until it records `done` again
static int staging_take(int dev, size_t bytes, StagingSlot** out) {
  std::lock_guard<std::mutex> lock(g_staging.mu);
  int idx;
  if (bytes <= kSmallBytes) {
    idx = g_staging.next_small[dev];
    g_staging.next_small[dev] = (idx + 1) % kSmallSlots;
  } else {
    idx = kSmallSlots + g_staging.next_large[dev];
    g_staging.next_large[dev] = (g_staging.next_large[dev] + 1) % kLargeSlots;
  }
  hipError_t rc;
  if (g_staging.block[dev] == nullptr) {
    rc = hipHostMalloc(&g_staging.block[dev], kSmallSlots * kSmallInitial, hipHostMallocDefault);
    if (rc != hipSuccess) { g_staging.block[dev] = nullptr; spg_set_error("hipHostMalloc: %s", hipGetErrorString(rc)); return (int)rc; }
    for (int k = 0; k < kSmallSlots; ++k) {
      g_staging.slot[dev][k].buf = (char*)g_staging.block[dev] + (size_t)k * kSmallInitial;
      g_staging.slot[dev][k].cap = kSmallInitial;
    }
  }
  StagingSlot& s = g_staging.slot[dev][idx];
  if (s.done == nullptr) {
    rc = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
    if (rc != hipSuccess) { spg_set_error("hipEventCreate: %s", hipGetErrorString(rc)); return (int)rc; }
  }
  if (s.pending) {      // the copy that last used this slot must have left the staging buffer
    rc = hipEventSynchronize(s.done);
    if (rc != hipSuccess) { spg_set_error("hipEventSynchronize: %s", hipGetErrorString(rc)); return (int)rc; }
    s.pending = false;
  }
  if (s.cap < bytes) {
    if (s.buf != nullptr && s.own) (void)hipHostFree(s.buf);
    s.buf = nullptr; s.cap = 0; s.own = false;
    size_t cap = 64 * 1024;
    while (cap < bytes) cap *= 2;
    rc = hipHostMalloc(&s.buf, cap, hipHostMallocDefault);
    if (rc != hipSuccess) { s.buf = nullptr; spg_set_error("hipHostMalloc(%zu): %s", cap, hipGetErrorString(rc)); return (int)rc; }
    s.cap = cap; s.own = true;
  }
  *out = &s;
  return 0;
}

static int upload_impl(const void* host, size_t bytes, void* device, void* stream) {
  if (bytes == 0) return 0;
  SPG_CHECK_ARG(host && device, "bad argument");
  int dev = 0;
  hipError_t rc = hipGetDevice(&dev);
  if (rc != hipSuccess || dev < 0 || dev >= SPG_MAX_DEVICES) { spg_set_error("spg_upload: hipGetDevice failed"); return rc ? (int)rc : 1; }
  StagingSlot* sp = nullptr;
  SPG_TRY(staging_take(dev, bytes, &sp));
  StagingSlot& s = *sp;
  std::memcpy(s.buf, host, bytes);
  rc = hipMemcpyAsync(device, s.buf, bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
  if (rc != hipSuccess) { spg_set_error("hipMemcpyAsync: %s", hipGetErrorString(rc)); return (int)rc; }
  rc = hipEventRecord(s.done, (hipStream_t)stream);
  if (rc != hipSuccess) { spg_set_error("hipEventRecord: %s", hipGetErrorString(rc)); return (int)rc; }
  s.pending = true;
  return 0;
}

extern "C" int spg_upload(const void* host, size_t bytes, void* device, void* stream) { return upload_impl(host, bytes, device, stream); }

// Several small host buffers -> ONE staging slot -> ONE asynchronous copy into `device` (piece i lands at device + offsets[i];
// the caller lays the pieces out: 256-byte aligned, non-overlapping, inside `total` bytes).  A fresh batch needs seven small
// vectors on the device (edge list, edge features, CloudEmbedder's two index vectors, labels, diameters): as seven spg_upload
// calls they cost the host ~0.1 ms each (memcpy + hipMemcpyAsync + event record) -- half of the fresh-batch loop's 1.0 ms of host
// time per step (profiles/r06_tw_host_profile.txt).
extern "C" int spg_upload_packed(const void* const* host, const size_t* bytes, const size_t* offsets, int n, void* device, size_t total,
                                 void* stream) {
  if (n <= 0 || total == 0) return 0;
  SPG_CHECK_ARG(host && bytes && offsets && device, "bad argument");
  for (int i = 0; i < n; ++i) SPG_CHECK_ARG((bytes[i] == 0 || host[i] != nullptr) && offsets[i] + bytes[i] <= total, "piece outside the packed buffer");
  int dev = 0;
  hipError_t rc = hipGetDevice(&dev);
  if (rc != hipSuccess || dev < 0 || dev >= SPG_MAX_DEVICES) { spg_set_error("spg_upload_packed: hipGetDevice failed"); return rc ? (int)rc : 1; }
  StagingSlot* sp = nullptr;
  SPG_TRY(staging_take(dev, total, &sp));
  for (int i = 0; i < n; ++i)
    if (bytes[i]) std::memcpy((char*)sp->buf + offsets[i], host[i], bytes[i]);
  rc = hipMemcpyAsync(device, sp->buf, total, hipMemcpyHostToDevice, (hipStream_t)stream);
  if (rc != hipSuccess) { spg_set_error("hipMemcpyAsync: %s", hipGetErrorString(rc)); return (int)rc; }
  rc = hipEventRecord(sp->done, (hipStream_t)stream);
  if (rc != hipSuccess) { spg_set_error("hipEventRecord: %s", hipGetErrorString(rc)); return (int)rc; }
  sp->pending = true;
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// spg_batch_graph_build: the whole batch construction of a SMALL batch in ONE launch
// ------------------------------------------------------------------------------------------------------------------
// GraphConvInfo.set_batch (learning/ecc/GraphConvInfo.py:33-69) + `.cuda()` (:71-79) + the CSR / reverse CSR of spg_graph_build
// for a batch of a few scenes (N <= 4096 nodes, E <= 32768 edges): the multi-launch path is ~20 launches of 3-10 us and 5
// copies per batch -- all latency, and on the side stream every one of them takes a workgroup slot next to the persistent
// GEMMs of the step in flight.  Here ONE workgroup of 1024 threads walks the phases (count -> scan -> bucket fill -> per-node
// order -> outputs; the per-node counters and cursors in LDS) with workgroup barriers in between, a second small launch
// reorders the edge features; the edge list and the edge features come from HOST pointers through
// the staging ring.  Same results as spg_set_batch + spg_gather_rows + spg_graph_build: the order by target is the stable one
// (ties by edge index), integer atomics only, deterministic.
#include "spg_ecc.h"

namespace {

constexpr int kBgMaxNodes = 4096, kBgMaxEdges = 32768;

struct BatchGraphArgs {
  const int64_t* edges;      // [E][2] device
  const float* feats;        // [E][F] device or null
  int N, E, F;
  int64_t* idxn;             // [E]
  int64_t* degs;             // [N]
  float* feats_sorted;       // [E][F]
  int *hdr, *rowptr, *src, *dst, *rev_rowptr, *rev_eid;
  float* invdeg;
  int *bucket_in, *bucket_out, *inv;      // scratch: [E] each
  int32_t* error_flag;       // may be null
};

// exclusive scan of cnt[0..n) (LDS) -> out[0..n] (global), and cnt[i] := out[i] (the bucket cursors start at the segment starts)
__device__ __forceinline__ void block_exscan_lds(int* __restrict__ cnt, int n, int* __restrict__ out, int* part) {
  const int t = threadIdx.x, chunk = (n + 1023) / 1024;
  const int b = min(n, t * chunk), e = min(n, b + chunk);
  int s = 0;
  for (int i = b; i < e; ++i) s += cnt[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int base = part[t] - s;
  for (int i = b; i < e; ++i) { const int c = cnt[i]; out[i] = base; cnt[i] = base; base += c; }
  if (t == 1023) out[n] = part[1023];
  __syncthreads();
}

__device__ __forceinline__ void insertion_sort(int* __restrict__ a, int b, int e) {
  for (int i = b + 1; i < e; ++i) {
    const int v = a[i];
    int k = i - 1;
    while (k >= b && a[k] > v) { a[k + 1] = a[k]; --k; }
    a[k + 1] = v;
  }
}

// one workgroup: the counters / cursors of all nodes live in LDS (integer LDS atomics: order-independent results)
__global__ __launch_bounds__(1024) void batch_graph_kernel(const BatchGraphArgs a) {
  __shared__ int s_in[kBgMaxNodes], s_out[kBgMaxNodes];
  __shared__ int part[1024];
  __shared__ int bad;
  const int t = threadIdx.x, N = a.N, E = a.E;
  if (t == 0) bad = 0;
  for (int i = t; i < N; i += 1024) { s_in[i] = 0; s_out[i] = 0; }
  __syncthreads();
  for (int e = t; e < E; e += 1024) {
    const int64_t s = a.edges[2 * (long)e], d = a.edges[2 * (long)e + 1];
    if (s < 0 || s >= N || d < 0 || d >= N) { bad = 1; continue; }
    atomicAdd(&s_in[(int)d], 1);
    atomicAdd(&s_out[(int)s], 1);
  }
  __syncthreads();
  if (t == 0) {
    a.hdr[0] = N; a.hdr[1] = N; a.hdr[2] = E; a.hdr[3] = bad;
    if (a.error_flag != nullptr) *a.error_flag = bad;
  }
  if (bad) return;                       // malformed edge list: flagged, nothing else is written (the host checked it before)
  for (int i = t; i < N; i += 1024) {
    const int d = s_in[i];
    a.degs[i] = d;
    a.invdeg[i] = d > 0 ? 1.0f / (float)d : 0.f;
  }
  __syncthreads();
  block_exscan_lds(s_in, N, a.rowptr, part);
  block_exscan_lds(s_out, N, a.rev_rowptr, part);
  for (int e = t; e < E; e += 1024) {
    const int s = (int)a.edges[2 * (long)e], d = (int)a.edges[2 * (long)e + 1];
    a.bucket_in[atomicAdd(&s_in[d], 1)] = e;
    a.bucket_out[atomicAdd(&s_out[s], 1)] = e;
  }
  __syncthreads();                       // s_in[i] / s_out[i] are now the segment ENDS
  // order every target's bucket by edge index (= the stable order by target), emit the edge arrays
  for (int i = t; i < N; i += 1024) {
    const int e = s_in[i], b = e - (int)a.degs[i];
    insertion_sort(a.bucket_in, b, e);
    for (int p = b; p < e; ++p) {
      const int eo = a.bucket_in[p];
      const int64_t s = a.edges[2 * (long)eo];
      a.idxn[p] = s;
      a.src[p] = (int)s;
      a.dst[p] = i;
      a.inv[eo] = p;
    }
  }
  __syncthreads();
  // reverse CSR: for every source the NEW ids of its out-edges, ascending
  for (int j = t; j < N; j += 1024) {
    const int e = s_out[j], b = a.rev_rowptr[j];
    for (int p = b; p < e; ++p) a.bucket_out[p] = a.inv[a.bucket_out[p]];
    insertion_sort(a.bucket_out, b, e);
    for (int p = b; p < e; ++p) a.rev_eid[p] = a.bucket_out[p];
  }
}

// edge features in the new order: feats_sorted[p, :] = feats[perm[p], :]
// `bad` = the builder's header word 3: a malformed edge list left `perm` unwritten (scratch) -- nothing is gathered then
__global__ void gather_rows_i32_kernel(const float* __restrict__ src, const int* __restrict__ perm, long rows, int cols, float* __restrict__ dst,
                                       const int* __restrict__ bad) {
  const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= rows * cols || bad[3] != 0) return;
  const long p = u / cols;
  dst[u] = src[(long)perm[p] * cols + (u - p * cols)];
}

size_t bg_al(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t spg_batch_graph_scratch_bytes(int N, int E, int F) {
  if (N < 1 || E < 0 || N > kBgMaxNodes || E > kBgMaxEdges) return 0;      // 0: not applicable, use the multi-launch path
  const size_t e = (size_t)(E > 0 ? E : 1);
  return bg_al(e * 16) + bg_al(e * (size_t)(F > 0 ? F : 1) * 4) + 3 * bg_al(e * 4) + 256;
}

// inputs_on_device: edges / feats are DEVICE arrays already (uploaded with the batch's other small vectors by ONE spg_upload_packed)
static int batch_graph_build_impl(const int64_t* edges_in, const float* feats_in, bool inputs_on_device, int N, int E, int F, int64_t* idxn,
                                  int64_t* degs, float* feats_sorted, void* graph_ws, void* scratch, int32_t* error_flag, void* stream) {
  SPG_CHECK_ARG(N >= 1 && N <= kBgMaxNodes && E >= 0 && E <= kBgMaxEdges && F >= 0, "batch too large for the single-launch builder");
  SPG_CHECK_ARG(degs && graph_ws && scratch && (E == 0 || (edges_in && idxn)) && (F == 0 || E == 0 || (feats_in && feats_sorted)), "null pointer");
  const size_t e = (size_t)(E > 0 ? E : 1);
  char* w = (char*)scratch;
  BatchGraphArgs a;
  a.edges = (const int64_t*)w; w += bg_al(e * 16);
  a.feats = F > 0 ? (const float*)w : nullptr; w += bg_al(e * (size_t)(F > 0 ? F : 1) * 4);
  a.bucket_in = (int*)w; w += bg_al(e * 4);
  a.bucket_out = (int*)w; w += bg_al(e * 4);
  a.inv = (int*)w;
  if (inputs_on_device) {
    a.edges = edges_in;
    a.feats = F > 0 ? feats_in : nullptr;
  } else if (E > 0) {
    SPG_TRY(upload_impl(edges_in, (size_t)E * 16, (void*)a.edges, stream));
    if (F > 0) SPG_TRY(upload_impl(feats_in, (size_t)E * F * 4, (void*)a.feats, stream));
  }
  SpgGraph g = spg_graph_view(graph_ws, N, E);
  a.N = N; a.E = E; a.F = F; a.idxn = idxn; a.degs = degs; a.feats_sorted = feats_sorted;
  a.hdr = (int*)g.hdr; a.rowptr = (int*)g.rowptr; a.src = (int*)g.src; a.dst = (int*)g.dst;
  a.rev_rowptr = (int*)g.rev_rowptr; a.rev_eid = (int*)g.rev_eid; a.invdeg = (float*)g.invdeg;
  a.error_flag = error_flag;
  hipLaunchKernelGGL(batch_graph_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  SPG_LAUNCH_CHECK();
  if (E > 0 && F > 0) {      // (a malformed edge list leaves bucket_in unwritten: the gather reads the builder's flag and does nothing)
    hipLaunchKernelGGL(gather_rows_i32_kernel, dim3(spg_cdiv((long)E * F, 256)), dim3(256), 0, (hipStream_t)stream, a.feats,
                       (const int*)a.bucket_in, (long)E, F, feats_sorted, (const int*)a.hdr);
    SPG_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int spg_batch_graph_build(const int64_t* edges_host, const float* feats_host, int N, int E, int F, int64_t* idxn, int64_t* degs,
                                     float* feats_sorted, void* graph_ws, void* scratch, int32_t* error_flag, void* stream) {
  return batch_graph_build_impl(edges_host, feats_host, false, N, E, F, idxn, degs, feats_sorted, graph_ws, scratch, error_flag, stream);
}
extern "C" int spg_batch_graph_build_dev(const int64_t* edges_dev, const float* feats_dev, int N, int E, int F, int64_t* idxn, int64_t* degs,
                                         float* feats_sorted, void* graph_ws, void* scratch, int32_t* error_flag, void* stream) {
  return batch_graph_build_impl(edges_dev, feats_dev, true, N, E, F, idxn, degs, feats_sorted, graph_ws, scratch, error_flag, stream);
}
