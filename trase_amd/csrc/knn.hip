// knn.hip -- simple_knn._C.distCUDA2 (call site scene/gaussian_model.py:237): for every point the
// mean of the squared distances to its 3 nearest neighbours (exact).
//
// MI355X design: a spatial hash instead of the lineage's Morton-sort + box pruning.  Points are
// bucketed by hash(cell) with the stable radix sort of binning.hip, a query walks cube shells of
// cells outwards until the 3rd-best distance is provably final.  Hash collisions only add
// candidates that are filtered by their true cell, so the result is exact for any cell size.
#include "common.h"

namespace trase {

// device-resident parameters (no host round trip after the bounding-box reduction)
struct KnnParams {
  uint32_t n;          // number of points (device copy for the sort)
  uint32_t bb[6];      // order-preserving uint encodings of min xyz / max xyz
  float origin[3];
  float h, inv_h;
  int dims[3];
};

__device__ __forceinline__ uint32_t f2ord(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

__device__ __forceinline__ uint32_t cell_hash(int x, int y, int z, uint32_t mask) {
  uint32_t h = (uint32_t)x * 73856093u ^ (uint32_t)y * 19349663u ^ (uint32_t)z * 83492791u;
  h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
  return h & mask;
}

__global__ void knn_init_kernel(KnnParams* p, uint32_t n) {
  p->n = n;
  p->bb[0] = p->bb[1] = p->bb[2] = 0xffffffffu;
  p->bb[3] = p->bb[4] = p->bb[5] = 0u;
}

__global__ __launch_bounds__(256) void knn_bbox_kernel(const float* __restrict__ pts, int n, KnnParams* p) {
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float v = pts[3 * i + k];
      mn[k] = fminf(mn[k], v);
      mx[k] = fmaxf(mx[k], v);
    }
  }
  __shared__ float smn[3][256], smx[3][256];
#pragma unroll
  for (int k = 0; k < 3; ++k) { smn[k][threadIdx.x] = mn[k]; smx[k][threadIdx.x] = mx[k]; }
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < (unsigned)s) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        smn[k][threadIdx.x] = fminf(smn[k][threadIdx.x], smn[k][threadIdx.x + s]);
        smx[k][threadIdx.x] = fmaxf(smx[k][threadIdx.x], smx[k][threadIdx.x + s]);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) {
    atomicMin(&p->bb[threadIdx.x], f2ord(smn[threadIdx.x][0]));
    atomicMax(&p->bb[3 + threadIdx.x], f2ord(smx[threadIdx.x][0]));
  }
}

__global__ void knn_params_kernel(KnnParams* p) {
  float e[3];
  for (int k = 0; k < 3; ++k) {
    p->origin[k] = ord2f(p->bb[k]);
    e[k] = fmaxf(ord2f(p->bb[3 + k]) - p->origin[k], 0.f);
  }
  // sort extents descending
  float a = e[0], b = e[1], c = e[2];
  if (a < b) { float t = a; a = b; b = t; }
  if (b < c) { float t = b; b = c; c = t; }
  if (a < b) { float t = a; a = b; b = t; }
  const float n = (float)(p->n > 0 ? p->n : 1);
  const float occ = 4.0f;                       // target points per cell
  const float h1 = occ * a / n;
  const float h2 = sqrtf(occ * a * b / n);
  const float h3 = cbrtf(occ * a * b * c / n);
  float h = (c < h2) ? ((b < h1) ? h1 : h2) : h3;  // effective dimensionality of the cloud
  h = fmaxf(h, a * (1.0f / 1048576.0f));           // at most 2^20 cells per axis
  if (!(h > 0.f)) h = 1.0f;                         // all points coincide
  p->h = h;
  p->inv_h = 1.0f / h;
  for (int k = 0; k < 3; ++k) p->dims[k] = (int)(e[k] * p->inv_h) + 1;
}

__device__ __forceinline__ void cell_of(const KnnParams& p, float x, float y, float z, int& cx, int& cy, int& cz) {
  cx = min(p.dims[0] - 1, max(0, (int)((x - p.origin[0]) * p.inv_h)));
  cy = min(p.dims[1] - 1, max(0, (int)((y - p.origin[1]) * p.inv_h)));
  cz = min(p.dims[2] - 1, max(0, (int)((z - p.origin[2]) * p.inv_h)));
}

__global__ __launch_bounds__(256) void knn_keys_kernel(const float* __restrict__ pts, int n, const KnnParams* pp,
                                                       uint32_t mask, uint32_t* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const KnnParams p = *pp;
  int cx, cy, cz;
  cell_of(p, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], cx, cy, cz);
  keys[i] = cell_hash(cx, cy, cz, mask);
}

__device__ __forceinline__ void best3_insert(float d, float& b0, float& b1, float& b2) {
  if (d < b2) {
    if (d < b1) {
      b2 = b1;
      if (d < b0) { b1 = b0; b0 = d; } else { b1 = d; }
    } else {
      b2 = d;
    }
  }
}

__global__ __launch_bounds__(256) void knn_query_kernel(const float* __restrict__ pts, int n, const KnnParams* pp,
                                                        uint32_t mask, const uint32_t* __restrict__ sorted_ids,
                                                        const uint2* __restrict__ buckets, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const KnnParams p = *pp;
  const uint32_t self = sorted_ids[i];
  const float x = pts[3 * self], y = pts[3 * self + 1], z = pts[3 * self + 2];
  int cx, cy, cz;
  cell_of(p, x, y, z, cx, cy, cz);
  float b0 = 3.0e38f, b1 = 3.0e38f, b2 = 3.0e38f;
  const int rmax = max(p.dims[0], max(p.dims[1], p.dims[2]));
  for (int r = 0; r <= rmax; ++r) {
    const int z0 = max(cz - r, 0), z1 = min(cz + r, p.dims[2] - 1);
    const int y0 = max(cy - r, 0), y1 = min(cy + r, p.dims[1] - 1);
    const int x0 = max(cx - r, 0), x1 = min(cx + r, p.dims[0] - 1);
    for (int zz = z0; zz <= z1; ++zz)
      for (int yy = y0; yy <= y1; ++yy) {
        const bool face = (abs(zz - cz) == r) || (abs(yy - cy) == r);
        // on a face row every x is part of the shell; otherwise only the two end cells
        const int step = face ? 1 : max(1, 2 * r);
        for (int xx = face ? x0 : cx - r; xx <= (face ? x1 : cx + r); xx += step) {
          if (xx < 0 || xx >= p.dims[0]) continue;
          const uint2 bk = buckets[cell_hash(xx, yy, zz, mask)];
          for (uint32_t k = bk.x; k < bk.y; ++k) {
            const uint32_t j = sorted_ids[k];
            if (j == self) continue;
            const float qx = pts[3 * j], qy = pts[3 * j + 1], qz = pts[3 * j + 2];
            int ox, oy, oz;
            cell_of(p, qx, qy, qz, ox, oy, oz);
            if (ox != xx || oy != yy || oz != zz) continue;   // hash collision: belongs to another cell
            const float dx = qx - x, dy = qy - y, dz = qz - z;
            best3_insert(dx * dx + dy * dy + dz * dz, b0, b1, b2);
          }
        }
      }
    // everything in shells > r is farther than r*h
    const float bound = (float)r * p.h;
    if (b2 <= bound * bound) break;
  }
  // fewer than 4 points: the lineage's behaviour is undefined; average what exists
  float sum = 0.f; int cnt = 0;
  if (b0 < 3.0e38f) { sum += b0; ++cnt; }
  if (b1 < 3.0e38f) { sum += b1; ++cnt; }
  if (b2 < 3.0e38f) { sum += b2; ++cnt; }
  out[self] = cnt ? sum / 3.0f : 0.f;
}

// top-KT nearest points of the hashed cloud `pts` for every query point; ascending, squared distance
template <int KT>
__global__ __launch_bounds__(256) void knn_points_kernel(const float* __restrict__ qpts, int nq,
                                                         const float* __restrict__ pts, int n, int K,
                                                         const KnnParams* pp, uint32_t mask,
                                                         const uint32_t* __restrict__ sorted_ids,
                                                         const uint2* __restrict__ buckets,
                                                         int64_t* __restrict__ idx_out, float* __restrict__ dist_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const KnnParams p = *pp;
  const float x = qpts[3 * i], y = qpts[3 * i + 1], z = qpts[3 * i + 2];
  int cx, cy, cz;
  cell_of(p, x, y, z, cx, cy, cz);
  // distance from the query to its (clamped) cell: queries outside the cloud's box start farther away
  float bd[KT];
  uint32_t bi[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) { bd[k] = 3.0e38f; bi[k] = 0u; }
  const int rmax = max(p.dims[0], max(p.dims[1], p.dims[2]));
  for (int r = 0; r <= rmax; ++r) {
    const int z0 = max(cz - r, 0), z1 = min(cz + r, p.dims[2] - 1);
    const int y0 = max(cy - r, 0), y1 = min(cy + r, p.dims[1] - 1);
    const int x0 = max(cx - r, 0), x1 = min(cx + r, p.dims[0] - 1);
    for (int zz = z0; zz <= z1; ++zz)
      for (int yy = y0; yy <= y1; ++yy) {
        const bool face = (abs(zz - cz) == r) || (abs(yy - cy) == r);
        const int step = face ? 1 : max(1, 2 * r);
        for (int xx = face ? x0 : cx - r; xx <= (face ? x1 : cx + r); xx += step) {
          if (xx < 0 || xx >= p.dims[0]) continue;
          const uint2 bk = buckets[cell_hash(xx, yy, zz, mask)];
          for (uint32_t k = bk.x; k < bk.y; ++k) {
            const uint32_t j = sorted_ids[k];
            const float qx = pts[3 * j], qy = pts[3 * j + 1], qz = pts[3 * j + 2];
            int ox, oy, oz;
            cell_of(p, qx, qy, qz, ox, oy, oz);
            if (ox != xx || oy != yy || oz != zz) continue;   // hash collision: belongs to another cell
            const float dx = qx - x, dy = qy - y, dz = qz - z;
            float d = dx * dx + dy * dy + dz * dz;
            uint32_t id = j;
            if (d < bd[KT - 1] || (d == bd[KT - 1] && id < bi[KT - 1])) {
              // sorted insertion (ties by index so the result does not depend on the bucket order)
#pragma unroll
              for (int t = 0; t < KT; ++t) {
                const bool lt = d < bd[t] || (d == bd[t] && id < bi[t]);
                const float td = lt ? bd[t] : d; const uint32_t ti = lt ? bi[t] : id;
                bd[t] = lt ? d : bd[t]; bi[t] = lt ? id : bi[t];
                d = td; id = ti;
              }
            }
          }
        }
      }
    const float bound = (float)r * p.h;
    if (bd[KT - 1] <= bound * bound && r > 0) break;
    if (KT == 1 && bd[0] == 0.0f) break;
  }
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    if (k < K) {
      const bool have = bd[k] < 3.0e38f;     // fewer than K points in the cloud: pad like pytorch3d (idx 0, dist 0)
      idx_out[(size_t)i * K + k] = have ? (int64_t)bi[k] : 0;
      dist_out[(size_t)i * K + k] = have ? bd[k] : 0.0f;
    }
  }
}

static int knn_bits(int N) {
  int bits = 4;
  while ((1u << bits) < (uint32_t)N && bits < 26) ++bits;
  return bits;
}

struct KnnWs { KnnParams* prm; SortBufs sort; uint2* buckets; };

static size_t knn_ws_bytes(int N) {
  const size_t n = (size_t)(N > 0 ? N : 1);
  const size_t nb = (n + 2047) / 2048;
  return align_up(sizeof(KnnParams)) + align_up(sizeof(uint32_t) * n) * 4 + align_up(sizeof(uint32_t) * 256 * nb) +
         align_up(sizeof(uint32_t) * 256 * 8) + align_up(sizeof(uint2) * ((size_t)1 << knn_bits(N)));
}

static KnnWs knn_carve(void* ptr, int N) {
  const size_t n = (size_t)(N > 0 ? N : 1);
  const size_t nb = (n + 2047) / 2048;
  char* c = (char*)ptr;
  KnnWs w;
  w.prm = (KnnParams*)c; c += align_up(sizeof(KnnParams));
  for (int i = 0; i < 2; ++i) { w.sort.keys[i] = (uint32_t*)c; c += align_up(sizeof(uint32_t) * n); }
  for (int i = 0; i < 2; ++i) { w.sort.vals[i] = (uint32_t*)c; c += align_up(sizeof(uint32_t) * n); }
  w.sort.hist = (uint32_t*)c; c += align_up(sizeof(uint32_t) * 256 * nb);
  w.sort.digit_total = (uint32_t*)c; c += align_up(sizeof(uint32_t) * 256 * 8);
  w.sort.nb_max = (int)nb;
  w.sort.hist_copies = 1;
  w.buckets = (uint2*)c;
  return w;
}

}  // namespace trase

using namespace trase;

extern "C" {

int trase_knn_sizes(int32_t N, size_t* ws_bytes) {
  if (!ws_bytes || N < 0) { set_error("trase_knn_sizes: bad arguments"); return TRASE_ERR_INVALID; }
  *ws_bytes = knn_ws_bytes(N);
  return TRASE_OK;
}

static int knn_build(const LaunchCtx& c, const float* points, int32_t N, KnnWs& w, uint32_t mask, int bits, int* idx_out) {
  hipStream_t stream = c.stream;
  const int blocks = (N + 255) / 256;
  {
    ProfScope ps("knn_bbox", stream);
    hipLaunchKernelGGL(knn_init_kernel, dim3(1), dim3(1), 0, stream, w.prm, (uint32_t)N);
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, stream, points, N, w.prm);
    hipLaunchKernelGGL(knn_params_kernel, dim3(1), dim3(1), 0, stream, w.prm);
    hipLaunchKernelGGL(knn_keys_kernel, dim3(blocks), dim3(256), 0, stream, points, N, w.prm, mask, w.sort.keys[0]);
  }
  TRASE_POST_LAUNCH("knn_keys", stream, 0);
  int rc = radix_sort_pairs(c, w.sort, &w.prm->n, (uint32_t)N, 0, bits, true, idx_out);
  if (rc) return rc;
  return launch_tile_ranges(c, w.sort.keys[*idx_out], &w.prm->n, (uint32_t)N, w.buckets, 1 << bits);
}

int trase_knn_dist2(const float* points, int32_t N, float* out, void* ws, size_t ws_bytes, int32_t device,
                    trase_stream_t stream_) {
  if (N < 0 || (N > 0 && (!points || !out))) { set_error("trase_knn_dist2: bad arguments"); return TRASE_ERR_INVALID; }
  if (N == 0) return TRASE_OK;
  if (!ws || ws_bytes < knn_ws_bytes(N)) { set_error("trase_knn_dist2: workspace too small"); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  LaunchCtx c{stream, 0, 0};
  KnnWs w = knn_carve(ws, N);
  const int bits = knn_bits(N);
  const uint32_t mask = (1u << bits) - 1u;
  int idx = 0;
  int rc = knn_build(c, points, N, w, mask, bits, &idx);
  if (rc) return rc;
  {
    ProfScope ps("knn_query", stream);
    hipLaunchKernelGGL(knn_query_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, points, N, w.prm, mask, w.sort.vals[idx],
                       w.buckets, out);
  }
  TRASE_POST_LAUNCH("knn_query", stream, 0);
  return TRASE_OK;
}

// pytorch3d.ops.knn_points replacement (call sites scene/gaussian_model.py:88-92 K=16 self-KNN,
// render.py:222 / gui.py:1048 / utils/loss_utils.py:141,192 cross-KNN): the K nearest points of
// p2 for every point of p1, ascending, squared distances; a query that is itself in p2 finds itself.
int trase_knn_points(const float* p1, int32_t N1, const float* p2, int32_t N2, int32_t K, int64_t* idx_out,
                     float* dist_out, void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream_) {
  if (N1 < 0 || N2 < 0 || K < 1 || K > 16) { set_error("trase_knn_points: need 1 <= K <= 16 (got %d)", K); return TRASE_ERR_INVALID; }
  if (N1 == 0) return TRASE_OK;
  if (!p1 || !idx_out || !dist_out || (N2 > 0 && !p2)) { set_error("trase_knn_points: null pointer"); return TRASE_ERR_INVALID; }
  if (!ws || ws_bytes < knn_ws_bytes(N2)) { set_error("trase_knn_points: workspace too small"); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  LaunchCtx c{stream, 0, 0};
  const int blocks = (N1 + 255) / 256;
  if (N2 == 0) {   // nothing to find: pytorch3d pads with idx 0 / dist 0 -- keep that convention
    launch_zero_bytes(idx_out, sizeof(int64_t) * (size_t)N1 * K, stream);
    launch_zero_bytes(dist_out, sizeof(float) * (size_t)N1 * K, stream);
    return TRASE_OK;
  }
  KnnWs w = knn_carve(ws, N2);
  const int bits = knn_bits(N2);
  const uint32_t mask = (1u << bits) - 1u;
  int idx = 0;
  int rc = knn_build(c, p2, N2, w, mask, bits, &idx);
  if (rc) return rc;
  {
    ProfScope ps("knn_points_query", stream);
#define TRASE_KQ(KT) hipLaunchKernelGGL((knn_points_kernel<KT>), dim3(blocks), dim3(256), 0, stream, p1, N1, p2, N2, K, w.prm, mask, w.sort.vals[idx], w.buckets, idx_out, dist_out)
    if (K <= 1) TRASE_KQ(1); else if (K <= 4) TRASE_KQ(4); else if (K <= 8) TRASE_KQ(8); else TRASE_KQ(16);
#undef TRASE_KQ
  }
  TRASE_POST_LAUNCH("knn_points_query", stream, 0);
  return TRASE_OK;
}

}  // extern "C"
