// densify.hip -- densification bookkeeping and the densify / prune stream compaction (second half).
//   per iteration (train.py:362-365, scene/gaussian_model.py:637-639):
//     vis = radii > 0;  max_radii2D[vis] = max(max_radii2D[vis], radii[vis]);
//     xyz_gradient_accum[vis] += |viewspace_grad[vis, :2]|;  denom[vis] += 1
//   every densification_interval iterations (scene/gaussian_model.py:617-635 densify_and_prune, with
//   densify_and_clone :594-615, densify_and_split :563-592, prune_points :491-509 and the optimizer surgery
//   :472-489 / :511-534): the reference runs ~300 boolean-index / cat launches over 7 parameter tensors and their 14
//   Adam moments, in three rounds (clone-append, split-append + prune parents, final prune).  The three rounds compose
//   into ONE row map, so here:  plan (flags + counts) -> scan -> map -> one gather launch over all tensors -> children.
//
//   For an original row i (g = accum/denom, nan -> 0;  smax = max exp(scaling);  o = sigmoid(opacity)):
//     sel = g >= grad_threshold;  clone = sel and smax <= percent_dense*extent;  split = sel and not clone
//     prune(row) = o < min_opacity  or  (max_screen_size given and  smax > 0.1*extent)
//       -- the reference's third term, max_radii2D > max_screen_size, is dead: densification_postfix (:553-555) has
//          zeroed max_radii2D before the final prune reads it.  A clone shares its source's opacity and scaling, hence
//          its prune decision; a split child has scaling' = log(exp(s) / 1.6) and is tested with that.
//   Final row order (what the three rounds of the reference produce):
//     [ originals that are neither split nor pruned | kept clones | kept children, sample 0 | kept children, sample 1 ]
//   Child k (in index order among ALL split-selected rows, M of them) of sample r uses normal draw r*M + k, the order of
//   torch.normal(mean=zeros(2M,3), std=stds.repeat(2,1)) at :572-574.  New rows get zero Adam moments.
#include "common.h"

namespace trase {

constexpr int DN_MAX = 32;            // tensors per gather launch
constexpr int DN_PER_BLOCK = 256 * 4; // 4-byte elements per gather workgroup

enum : uint32_t { DF_KEEP_O = 1u, DF_KEEP_C = 2u, DF_KEEP_S = 4u, DF_SPLIT = 8u };

struct DensifyWs {                    // carved from the caller's workspace
  uint8_t* flags;                     // [P]
  uint32_t* blk;                      // [nblk][4]: counts, then exclusive offsets (keepO, keepC, keepS, split)
  uint32_t* totals;                   // [5]: kept originals, kept clones, kept children per sample, split-selected, clone-selected
  uint32_t* map;                      // [2P]: source row | kind << 30 (0 original, 1 clone, 2 / 3 child of sample 0 / 1)
  uint32_t* aux;                      // [2P]: children: row of the normal draw
};

static size_t densify_ws_carve(int P, DensifyWs* w, void* base) {
  const int nblk = (P + 255) / 256;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes); return o; };
  const size_t o_flags = take((size_t)P), o_blk = take(sizeof(uint32_t) * 4 * (size_t)nblk), o_tot = take(sizeof(uint32_t) * 8),
               o_map = take(sizeof(uint32_t) * 2 * (size_t)P), o_aux = take(sizeof(uint32_t) * 2 * (size_t)P);
  if (w && base) {
    char* b = (char*)base;
    w->flags = (uint8_t*)(b + o_flags); w->blk = (uint32_t*)(b + o_blk); w->totals = (uint32_t*)(b + o_tot);
    w->map = (uint32_t*)(b + o_map); w->aux = (uint32_t*)(b + o_aux);
  }
  return off;
}

__global__ __launch_bounds__(256) void densify_stats_kernel(const float* __restrict__ vgrad, const int32_t* __restrict__ radii,
                                                            float* __restrict__ accum, float* __restrict__ denom,
                                                            float* __restrict__ max_radii, int P,
                                                            const uint32_t* __restrict__ guard) {
  // guard = header of the forward's geom workspace: the statistics of an overflowed (incomplete) view are not taken
  if (guard && (guard[HDR_OVERFLOW] | guard[16] | guard[20])) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const int r = radii[i];
  if (r <= 0) return;                                                  // visibility_filter = radii > 0
  max_radii[i] = fmaxf(max_radii[i], (float)r);
  const float gx = vgrad[(size_t)i * 3], gy = vgrad[(size_t)i * 3 + 1];
  accum[i] += sqrtf(gx * gx + gy * gy);
  denom[i] += 1.0f;
}

__device__ __forceinline__ uint32_t block256_scan_u32(uint32_t v, uint32_t* part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = (uint32_t)__shfl_up((int)x, o);
    if (lane >= o) x += y;
  }
  if (lane == 63) part[wave] = x;
  __syncthreads();
  uint32_t add = 0;
#pragma unroll
  for (int w = 0; w < 3; ++w) add += (w < wave) ? part[w] : 0u;
  __syncthreads();
  return x + add;
}

__global__ __launch_bounds__(256) void densify_plan_kernel(const float* __restrict__ accum, const float* __restrict__ denom,
                                                           const float* __restrict__ scaling, const float* __restrict__ opacity,
                                                           int P, float grad_threshold, float dense_extent, float min_opacity,
                                                           int use_ws, float big_ws, uint8_t* __restrict__ flags,
                                                           uint32_t* __restrict__ blk, uint32_t* __restrict__ totals) {
  __shared__ uint32_t part_a[4], part_b[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  uint32_t f = 0;
  bool is_clone = false;
  if (i < P) {
    float g = accum[i] / denom[i];
    if (g != g) g = 0.0f;                                              // grads[grads.isnan()] = 0.0
    const float s0 = scaling[(size_t)i * 3], s1 = scaling[(size_t)i * 3 + 1], s2 = scaling[(size_t)i * 3 + 2];
    const float e0 = expf(s0), e1 = expf(s1), e2 = expf(s2);
    const float smax = fmaxf(e0, fmaxf(e1, e2));
    const float op = 1.0f / (1.0f + expf(-opacity[i]));
    const bool sel = g >= grad_threshold;
    const bool clone = sel && smax <= dense_extent;
    const bool split = sel && smax > dense_extent;
    const bool prune = op < min_opacity || (use_ws && smax > big_ws);
    // the child's own activated scale, as the reference re-evaluates it: exp(log(exp(s) / (0.8 * 2)))
    const float c0 = expf(logf(e0 / 1.6f)), c1 = expf(logf(e1 / 1.6f)), c2 = expf(logf(e2 / 1.6f));
    const bool prune_child = op < min_opacity || (use_ws && fmaxf(c0, fmaxf(c1, c2)) > big_ws);
    f = ((!split && !prune) ? DF_KEEP_O : 0u) | ((clone && !prune) ? DF_KEEP_C : 0u) | ((split && !prune_child) ? DF_KEEP_S : 0u) |
        (split ? DF_SPLIT : 0u);
    flags[i] = (uint8_t)f;
    is_clone = clone;
  }
  const unsigned long long cb = __ballot(is_clone);                    // the reference's return value num_clone
  if ((threadIdx.x & 63) == 0 && cb) atomicAdd(&totals[4], (uint32_t)__popcll(cb));
  // two 16-bit fields per scan (a workgroup holds at most 256 of each)
  const uint32_t a = block256_scan_u32((f & DF_KEEP_O ? 1u : 0u) | (f & DF_KEEP_C ? 0x10000u : 0u), part_a);
  const uint32_t b = block256_scan_u32((f & DF_KEEP_S ? 1u : 0u) | (f & DF_SPLIT ? 0x10000u : 0u), part_b);
  if (threadIdx.x == 255) {
    blk[4 * blockIdx.x + 0] = a & 0xffffu; blk[4 * blockIdx.x + 1] = a >> 16;
    blk[4 * blockIdx.x + 2] = b & 0xffffu; blk[4 * blockIdx.x + 3] = b >> 16;
  }
}

// one workgroup: exclusive scan of the per-workgroup counts (four columns), totals to ws and to the caller
__global__ __launch_bounds__(256) void densify_scan_kernel(uint32_t* __restrict__ blk, int nblk, uint32_t* __restrict__ totals,
                                                           int32_t* __restrict__ counts) {
  __shared__ uint32_t part[4];
  __shared__ uint32_t carry[4];
  if (threadIdx.x < 4) carry[threadIdx.x] = 0u;
  __syncthreads();
  for (int base = 0; base < nblk; base += 256) {
    const int b = base + threadIdx.x;
    uint32_t v[4], incl[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (b < nblk) ? blk[4 * b + c] : 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) incl[c] = block256_scan_u32(v[c], part);
    if (b < nblk) {
#pragma unroll
      for (int c = 0; c < 4; ++c) blk[4 * b + c] = carry[c] + incl[c] - v[c];
    }
    __syncthreads();
    if (threadIdx.x == 255) {
#pragma unroll
      for (int c = 0; c < 4; ++c) carry[c] += incl[c];
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) { totals[threadIdx.x] = carry[threadIdx.x]; counts[threadIdx.x] = (int32_t)carry[threadIdx.x]; }
  if (threadIdx.x == 4) counts[4] = (int32_t)totals[4];
}

__global__ __launch_bounds__(256) void densify_map_kernel(const uint8_t* __restrict__ flags, const uint32_t* __restrict__ blk,
                                                          const uint32_t* __restrict__ totals, int P,
                                                          uint32_t* __restrict__ map, uint32_t* __restrict__ aux) {
  __shared__ uint32_t part_a[4], part_b[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const uint32_t f = (i < P) ? flags[i] : 0u;
  const uint32_t va = (f & DF_KEEP_O ? 1u : 0u) | (f & DF_KEEP_C ? 0x10000u : 0u);
  const uint32_t vb = (f & DF_KEEP_S ? 1u : 0u) | (f & DF_SPLIT ? 0x10000u : 0u);
  const uint32_t a = block256_scan_u32(va, part_a) - va, b = block256_scan_u32(vb, part_b) - vb;   // exclusive
  if (i >= P) return;
  const uint32_t nO = totals[0], nC = totals[1], nS = totals[2], M = totals[3];
  if (f & DF_KEEP_O) map[blk[4 * blockIdx.x + 0] + (a & 0xffffu)] = (uint32_t)i;
  if (f & DF_KEEP_C) map[nO + blk[4 * blockIdx.x + 1] + (a >> 16)] = (uint32_t)i | (1u << 30);
  if (f & DF_KEEP_S) {
    const uint32_t d0 = nO + nC + blk[4 * blockIdx.x + 2] + (b & 0xffffu);
    const uint32_t k = blk[4 * blockIdx.x + 3] + (b >> 16);            // rank among all split-selected rows
    map[d0] = (uint32_t)i | (2u << 30);      aux[d0] = k;
    map[d0 + nS] = (uint32_t)i | (3u << 30); aux[d0 + nS] = M + k;
  }
}

struct GatherTensors {
  const uint32_t* src[DN_MAX]; uint32_t* dst[DN_MAX];
  int row[DN_MAX];          // 4-byte elements per row
  int zero_new[DN_MAX];     // rows that are not kept originals are written as zeros (Adam moments)
  int first_block[DN_MAX + 1];
  int count;
};

__global__ __launch_bounds__(256) void densify_gather_kernel(GatherTensors t, const uint32_t* __restrict__ map, long long new_rows) {
  int k = 0;
  while (k + 1 < t.count && (int)blockIdx.x >= t.first_block[k + 1]) ++k;      // wave-uniform
  const uint32_t* __restrict__ src = t.src[k];
  uint32_t* __restrict__ dst = t.dst[k];
  const int rf = t.row[k], zn = t.zero_new[k];
  const long long total = new_rows * rf;
  const long long base = (long long)((int)blockIdx.x - t.first_block[k]) * DN_PER_BLOCK + threadIdx.x;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long e = base + j * 256;
    if (e < total) {
      const long long r = e / rf;
      const int c = (int)(e - r * rf);
      const uint32_t m = map[r];
      dst[e] = (zn && (m >> 30)) ? 0u : src[(size_t)(m & 0x3fffffffu) * rf + c];
    }
  }
}

// split children: xyz' = R(rotation) (z * exp(scaling)) + xyz,  scaling' = log(exp(scaling) / 1.6)
// (scene/gaussian_model.py:572-577; R = utils/general_utils.py:122-143 build_rotation of the raw quaternion)
__global__ __launch_bounds__(256) void densify_children_kernel(const uint32_t* __restrict__ map, const uint32_t* __restrict__ aux,
                                                               const uint32_t* __restrict__ totals,
                                                               const float* __restrict__ xyz, const float* __restrict__ scaling,
                                                               const float* __restrict__ rotation,
                                                               const float* __restrict__ normal, float* __restrict__ new_xyz,
                                                               float* __restrict__ new_scaling) {
  const uint32_t first = totals[0] + totals[1], n = 2u * totals[2];
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t d = first + j;
  const size_t s = map[d] & 0x3fffffffu, z = aux[d];
  const float q0 = rotation[s * 4], q1 = rotation[s * 4 + 1], q2 = rotation[s * 4 + 2], q3 = rotation[s * 4 + 3];
  const float norm = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  const float r = q0 / norm, x = q1 / norm, y = q2 / norm, zz = q3 / norm;
  const float e0 = expf(scaling[s * 3]), e1 = expf(scaling[s * 3 + 1]), e2 = expf(scaling[s * 3 + 2]);
  const float a0 = normal[z * 3] * e0, a1 = normal[z * 3 + 1] * e1, a2 = normal[z * 3 + 2] * e2;
  const float R00 = 1.f - 2.f * (y * y + zz * zz), R01 = 2.f * (x * y - r * zz), R02 = 2.f * (x * zz + r * y);
  const float R10 = 2.f * (x * y + r * zz), R11 = 1.f - 2.f * (x * x + zz * zz), R12 = 2.f * (y * zz - r * x);
  const float R20 = 2.f * (x * zz - r * y), R21 = 2.f * (y * zz + r * x), R22 = 1.f - 2.f * (x * x + y * y);
  new_xyz[(size_t)d * 3] = (R00 * a0 + R01 * a1 + R02 * a2) + xyz[s * 3];
  new_xyz[(size_t)d * 3 + 1] = (R10 * a0 + R11 * a1 + R12 * a2) + xyz[s * 3 + 1];
  new_xyz[(size_t)d * 3 + 2] = (R20 * a0 + R21 * a1 + R22 * a2) + xyz[s * 3 + 2];
  new_scaling[(size_t)d * 3] = logf(e0 / 1.6f);
  new_scaling[(size_t)d * 3 + 1] = logf(e1 / 1.6f);
  new_scaling[(size_t)d * 3 + 2] = logf(e2 / 1.6f);
}

}  // namespace trase

using namespace trase;

extern "C" {

int trase_densify_stats(const float* viewspace_grad, const int32_t* radii, float* xyz_gradient_accum, float* denom,
                        float* max_radii2D, int32_t P, int32_t device, trase_stream_t stream_) {
  return trase_densify_stats_guarded(viewspace_grad, radii, xyz_gradient_accum, denom, max_radii2D, P, nullptr, device, stream_);
}

int trase_densify_stats_guarded(const float* viewspace_grad, const int32_t* radii, float* xyz_gradient_accum, float* denom,
                                float* max_radii2D, int32_t P, const void* guard, int32_t device, trase_stream_t stream_) {
  if (P < 0 || (P > 0 && (!viewspace_grad || !radii || !xyz_gradient_accum || !denom || !max_radii2D))) {
    set_error("trase_densify_stats: bad arguments"); return TRASE_ERR_INVALID;
  }
  if (P == 0) return TRASE_OK;
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  {
    ProfScope ps("densify_stats", stream);
    hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, viewspace_grad, radii,
                       xyz_gradient_accum, denom, max_radii2D, P, (const uint32_t*)guard);
  }
  TRASE_POST_LAUNCH("densify_stats", stream, 0);
  return TRASE_OK;
}

int trase_densify_sizes(int32_t P, size_t* ws_bytes) {
  if (!ws_bytes || P < 1 || P >= (1 << 29)) { set_error("trase_densify_sizes: bad arguments"); return TRASE_ERR_INVALID; }
  *ws_bytes = densify_ws_carve(P, nullptr, nullptr);
  return TRASE_OK;
}

int trase_densify_plan(const float* xyz_gradient_accum, const float* denom, const float* scaling, const float* opacity,
                       int32_t P, float grad_threshold, float dense_extent, float min_opacity, int32_t use_screen_size,
                       float big_extent, int32_t* counts, void* ws, size_t ws_bytes, int32_t device,
                       trase_stream_t stream_) {
  if (!xyz_gradient_accum || !denom || !scaling || !opacity || !counts || P < 1 || P >= (1 << 29)) {
    set_error("trase_densify_plan: bad arguments"); return TRASE_ERR_INVALID;
  }
  DensifyWs w;
  if (!ws || ws_bytes < densify_ws_carve(P, &w, ws)) { set_error("trase_densify_plan: workspace too small"); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  const int nblk = (P + 255) / 256;
  launch_zero_bytes(w.totals, sizeof(uint32_t) * 8, stream);
  {
    ProfScope ps("densify_plan", stream);
    hipLaunchKernelGGL(densify_plan_kernel, dim3(nblk), dim3(256), 0, stream, xyz_gradient_accum, denom, scaling, opacity, P,
                       grad_threshold, dense_extent, min_opacity, use_screen_size, big_extent, w.flags, w.blk, w.totals);
    hipLaunchKernelGGL(densify_scan_kernel, dim3(1), dim3(256), 0, stream, w.blk, nblk, w.totals, counts);
    hipLaunchKernelGGL(densify_map_kernel, dim3(nblk), dim3(256), 0, stream, w.flags, w.blk, w.totals, P, w.map, w.aux);
  }
  TRASE_POST_LAUNCH("densify_plan", stream, 0);
  return TRASE_OK;
}

int trase_densify_apply(int32_t count, const void* const* src, void* const* dst, const int32_t* row_elems,
                        const int32_t* zero_new, int32_t P, int32_t new_P, const float* xyz, const float* scaling,
                        const float* rotation, const float* normal_samples, float* new_xyz, float* new_scaling,
                        const void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream_) {
  if (count < 0 || count > DN_MAX || P < 1 || new_P < 0 || new_P > 2 * (int64_t)P) {
    set_error("trase_densify_apply: bad arguments (at most %d tensors per call)", DN_MAX); return TRASE_ERR_INVALID;
  }
  if (new_P == 0) return TRASE_OK;
  if ((count > 0 && (!src || !dst || !row_elems || !zero_new)) || !xyz || !scaling || !rotation || !new_xyz || !new_scaling) {
    set_error("trase_densify_apply: null pointer"); return TRASE_ERR_INVALID;
  }
  DensifyWs w;
  if (!ws || ws_bytes < densify_ws_carve(P, &w, const_cast<void*>(ws))) { set_error("trase_densify_apply: workspace too small"); return TRASE_ERR_WORKSPACE; }
  GatherTensors t;
  int blocks = 0, k = 0;
  for (int i = 0; i < count; ++i) {
    if (row_elems[i] < 1 || !src[i] || !dst[i]) { set_error("trase_densify_apply: tensor %d: null pointer or empty row", i); return TRASE_ERR_INVALID; }
    t.src[k] = (const uint32_t*)src[i]; t.dst[k] = (uint32_t*)dst[i]; t.row[k] = row_elems[i]; t.zero_new[k] = zero_new[i];
    t.first_block[k] = blocks;
    blocks += (int)(((long long)new_P * row_elems[i] + DN_PER_BLOCK - 1) / DN_PER_BLOCK);
    ++k;
  }
  t.count = k; t.first_block[k] = blocks;
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  {
    ProfScope ps("densify_apply", stream);
    if (k > 0) hipLaunchKernelGGL(densify_gather_kernel, dim3(blocks), dim3(256), 0, stream, t, w.map, (long long)new_P);
    // the children are at most new_P rows; the kernel reads their exact range from the totals
    if (normal_samples)
      hipLaunchKernelGGL(densify_children_kernel, dim3((new_P + 255) / 256), dim3(256), 0, stream, w.map, w.aux, w.totals, xyz,
                         scaling, rotation, normal_samples, new_xyz, new_scaling);
  }
  TRASE_POST_LAUNCH("densify_apply", stream, 0);
  return TRASE_OK;
}

}  // extern "C"
