// smooth.hip -- KNN feature smoothing of the FEATURE state:
// GaussianModel.get_smoothed_gaussian_features (scene/gaussian_model.py:79-104, called from
// gaussian_renderer/__init__.py:118 when opt.smooth_K != 1, train.py:274-275):
//     out[i] = mean over the selected neighbour slots s of  normalize(features[idx[i][sel[s]]])
// (F.normalize: x / max(||x||, 1e-12)).  The reference materialises the (P, S, 32) gather and back-propagates with
// an atomic index_put; here the forward is one gather kernel and the backward a gather over the REVERSE
// adjacency (built once per KNN map by the host), so it needs no atomics and is bit-reproducible.
// Eight lanes own one Gaussian (a float4 of its 32-float row each): a row access is one coalesced 128-byte read.
#include "common.h"

namespace trase {

constexpr int SM_F = 32;          // feature width (gaussian_features_dim, scene/gaussian_model.py:63)
constexpr float SM_EPS = 1e-12f;  // torch.nn.functional.normalize default eps

__device__ __forceinline__ float group8_sum(float v) {   // all-reduce inside an aligned group of 8 lanes
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}

__global__ __launch_bounds__(256) void smooth_inv_norm_kernel(const float* __restrict__ feat, int P, float* __restrict__ inv_norm) {
  const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, t = threadIdx.x & 7;
  const int j = g < P ? g : P - 1;
  const float4 x = *reinterpret_cast<const float4*>(feat + (size_t)j * SM_F + 4 * t);
  const float n2 = group8_sum((x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w));
  if (g < P && t == 0) inv_norm[j] = 1.0f / fmaxf(sqrtf(n2), SM_EPS);
}

__global__ __launch_bounds__(256) void smooth_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ inv_norm,
                                                         const int64_t* __restrict__ idx, int P, int K,
                                                         const int32_t* __restrict__ sel, int S, float* __restrict__ out) {
  const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, t = threadIdx.x & 7;
  if (g >= P) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < S; ++s) {
    const int64_t j = idx[(size_t)g * K + sel[s]];
    const float w = inv_norm[j];
    const float4 x = *reinterpret_cast<const float4*>(feat + (size_t)j * SM_F + 4 * t);
    acc.x = fmaf(x.x, w, acc.x); acc.y = fmaf(x.y, w, acc.y); acc.z = fmaf(x.z, w, acc.z); acc.w = fmaf(x.w, w, acc.w);
  }
  const float r = 1.0f / (float)S;
  *reinterpret_cast<float4*>(out + (size_t)g * SM_F + 4 * t) = make_float4(acc.x * r, acc.y * r, acc.z * r, acc.w * r);
}

// dL/dfeatures[j] = J_j^T G_j,  G_j = (1/S) sum of dL/dout[i] over the selected slots (i, k) that point at j,
// J = d normalize(x)/dx = (I - n n^T) / max(||x||, eps)  (for ||x|| below eps the normalisation is linear: I / eps)
__global__ __launch_bounds__(256) void smooth_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ inv_norm,
                                                         const int32_t* __restrict__ rev_ptr,
                                                         const int32_t* __restrict__ rev_src, int P, int K, uint32_t sel_mask,
                                                         int S, const float* __restrict__ g_out, float* __restrict__ g_feat) {
  const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, t = threadIdx.x & 7;
  if (g >= P) return;
  float4 G = make_float4(0.f, 0.f, 0.f, 0.f);
  const int e1 = rev_ptr[g + 1];
  for (int e = rev_ptr[g]; e < e1; e += 4) {      // four edges at a time: their two-level loads (edge -> row) overlap
    int src[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) src[u] = (e + u < e1) ? rev_src[e + u] : -1;   // i * K + k, ascending: fixed summation order
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = src[u] >= 0 ? src[u] / K : 0, k = src[u] >= 0 ? src[u] - i * K : 0;
      const bool on = src[u] >= 0 && ((sel_mask >> k) & 1u);
      v[u] = on ? *reinterpret_cast<const float4*>(g_out + (size_t)i * SM_F + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (!on) src[u] = -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (src[u] >= 0) { G.x += v[u].x; G.y += v[u].y; G.z += v[u].z; G.w += v[u].w; }
  }
  const float r = 1.0f / (float)S;
  G.x *= r; G.y *= r; G.z *= r; G.w *= r;
  const float4 x = *reinterpret_cast<const float4*>(feat + (size_t)g * SM_F + 4 * t);
  const float w = inv_norm[g];
  float4 d;
  if (w < 1.0f / SM_EPS) {                        // the usual case: ||x|| > eps
    const float dot = group8_sum((x.x * G.x + x.y * G.y) + (x.z * G.z + x.w * G.w)) * w * w;   // (n . G) / ||x||
    d = make_float4((G.x - x.x * dot) * w, (G.y - x.y * dot) * w, (G.z - x.z * dot) * w, (G.w - x.w * dot) * w);
  } else {
    d = make_float4(G.x * w, G.y * w, G.z * w, G.w * w);
  }
  *reinterpret_cast<float4*>(g_feat + (size_t)g * SM_F + 4 * t) = d;
}

}  // namespace trase

using namespace trase;

extern "C" {

static int smooth_check(const char* who, int32_t P, int32_t F, int32_t K, int32_t S) {
  if (P < 0 || K < 1 || K > 32 || S < 1 || S > K) { set_error("%s: bad sizes (P %d, K %d, S %d; K <= 32, 1 <= S <= K)", who, P, K, S); return TRASE_ERR_INVALID; }
  if (F != SM_F) { set_error("%s: feature width %d not compiled in (32)", who, F); return TRASE_ERR_UNSUPPORTED; }
  return TRASE_OK;
}

int trase_smooth_forward(const float* features, const int64_t* knn_idx, int32_t P, int32_t F, int32_t K, const int32_t* select,
                         int32_t S, float* inv_norm, float* out, int32_t device, trase_stream_t stream_) {
  if (int rc = smooth_check("trase_smooth_forward", P, F, K, S)) return rc;
  if (P == 0) return TRASE_OK;
  if (!features || !knn_idx || !select || !inv_norm || !out) { set_error("trase_smooth_forward: null pointer"); return TRASE_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  const dim3 grid((unsigned)(((size_t)P * 8 + 255) / 256)), block(256);
  {
    ProfScope ps("smooth_inv_norm", stream);
    hipLaunchKernelGGL(smooth_inv_norm_kernel, grid, block, 0, stream, features, P, inv_norm);
  }
  TRASE_POST_LAUNCH("smooth_inv_norm", stream, 0);
  {
    ProfScope ps("smooth_fwd", stream);
    hipLaunchKernelGGL(smooth_fwd_kernel, grid, block, 0, stream, features, inv_norm, knn_idx, P, K, select, S, out);
  }
  TRASE_POST_LAUNCH("smooth_fwd", stream, 0);
  return TRASE_OK;
}

int trase_smooth_backward(const float* features, const float* inv_norm, int32_t P, int32_t F, int32_t K, uint32_t select_mask,
                          int32_t S, const int32_t* rev_ptr, const int32_t* rev_src, const float* dL_dout,
                          float* dL_dfeatures, int32_t device, trase_stream_t stream_) {
  if (int rc = smooth_check("trase_smooth_backward", P, F, K, S)) return rc;
  if (P == 0) return TRASE_OK;
  if (!features || !inv_norm || !rev_ptr || !rev_src || !dL_dout || !dL_dfeatures) { set_error("trase_smooth_backward: null pointer"); return TRASE_ERR_INVALID; }
  if (__builtin_popcount(select_mask) != S) { set_error("trase_smooth_backward: select_mask has %d bits, S = %d", __builtin_popcount(select_mask), S); return TRASE_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  const dim3 grid((unsigned)(((size_t)P * 8 + 255) / 256)), block(256);
  {
    ProfScope ps("smooth_bwd", stream);
    hipLaunchKernelGGL(smooth_bwd_kernel, grid, block, 0, stream, features, inv_norm, rev_ptr, rev_src, P, K, select_mask, S,
                       dL_dout, dL_dfeatures);
  }
  TRASE_POST_LAUNCH("smooth_bwd", stream, 0);
  return TRASE_OK;
}

}  // extern "C"
