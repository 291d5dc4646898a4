// optim.hip -- multi-tensor Adam step (first half): the reference steps
// torch.optim.Adam(l, lr=0.0, eps=1e-15) over six (GAUSSIAN) + one (FEATURE) parameter groups with per-group
// learning rates (scene/gaussian_model.py:253-300; train.py:376-389).  One launch updates every tensor of a group
// list: p, exp_avg, exp_avg_sq in place, the same arithmetic as torch's single-tensor path
//   m <- lerp(m, g, 1-b1);  v <- b2 v + (1-b2) g g;  p <- p - (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps).
#include "common.h"

namespace trase {

constexpr int AD_MAX = 16;
struct AdamTensors {
  float* p[AD_MAX]; const float* g[AD_MAX]; float* m[AD_MAX]; float* v[AD_MAX];
  long long n[AD_MAX]; float lr[AD_MAX]; float bc1[AD_MAX]; float bc2_sqrt[AD_MAX];
  int first_block[AD_MAX + 1];
  int count;
};
constexpr int AD_PER_BLOCK = 256 * 4;

// guard: the 64-word header of a forward's geom workspace (or null).  A forward whose pair buffer overflowed, or whose
// binning guards tripped, rendered incomplete images and gradients: the step that would consume them is skipped ON THE
// DEVICE -- parameters and both moments stay bit-identical -- like the reference skips its optimizer step on a bad
// iteration (train.py:298-301, :378), without the host ever waiting for the flag.
__global__ __launch_bounds__(256) void adam_kernel(AdamTensors t, float w1, float beta2, float w2, float eps,
                                                   const uint32_t* __restrict__ guard) {
  if (guard && (guard[HDR_OVERFLOW] | guard[16] | guard[20])) return;
  int k = 0;
  while (k + 1 < t.count && (int)blockIdx.x >= t.first_block[k + 1]) ++k;      // wave-uniform, <= 16 steps
  const long long base = (long long)((int)blockIdx.x - t.first_block[k]) * AD_PER_BLOCK + threadIdx.x * 4;
  float* __restrict__ p = t.p[k];
  const float* __restrict__ g = t.g[k];
  float* __restrict__ m = t.m[k];
  float* __restrict__ v = t.v[k];
  const long long n = t.n[k];
  const float step = t.lr[k] / t.bc1[k], bc2s = t.bc2_sqrt[k];
  auto upd = [&](float pi, float gi, float& mi, float& vi) -> float {
    mi = mi + (gi - mi) * w1;                             // torch: exp_avg.lerp_(grad, 1 - beta1)
    vi = vi * beta2 + w2 * gi * gi;                       // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(vi) / bc2s + eps;
    return pi - step * (mi / denom);
  };
  if (base + 3 < n && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0)) {
    float4 P = *reinterpret_cast<float4*>(p + base), M = *reinterpret_cast<float4*>(m + base), V = *reinterpret_cast<float4*>(v + base);
    const float4 G = *reinterpret_cast<const float4*>(g + base);
    P.x = upd(P.x, G.x, M.x, V.x); P.y = upd(P.y, G.y, M.y, V.y); P.z = upd(P.z, G.z, M.z, V.z); P.w = upd(P.w, G.w, M.w, V.w);
    *reinterpret_cast<float4*>(p + base) = P; *reinterpret_cast<float4*>(m + base) = M; *reinterpret_cast<float4*>(v + base) = V;
  } else {
    for (int e = 0; e < 4; ++e) {
      const long long i = base + e;
      if (i < n) { float mi = m[i], vi = v[i]; p[i] = upd(p[i], g[i], mi, vi); m[i] = mi; v[i] = vi; }
    }
  }
}

}  // namespace trase

using namespace trase;

extern "C" {

int trase_adam_step(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const int64_t* numel, const float* lr, const int64_t* step, double beta1,
                    double beta2, float eps, int32_t device, trase_stream_t stream_) {
  return trase_adam_step_guarded(count, params, grads, exp_avg, exp_avg_sq, numel, lr, step, beta1, beta2, eps, nullptr, device, stream_);
}

int trase_adam_step_guarded(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                            float* const* exp_avg_sq, const int64_t* numel, const float* lr, const int64_t* step, double beta1,
                            double beta2, float eps, const void* guard, int32_t device, trase_stream_t stream_) {
  if (count < 0 || count > AD_MAX) { set_error("trase_adam_step: %d tensors (max %d per call)", count, AD_MAX); return TRASE_ERR_INVALID; }
  if (count == 0) return TRASE_OK;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr || !step) { set_error("trase_adam_step: null table"); return TRASE_ERR_INVALID; }
  AdamTensors t;
  int blocks = 0, k = 0;
  for (int i = 0; i < count; ++i) {
    if (numel[i] == 0) continue;
    if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] < 0 || step[i] < 1) {
      set_error("trase_adam_step: tensor %d: null pointer, negative size or step < 1", i); return TRASE_ERR_INVALID;
    }
    t.p[k] = params[i]; t.g[k] = grads[i]; t.m[k] = exp_avg[i]; t.v[k] = exp_avg_sq[i]; t.n[k] = numel[i]; t.lr[k] = lr[i];
    t.bc1[k] = (float)(1.0 - pow(beta1, (double)step[i]));
    t.bc2_sqrt[k] = (float)sqrt(1.0 - pow(beta2, (double)step[i]));
    t.first_block[k] = blocks;
    blocks += (int)((numel[i] + AD_PER_BLOCK - 1) / AD_PER_BLOCK);
    ++k;
  }
  t.count = k; t.first_block[k] = blocks;
  if (k == 0) return TRASE_OK;
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  {
    ProfScope ps("adam", stream);
    // 1 - beta in double, then rounded (as Python does for torch): 1.0f - 0.999f would lose five digits
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, t, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), eps, (const uint32_t*)guard);
  }
  TRASE_POST_LAUNCH("adam", stream, 0);
  return TRASE_OK;
}

}  // extern "C"
