// contrastive.hip -- the FEATURE-state pixel-pair losses, all three opt.contrastive_mode values
// (arguments/__init__.py:131; the tables positive_/negative_pixel_pair_loss at utils/loss_utils.py:396-406), called at
// train.py:290-291 on the S x S matrices C (pixel-mask correspondence, 0/1), C_F (feature similarity) and weights.
//   'soft' (utils/loss_utils.py:304-349, the default):
//     col[j]  = any_i ( C_F[i][j] < th  and C[i][j] == 1 )           (negative: C_F > th and C == 0)
//     pairs   = { (i, j) : i < j, col[j] },  N = |pairs| = sum_j col[j] * j
//     loss    = sum over pairs with C == 1 of  -w * C_F  / N          (negative: C == 0,  +w * relu(C_F) / N)
//   'all'  (utils/loss_utils.py:275-302): the same with col[j] = any_i ( C[i][j] == 1 )   (negative: C == 0)
//   'hard' (utils/loss_utils.py:351-394): sel = { (i, j) : i < j, C_F < th, C == 1 }       (negative: C_F > th, C == 0)
//     loss    = mean over sel of -w * C_F                              (negative: +w * relu(C_F));  N = |sel|
// The reference builds ~10 S x S boolean / float temporaries (S ~ 5000: 25-100 MB each) and synchronises twice on
// torch.nonzero(...).shape[0]; here: one pass for the column flags, one pass for the masked sum (per-block partials,
// reduced in a fixed order), N in closed form on the device, and a dense gradient kernel -- no host sync.
#include "common.h"

namespace trase {

constexpr int CT_ROWS = 64;     // rows per workgroup

__global__ __launch_bounds__(256) void contrastive_flags_kernel(const float* __restrict__ C, const float* __restrict__ CF, int S,
                                                                float th, int kind, int* __restrict__ col) {
  const int negative = kind & 1, all = (kind >> 1) == 1;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * CT_ROWS, i1 = min(i0 + CT_ROWS, S);
  if (j >= S) return;
  bool any = false;
  for (int i = i0; i < i1; ++i) {
    const float c = C[(size_t)i * S + j], f = CF[(size_t)i * S + j];
    any |= negative ? ((all || f > th) && c == 0.0f) : ((all || f < th) && c == 1.0f);
  }
  if (any) atomicOr(&col[j], 1);
}

__global__ __launch_bounds__(256) void contrastive_sum_kernel(const float* __restrict__ C, const float* __restrict__ CF,
                                                              const float* __restrict__ Wt, int S, float th, int kind,
                                                              const int* __restrict__ col, float* __restrict__ partial) {
  __shared__ float red[2][4];
  const int negative = kind & 1, hard = (kind >> 1) == 2;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * CT_ROWS, i1 = min(i0 + CT_ROWS, S);
  float acc = 0.f, cnt = 0.f;                       // cnt: 'hard' selection size (<= 64 per lane, exact in float)
  if (j < S && (hard || col[j])) {
    const int ie = min(i1, j);                      // strictly upper triangle: i < j
    for (int i = i0; i < ie; ++i) {
      const size_t o = (size_t)i * S + j;
      const float c = C[o], f = CF[o];
      const float w = Wt ? Wt[o] : 1.0f;
      if (negative) { if (c == 0.0f && (!hard || f > th)) { acc += w * fmaxf(f, 0.0f); cnt += 1.f; } }
      else if (c == 1.0f && (!hard || f < th)) { acc -= w * f; cnt += 1.f; }
    }
  }
  acc = wave_sum_all(acc);
  cnt = wave_sum_all(cnt);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = acc; red[1][threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    partial[2 * b] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[2 * b + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

__global__ __launch_bounds__(256) void contrastive_final_kernel(const float* __restrict__ partial, int nblocks,
                                                                const int* __restrict__ col, int S, int kind,
                                                                float* __restrict__ out2) {
  __shared__ double sh[2][256];
  const int hard = (kind >> 1) == 2;
  double a = 0.0, n = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) { a += (double)partial[2 * i]; if (hard) n += (double)partial[2 * i + 1]; }
  if (!hard) for (int j = threadIdx.x; j < S; j += 256) n += col[j] ? (double)j : 0.0;
  sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = n;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) { sh[0][threadIdx.x] += sh[0][threadIdx.x + s]; sh[1][threadIdx.x] += sh[1][threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double N = sh[1][0];
    out2[0] = (float)(N > 0.0 ? sh[0][0] / N : 0.0);   // no pair at all: the reference returns 0.0
    out2[1] = (float)N;
  }
}

__global__ __launch_bounds__(256) void contrastive_bwd_kernel(const float* __restrict__ C, const float* __restrict__ CF,
                                                              const float* __restrict__ Wt, int S, float th, int kind,
                                                              const int* __restrict__ col, const float* __restrict__ out2,
                                                              const float* __restrict__ g, float* __restrict__ dCF) {
  const int negative = kind & 1, hard = (kind >> 1) == 2;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * CT_ROWS, i1 = min(i0 + CT_ROWS, S);
  if (j >= S) return;
  const float N = out2[1];
  const float k = (N > 0.f) ? g[0] / N : 0.f;
  const bool cj = hard || col[j] != 0;
  for (int i = i0; i < i1; ++i) {
    const size_t o = (size_t)i * S + j;
    float d = 0.f;
    if (cj && i < j) {
      const float c = C[o], f = CF[o];
      const float w = Wt ? Wt[o] : 1.0f;
      if (negative) { if (c == 0.0f && f > 0.0f && (!hard || f > th)) d = k * w; }
      else if (c == 1.0f && (!hard || f < th)) d = -k * w;
    }
    dCF[o] = d;
  }
}

static size_t contrastive_ws_bytes(int S, int* nblocks) {
  const int bx = (S + 255) / 256, by = (S + CT_ROWS - 1) / CT_ROWS;
  if (nblocks) *nblocks = bx * by;
  return align_up(sizeof(int) * (size_t)S) + align_up(2 * sizeof(float) * (size_t)bx * by);
}

}  // namespace trase

using namespace trase;

extern "C" {

int trase_contrastive_sizes(int32_t S, size_t* ws_bytes) {
  if (!ws_bytes || S < 1) { set_error("trase_contrastive_sizes: bad arguments"); return TRASE_ERR_INVALID; }
  *ws_bytes = contrastive_ws_bytes(S, nullptr);
  return TRASE_OK;
}

int trase_contrastive_forward(const float* C, const float* C_F, const float* weights, int32_t S, float threshold,
                              int32_t kind, float* out2, void* ws, size_t ws_bytes, int32_t device,
                              trase_stream_t stream_) {
  if (!C || !C_F || !out2 || S < 1 || kind < 0 || kind > 5) { set_error("trase_contrastive_forward: bad arguments"); return TRASE_ERR_INVALID; }
  int nblocks = 0;
  if (!ws || ws_bytes < contrastive_ws_bytes(S, &nblocks)) { set_error("trase_contrastive_forward: workspace too small"); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  int* col = (int*)ws;
  float* partial = (float*)((char*)ws + align_up(sizeof(int) * (size_t)S));
  launch_zero_bytes(col, sizeof(int) * (size_t)S, stream);
  const dim3 grid((S + 255) / 256, (S + CT_ROWS - 1) / CT_ROWS);
  {
    ProfScope ps("contrastive_fwd", stream);
    if ((kind >> 1) != 2)
      hipLaunchKernelGGL(contrastive_flags_kernel, grid, dim3(256), 0, stream, C, C_F, S, threshold, kind, col);
    hipLaunchKernelGGL(contrastive_sum_kernel, grid, dim3(256), 0, stream, C, C_F, weights, S, threshold, kind, col, partial);
    hipLaunchKernelGGL(contrastive_final_kernel, dim3(1), dim3(256), 0, stream, partial, nblocks, col, S, kind, out2);
  }
  TRASE_POST_LAUNCH("contrastive_fwd", stream, 0);
  return TRASE_OK;
}

int trase_contrastive_backward(const float* C, const float* C_F, const float* weights, int32_t S, float threshold,
                               int32_t kind, const float* out2, const float* g, const void* ws, size_t ws_bytes, float* dL_dC_F,
                               int32_t device, trase_stream_t stream_) {
  if (!C || !C_F || !out2 || !g || !dL_dC_F || S < 1 || kind < 0 || kind > 5) { set_error("trase_contrastive_backward: bad arguments"); return TRASE_ERR_INVALID; }
  if (!ws || ws_bytes < contrastive_ws_bytes(S, nullptr)) { set_error("trase_contrastive_backward: workspace too small"); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  const dim3 grid((S + 255) / 256, (S + CT_ROWS - 1) / CT_ROWS);
  {
    ProfScope ps("contrastive_bwd", stream);
    hipLaunchKernelGGL(contrastive_bwd_kernel, grid, dim3(256), 0, stream, C, C_F, weights, S, threshold, kind, (const int*)ws, out2,
                       g,
                       dL_dC_F);
  }
  TRASE_POST_LAUNCH("contrastive_bwd", stream, 0);
  return TRASE_OK;
}

}  // extern "C"
