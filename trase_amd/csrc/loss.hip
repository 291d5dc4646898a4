// loss.hip -- photometric loss heads of the training loop:
//   l1_loss (utils/loss_utils.py:30-31) and ssim (:56-86, 11x11 Gaussian window sigma 1.5, zero padding,
//   C1 = 0.01^2, C2 = 0.03^2), combined at train.py:235-238.
// The reference runs five depthwise 11x11 convolutions forward and their transposes backward through autograd.
// Here the window is applied separably from LDS tiles:
//   forward : one pass computes mu1, mu2, E[x^2], E[y^2], E[xy] for a 32x32 tile, the SSIM map value and the
//             three partial derivatives dS/dmu1, dS/dE[x^2], dS/dE[xy], which it stores for the backward; |x-y| and
//             S are summed per block into a partial array that a one-block kernel reduces (deterministic order).
//   backward: dL/dx = w_s * (blur(dS/dmu1) + 2 x blur(dS/dE[x^2]) + y blur(dS/dE[xy])) + w_l * sign(x - y)
//             (the window is symmetric and the padding is zero, so the transposed convolution is the same blur).
#include "common.h"

namespace trase {

constexpr int LW = 11, LR = 5;            // window, radius
constexpr int LT = 32;                    // tile edge
constexpr int LH = LT + 2 * LR;           // 42: tile + halo

struct LossWin { float g[LW]; };

__device__ __forceinline__ float tile_load(const float* __restrict__ p, int H, int W, int y, int x) {
  return (x >= 0 && x < W && y >= 0 && y < H) ? p[(size_t)y * W + x] : 0.f;    // conv2d zero padding
}

// Round 5: a thread owns FOUR consecutive outputs along the filtered direction and slides the window over the 14 inputs they
// share -- 28 + 70 LDS reads per four pixels instead of 88 + 220 -- with every output's eleven fused multiply-adds in the same
// order as before (k ascending).  Measured (profiles/r5_ab_experiments.txt): forward 0.075 -> 0.072 ms, backward unchanged at
// 0.062 -- the tap reads were NOT what bounds these kernels (the halo loads and their bounds checks are next).  Pitch 40 of the
// intermediate rows: the two row groups of a wave (4 rows apart) land 32 banks apart.
constexpr int HP = 40;

__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float* __restrict__ img, const float* __restrict__ gt, int H, int W,
                                                       LossWin win, float* __restrict__ dmaps /* [3][C][H][W] */,
                                                       float* __restrict__ partial /* [blocks][2] */) {
  __shared__ float sx[LH][LH + 1], sy[LH][LH + 1];
  __shared__ float hb[5][LH][HP];
  __shared__ float red[2][4];
  const int c = blockIdx.z;
  const size_t plane = (size_t)H * W;
  const float* X = img + c * plane;
  const float* Y = gt + c * plane;
  const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
  {
    // the tile + halo in registers first: all of a thread's loads in flight at once (as a loop of load -> LDS store the
    // seven round trips per thread were most of this kernel's run time at three workgroups per CU)
    constexpr int NL = (LH * LH + 255) / 256;
    float vx[NL], vy[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = threadIdx.x + 256 * k, r = i / LH, q = i % LH;
      const bool in = i < LH * LH;
      vx[k] = in ? tile_load(X, H, W, y0 + r - LR, x0 + q - LR) : 0.f;
      vy[k] = in ? tile_load(Y, H, W, y0 + r - LR, x0 + q - LR) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = threadIdx.x + 256 * k, r = i / LH, q = i % LH;
      if (i < LH * LH) { sx[r][q] = vx[k]; sy[r][q] = vy[k]; }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < LH * (LT / 4); i += 256) {   // horizontal pass: 42 rows x 8 groups of four columns
    const int r = i >> 3, q0 = (i & 7) * 4;
    float xs[LW + 3], ys[LW + 3];
#pragma unroll
    for (int t = 0; t < LW + 3; ++t) { xs[t] = sx[r][q0 + t]; ys[t] = sy[r][q0 + t]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = 0.f, b = 0.f, p = 0.f, qq = 0.f, rr = 0.f;
#pragma unroll
      for (int k = 0; k < LW; ++k) {
        const float w = win.g[k], x = xs[j + k], y = ys[j + k];
        a = fmaf(w, x, a); b = fmaf(w, y, b); p = fmaf(w, x * x, p); qq = fmaf(w, y * y, qq); rr = fmaf(w, x * y, rr);
      }
      hb[0][r][q0 + j] = a; hb[1][r][q0 + j] = b; hb[2][r][q0 + j] = p; hb[3][r][q0 + j] = qq; hb[4][r][q0 + j] = rr;
    }
  }
  __syncthreads();
  float s_l1 = 0.f, s_ss = 0.f;
  const size_t cp = (size_t)gridDim.z * plane;
  {                                                           // vertical pass + SSIM: column q, rows r0 .. r0 + 3
    const int q = threadIdx.x & 31, r0 = (threadIdx.x >> 5) * 4;
    float acc[4][5];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int m = 0; m < 5; ++m) acc[j][m] = 0.f;
#pragma unroll
    for (int t = 0; t < LW + 3; ++t) {
      float v[5];
#pragma unroll
      for (int m = 0; m < 5; ++m) v[m] = hb[m][r0 + t][q];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = t - j;
        if (k >= 0 && k < LW) {
#pragma unroll
          for (int m = 0; m < 5; ++m) acc[j][m] = fmaf(win.g[k], v[m], acc[j][m]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = r0 + j, y = y0 + r, x = x0 + q;
      if (y >= H || x >= W) continue;
      const float a = acc[j][0], b = acc[j][1], p = acc[j][2], qq = acc[j][3], rr = acc[j][4];
      constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
      const float A1 = 2.f * a * b + C1, A2 = 2.f * (rr - a * b) + C2;
      const float B1 = a * a + b * b + C1, B2 = (p - a * a) + (qq - b * b) + C2;
      const float iB1 = 1.f / B1, iB2 = 1.f / B2;
      const float S = A1 * A2 * iB1 * iB2;
      const size_t o = c * plane + (size_t)y * W + x;
      dmaps[o] = 2.f * b * (A2 - A1) * iB1 * iB2 - 2.f * a * S * (iB1 - iB2);   // dS/dmu1
      dmaps[cp + o] = -S * iB2;                                                   // dS/dE[x^2]
      dmaps[2 * cp + o] = 2.f * A1 * iB1 * iB2;                                   // dS/dE[xy]
      s_ss += S;
      s_l1 += fabsf(sx[r + LR][q + LR] - sy[r + LR][q + LR]);
    }
  }
  // block reduction in a fixed order
  s_l1 = wave_sum_all(s_l1); s_ss = wave_sum_all(s_ss);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s_l1; red[1][threadIdx.x >> 6] = s_ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partial[2 * blk] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[2 * blk + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// (1 - lambda) l1 + lambda (1 - ssim) with every operation rounded on its own (HIP's default -ffp-contract=fast fuses a * b + c,
// also through __fmul_rn / __fadd_rn, which are plain operators there)
__device__ __noinline__ float combine_rounded(float w1, float l1, float w2, float ss) {
#pragma clang fp contract(off)
  const float p1 = w1 * l1;
  const float om = 1.0f - ss;
  const float p2 = w2 * om;
  return p1 + p2;
}

// lambda_dssim >= 0: out2 has a third slot that receives train.py:235-238's combination (1 - lambda) l1 + lambda (1 - ssim)
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ partial, int nblocks, float inv_count,
                                                          float* __restrict__ out2, float lambda_dssim, float one_minus_lambda) {
  __shared__ double sh[2][256];
  double a = 0.0, b = 0.0;
  for (int i0 = threadIdx.x; i0 < nblocks; i0 += 8 * 256) {   // eight pairs in flight; same summation order as one by one
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 256 * u;
      v[u] = (i < nblocks) ? *reinterpret_cast<const float2*>(partial + 2 * (size_t)i) : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { a += (double)v[u].x; b += (double)v[u].y; }
  }
  sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) { sh[0][threadIdx.x] += sh[0][threadIdx.x + s]; sh[1][threadIdx.x] += sh[1][threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float l1 = (float)(sh[0][0] * inv_count), ss = (float)(sh[1][0] * inv_count);
    out2[0] = l1; out2[1] = ss;
    // the two products and the sum individually rounded, as torch's tensor arithmetic forms train.py:235-238 (a contracted fma
    // differs in the last bit: the combination is asserted bit-identical to the composition around l1_ssim)
    if (lambda_dssim >= 0.f) out2[2] = combine_rounded(one_minus_lambda, l1, lambda_dssim, ss);
  }
}

__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float* __restrict__ img, const float* __restrict__ gt, int H, int W,
                                                       LossWin win, const float* __restrict__ dmaps,
                                                       const float* __restrict__ g2 /* dL/dl1, dL/dssim */, float inv_count,
                                                       float* __restrict__ d_img, int g_stride, float s_l1, float s_ssim) {
  // (g_stride 1, scales 1: the two cotangents as given; g_stride 0: ONE cotangent dL/dloss of the combined loss, scaled by
  // d loss / d l1 = 1 - lambda and d loss / d ssim = -lambda -- the products autograd's scalar chain would have formed)
  __shared__ float sd[3][LH][LH + 1];
  __shared__ float hb[3][LH][HP];
  const int c = blockIdx.z;
  const size_t plane = (size_t)H * W, cp = (size_t)gridDim.z * plane;
  const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
  {
    constexpr int NL = (LH * LH + 255) / 256;            // every load of the thread in flight at once (see ssim_fwd_kernel)
    float v[NL][3];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = threadIdx.x + 256 * k, r = i / LH, q = i % LH;
      const bool in = i < LH * LH;
#pragma unroll
      for (int m = 0; m < 3; ++m) v[k][m] = in ? tile_load(dmaps + m * cp + c * plane, H, W, y0 + r - LR, x0 + q - LR) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = threadIdx.x + 256 * k, r = i / LH, q = i % LH;
      if (i < LH * LH) {
#pragma unroll
        for (int m = 0; m < 3; ++m) sd[m][r][q] = v[k][m];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < LH * (LT / 4); i += 256) {   // horizontal pass, four columns per thread (see ssim_fwd_kernel)
    const int r = i >> 3, q0 = (i & 7) * 4;
    float v[3][LW + 3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int t = 0; t < LW + 3; ++t) v[m][t] = sd[m][r][q0 + t];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = 0.f, p = 0.f, rr = 0.f;
#pragma unroll
      for (int k = 0; k < LW; ++k) {
        const float w = win.g[k];
        a = fmaf(w, v[0][j + k], a); p = fmaf(w, v[1][j + k], p); rr = fmaf(w, v[2][j + k], rr);
      }
      hb[0][r][q0 + j] = a; hb[1][r][q0 + j] = p; hb[2][r][q0 + j] = rr;
    }
  }
  __syncthreads();
  const float wl = (g2[0] * s_l1) * inv_count, ws = (g2[g_stride] * s_ssim) * inv_count;
  {
    const int q = threadIdx.x & 31, r0 = (threadIdx.x >> 5) * 4;
    float acc[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int m = 0; m < 3; ++m) acc[j][m] = 0.f;
#pragma unroll
    for (int t = 0; t < LW + 3; ++t) {
      float v[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) v[m] = hb[m][r0 + t][q];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = t - j;
        if (k >= 0 && k < LW) {
#pragma unroll
          for (int m = 0; m < 3; ++m) acc[j][m] = fmaf(win.g[k], v[m], acc[j][m]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = y0 + r0 + j, x = x0 + q;
      if (y >= H || x >= W) continue;
      const size_t o = c * plane + (size_t)y * W + x;
      const float xv = img[o], yv = gt[o];
      const float d = xv - yv;
      const float sgn = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
      d_img[o] = ws * (acc[j][0] + 2.f * xv * acc[j][1] + yv * acc[j][2]) + wl * sgn;
    }
  }
}

static LossWin make_window() {
  // gaussian(11, 1.5) of utils/loss_utils.py:46-48: Python floats (double) -> float32 tensor -> divided by its float32 sum
  LossWin w;
  float g[LW];
  float sum = 0.f;
  for (int x = 0; x < LW; ++x) { g[x] = (float)exp(-(double)((x - LW / 2) * (x - LW / 2)) / (2.0 * 1.5 * 1.5)); }
  for (int x = 0; x < LW; ++x) sum += g[x];
  for (int x = 0; x < LW; ++x) w.g[x] = g[x] / sum;
  return w;
}

static size_t loss_ws_bytes(int C, int H, int W, int* nblocks) {
  const int bx = (W + LT - 1) / LT, by = (H + LT - 1) / LT;
  if (nblocks) *nblocks = bx * by * C;
  return align_up(sizeof(float) * 3 * (size_t)C * H * W) + align_up(sizeof(float) * 2 * (size_t)bx * by * C);
}

}  // namespace trase

using namespace trase;

extern "C" {

int trase_loss_sizes(int32_t C, int32_t H, int32_t W, size_t* ws_bytes) {
  if (!ws_bytes || C < 1 || H < 1 || W < 1) { set_error("trase_loss_sizes: bad arguments"); return TRASE_ERR_INVALID; }
  *ws_bytes = loss_ws_bytes(C, H, W, nullptr);
  return TRASE_OK;
}

static int loss_forward(const char* who, const float* img, const float* gt, int32_t C, int32_t H, int32_t W, float* out2, double lambda_dssim,
                        void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream_) {
  if (!img || !gt || !out2 || C < 1 || H < 1 || W < 1) { set_error("%s: bad arguments", who); return TRASE_ERR_INVALID; }
  int nblocks = 0;
  if (!ws || ws_bytes < loss_ws_bytes(C, H, W, &nblocks)) { set_error("%s: workspace too small", who); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  float* dmaps = (float*)ws;
  float* partial = (float*)((char*)ws + align_up(sizeof(float) * 3 * (size_t)C * H * W));
  const dim3 grid((W + LT - 1) / LT, (H + LT - 1) / LT, C);
  {
    ProfScope ps("ssim_fwd", stream);
    hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(256), 0, stream, img, gt, H, W, make_window(), dmaps, partial);
  }
  TRASE_POST_LAUNCH("ssim_fwd", stream, 0);
  {
    ProfScope ps("loss_reduce", stream);
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, stream, partial, nblocks, 1.0f / ((float)C * H * W), out2,
                       (float)lambda_dssim, (float)(1.0 - lambda_dssim));
  }
  TRASE_POST_LAUNCH("loss_reduce", stream, 0);
  return TRASE_OK;
}

int trase_loss_l1_ssim_forward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, float* out2, void* ws,
                               size_t ws_bytes, int32_t device, trase_stream_t stream_) {
  return loss_forward("trase_loss_l1_ssim_forward", img, gt, C, H, W, out2, -1.0, ws, ws_bytes, device, stream_);
}

int trase_loss_photometric_forward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, double lambda_dssim, float* out3,
                                   void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream_) {
  if (!(lambda_dssim >= 0.0 && lambda_dssim <= 1.0)) { set_error("trase_loss_photometric_forward: lambda_dssim outside [0, 1]"); return TRASE_ERR_INVALID; }
  return loss_forward("trase_loss_photometric_forward", img, gt, C, H, W, out3, lambda_dssim, ws, ws_bytes, device, stream_);
}

static int loss_backward(const char* who, const float* img, const float* gt, int32_t C, int32_t H, int32_t W, const float* g2, int g_stride,
                         float s_l1, float s_ssim, const void* ws, size_t ws_bytes, float* dL_dimg, int32_t device,
                         trase_stream_t stream_) {
  if (!img || !gt || !g2 || !dL_dimg || C < 1 || H < 1 || W < 1) { set_error("%s: bad arguments", who); return TRASE_ERR_INVALID; }
  if (!ws || ws_bytes < loss_ws_bytes(C, H, W, nullptr)) { set_error("%s: workspace too small", who); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  const dim3 grid((W + LT - 1) / LT, (H + LT - 1) / LT, C);
  {
    ProfScope ps("ssim_bwd", stream);
    hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(256), 0, stream, img, gt, H, W, make_window(), (const float*)ws, g2,
                       1.0f / ((float)C * H * W), dL_dimg, g_stride, s_l1, s_ssim);
  }
  TRASE_POST_LAUNCH("ssim_bwd", stream, 0);
  return TRASE_OK;
}

int trase_loss_l1_ssim_backward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, const float* g2,
                                const void* ws, size_t ws_bytes, float* dL_dimg, int32_t device, trase_stream_t stream_) {
  return loss_backward("trase_loss_l1_ssim_backward", img, gt, C, H, W, g2, 1, 1.0f, 1.0f, ws, ws_bytes, dL_dimg, device, stream_);
}

int trase_loss_photometric_backward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, double lambda_dssim,
                                    const float* g, const void* ws, size_t ws_bytes, float* dL_dimg, int32_t device,
                                    trase_stream_t stream_) {
  return loss_backward("trase_loss_photometric_backward", img, gt, C, H, W, g, 0, (float)(1.0 - lambda_dssim), -(float)lambda_dssim, ws, ws_bytes, dL_dimg,
                       device, stream_);
}

}  // extern "C"
