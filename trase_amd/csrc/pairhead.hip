// pairhead.hip -- the FEATURE-state loss head (train.py:251-296) without any S x S matrix (SURVEY.md 8(f) rank 3:
// "contrastive S x S pair loss without materialising ~10 S x S fp32 temporaries").
// The reference builds, per iteration, for S ~ 5000 sampled pixels and N SAM masks at mask resolution H x W:
//   utils/feature_utils.py:17-26  get_sample_pixel_and_mask: sam_masks.sum(dim=0) == 0 (N x H x W pass, int64 result)
//   utils/feature_utils.py:28-38  get_pixel_weights: sam_masks * size (an N x H x W int64 temporary), S x S products,
//                                 max, clamp, min / max normalisation to [1, 10]
//   utils/feature_utils.py:40-49  get_pixel_mask_correspondence_matrix: C = (V^T V != 0), V = sampled masks x S
//   utils/feature_utils.py:51-57  get_features_correspondence_matrix: C_F = F F^T of the normalised sampled features
//   utils/loss_utils.py:275-406   the two pair losses on (C, C_F, weights);  train.py:295-296 the two mean similarities
// Every S x S entry is a function of per-pixel factors: C[i][j] = (bits_i & bits_j) != 0 with bits = membership in the
// sampled masks, C_F[i][j] = <f_i, f_j>, w[i][j] = g(a_i * a_j) with a = mean size of the masks covering the pixel (the
// matrix min / max needed by g are closed forms of min / max a).  So:
//   mask_stats      one pass over the masks: per-pixel cover count (the sampler's non_mask_region) + per-mask sizes
//   ph_gather       per sampled pixel: normalised feature row, membership bits, a
//   ph_consts       max a, min non-zero a  ->  ptp_max, w_max
//   ph_flags        all (i, j): column flags of the 'soft' / 'all' modes (atomicOr) + the four similarity sums
//   ph_sum          i < j: both masked sums (+ 'hard' selection sizes), per-workgroup partials, fixed-order reduction
//   ph_final        losses, candidate counts, similarities
//   ph_bwd          per sampled pixel t, all u: d f_t += coeff(t, u) f_u  (32 chunks of u, partials reduced in order)
//   ph_scatter      back through the normalisation, scattered into the zero-filled (32, H, W) gradient image
// plus featnorm_fwd / _bwd for the regulariser (1 - mean_p |F_p|)^2 (train.py:281-282).
// Built with -ffp-contract=off: the weight arithmetic follows torch's op order; dot products use explicit fmaf.
#include "common.h"

namespace trase {

constexpr int PH_F = 32;        // rendered feature channels (scene/gaussian_model.py:64 gaussian_features_dim)
constexpr int PH_ROWS = 64;     // rows per workgroup in the pair passes
constexpr int PH_CHUNKS = 32;   // u-chunks of the backward
constexpr int PH_MAXW = 8;      // 32-bit membership words: up to 256 sampled masks
constexpr int PH_MAXN = 8192;   // masks per view

struct PairWs {
  float* fn;        // [S][32] normalised sampled features
  float* rinv;      // [S] 1 / max(|f|, 1e-12)
  float* a;         // [S] mean size of the masks covering the pixel
  uint32_t* bits;   // [S][PH_MAXW]
  int* colP; int* colN;   // [S]
  float* consts;    // [8]: ptp_max, w_max, number of sampled masks
  double* partial;  // [nblk][8]
  float* dpart;     // [PH_CHUNKS][S][32]
  int* rank;        // [PH_MAXN] bit position of a sampled mask (-1: not sampled)
  int nblk;
};

static size_t pair_ws_carve(int S, PairWs* w, void* base) {
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes); return o; };
  const int bx = (S + 255) / 256, by = (S + PH_ROWS - 1) / PH_ROWS;
  const size_t o_fn = take(sizeof(float) * PH_F * (size_t)S), o_r = take(sizeof(float) * (size_t)S), o_a = take(sizeof(float) * (size_t)S),
               o_b = take(sizeof(uint32_t) * PH_MAXW * (size_t)S), o_cp = take(sizeof(int) * (size_t)S), o_cn = take(sizeof(int) * (size_t)S),
               o_c = take(sizeof(float) * 8), o_p = take(sizeof(double) * 8 * (size_t)bx * by),
               o_d = take(sizeof(float) * PH_F * (size_t)S * PH_CHUNKS), o_k = take(sizeof(int) * PH_MAXN);
  if (w && base) {
    char* b = (char*)base;
    w->fn = (float*)(b + o_fn); w->rinv = (float*)(b + o_r); w->a = (float*)(b + o_a); w->bits = (uint32_t*)(b + o_b);
    w->colP = (int*)(b + o_cp); w->colN = (int*)(b + o_cn); w->consts = (float*)(b + o_c); w->partial = (double*)(b + o_p);
    w->dpart = (float*)(b + o_d); w->rank = (int*)(b + o_k); w->nblk = bx * by;
  }
  return off;
}

// ---- mask statistics ---------------------------------------------------------------------------------------------------
// 4 pixels per thread (one 32-bit load of four bool bytes per mask); per-mask sizes: wave popcount -> LDS -> one global
// integer atomic per workgroup and mask (integers: order-independent)
__global__ __launch_bounds__(256) void mask_stats_kernel(const uint8_t* __restrict__ masks, int N, long long HW,
                                                         int32_t* __restrict__ cover, uint32_t* __restrict__ size) {
  extern __shared__ uint32_t lsize[];
  for (int n = threadIdx.x; n < N; n += 256) lsize[n] = 0u;
  __syncthreads();
  const long long p0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  const bool vec = (HW & 3) == 0 && p0 + 3 < HW;
  uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  for (int n0 = 0; n0 < N; n0 += 8) {
   // eight masks' words requested together: every resident wave walks the N masks in step, so with one load per trip the
   // kernel took N memory round trips (0.098 ms for 100 masks)
   uint32_t wv[8];
#pragma unroll
   for (int u = 0; u < 8; ++u) {
     uint32_t w = 0;
     if (n0 + u < N) {
       const uint8_t* row = masks + (size_t)(n0 + u) * HW;
       if (vec) w = *reinterpret_cast<const uint32_t*>(row + p0);
       else {
#pragma unroll
         for (int e = 0; e < 4; ++e) if (p0 + e < HW) w |= (row[p0 + e] ? 1u : 0u) << (8 * e);
       }
     }
     wv[u] = w;
   }
#pragma unroll
   for (int u = 0; u < 8; ++u) {
    const int n = n0 + u;
    if (n >= N) break;
    uint32_t w = wv[u];
    w = (w | (w >> 1) | (w >> 2) | (w >> 3) | (w >> 4) | (w >> 5) | (w >> 6) | (w >> 7)) & 0x01010101u;   // any non-zero byte -> 1
    c0 += w & 1u; c1 += (w >> 8) & 1u; c2 += (w >> 16) & 1u; c3 += w >> 24;
    // wave total of the four byte flags: four ballots + scalar popcounts instead of a six-step shuffle reduction
    const uint32_t cnt = (uint32_t)(__popcll(__ballot(w & 1u)) + __popcll(__ballot(w & 0x100u)) + __popcll(__ballot(w & 0x10000u)) +
                                    __popcll(__ballot(w & 0x1000000u)));
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&lsize[n], cnt);
   }
  }
  if (p0 < HW) cover[p0] = (int32_t)c0;
  if (p0 + 1 < HW) cover[p0 + 1] = (int32_t)c1;
  if (p0 + 2 < HW) cover[p0 + 2] = (int32_t)c2;
  if (p0 + 3 < HW) cover[p0 + 3] = (int32_t)c3;
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += 256) if (lsize[n]) atomicAdd(&size[n], lsize[n]);
}

// 16 pixels per thread (one 16-byte load per mask; HW a multiple of 16): round 5 -- with four bytes per lane a wave moved 256 bytes
// per load instruction and the kernel ran at 2.6 TB/s (0.080 ms for 100 masks at 1080p).  Cover counts accumulate as packed
// bytes (flushed every 255 masks); a mask's size is the wave total of the lanes' 0..16 set flags, gathered with five ballots.
__global__ __launch_bounds__(256) void mask_stats16_kernel(const uint8_t* __restrict__ masks, int N, long long HW,
                                                           int32_t* __restrict__ cover, uint32_t* __restrict__ size) {
  extern __shared__ uint32_t lsize[];
  for (int n = threadIdx.x; n < N; n += 256) lsize[n] = 0u;
  __syncthreads();
  const long long p0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 16;
  const bool in = p0 < HW;                                   // (HW % 16 == 0: a thread is inside with all sixteen pixels or not at all)
  uint32_t cnt32[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) cnt32[e] = 0u;
  uint32_t acc[4] = {0u, 0u, 0u, 0u};
  int since_flush = 0;
  auto flush = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int b = 0; b < 4; ++b) cnt32[4 * j + b] += (acc[j] >> (8 * b)) & 0xffu;
      acc[j] = 0u;
    }
    since_flush = 0;
  };
  for (int n0 = 0; n0 < N; n0 += 8) {
    uint4 wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      wv[u] = (in && n0 + u < N) ? *reinterpret_cast<const uint4*>(masks + (size_t)(n0 + u) * HW + p0) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = n0 + u;
      if (n >= N) break;
      uint32_t w[4] = {wv[u].x, wv[u].y, wv[u].z, wv[u].w};
      uint32_t c = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t v = w[j];
        v = ((((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v) & 0x80808080u) >> 7;       // any non-zero byte -> 1 (five operations)
        acc[j] += v;
        c += (uint32_t)__popc(v);
      }
      uint32_t tot = 0;
#pragma unroll
      for (int b = 0; b < 5; ++b) tot += (uint32_t)__popcll(__ballot((c >> b) & 1u)) << b;
      if ((threadIdx.x & 63) == 0 && tot) atomicAdd(&lsize[n], tot);
      if (++since_flush == 255) flush();
    }
  }
  flush();
  if (in) {
#pragma unroll
    for (int e = 0; e < 16; e += 4)
      *reinterpret_cast<int4*>(cover + p0 + e) = make_int4((int)cnt32[e], (int)cnt32[e + 1], (int)cnt32[e + 2], (int)cnt32[e + 3]);
  }
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += 256) if (lsize[n]) atomicAdd(&size[n], lsize[n]);
}

// ---- sampled pixels -> ascending index list, on the device ------------------------------------------------------------------
constexpr int CP_TILE = 4096;             // bytes of the mask per workgroup (16 per thread)
__global__ __launch_bounds__(256) void cp_count_kernel(const uint8_t* __restrict__ flags, long long HW, uint32_t* __restrict__ block_cnt) {
  const long long p0 = (long long)blockIdx.x * CP_TILE + (long long)threadIdx.x * 16;
  uint32_t c = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) c += (p0 + e < HW && flags[p0 + e]) ? 1u : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
  __shared__ uint32_t part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(256) void cp_scatter_kernel(const uint8_t* __restrict__ flags, long long HW, const uint32_t* __restrict__ block_cnt,
                                                         int nblocks, int32_t* __restrict__ pix, int cap, int32_t* __restrict__ count2) {
  __shared__ uint32_t sh[4], sh2[4], part[4];
  uint32_t before = 0, total = 0;                            // set bytes in the blocks in front of this one / in all blocks
  for (int b = threadIdx.x; b < nblocks; b += 256) {
    const uint32_t v = block_cnt[b];
    total += v;
    before += (b < (int)blockIdx.x) ? v : 0u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { before += (uint32_t)__shfl_xor((int)before, o); total += (uint32_t)__shfl_xor((int)total, o); }
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = before; sh2[threadIdx.x >> 6] = total; }
  __syncthreads();
  before = sh[0] + sh[1] + sh[2] + sh[3];
  total = sh2[0] + sh2[1] + sh2[2] + sh2[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) { count2[0] = (int32_t)min(total, (uint32_t)cap); count2[1] = (int32_t)total; }
  const long long p0 = (long long)blockIdx.x * CP_TILE + (long long)threadIdx.x * 16;
  uint32_t bits = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) bits |= (p0 + e < HW && flags[p0 + e]) ? (1u << e) : 0u;
  const uint32_t mine = (uint32_t)__popc(bits);
  uint32_t x = mine;                                         // inclusive scan over the workgroup
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, o); if ((int)(threadIdx.x & 63) >= o) x += y; }
  if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = x;
  __syncthreads();
  uint32_t at = before + x - mine;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) at += part[w];
#pragma unroll
  for (int e = 0; e < 16; ++e)
    if (bits & (1u << e)) { if (at < (uint32_t)cap) pix[at] = (int32_t)(p0 + e); ++at; }
}

// ---- per sampled pixel ---------------------------------------------------------------------------------------------------
// bit position of every sampled mask = its rank among the sampled ones (the row order of sam_masks[sampled_mask])
__global__ __launch_bounds__(256) void ph_rank_kernel(const uint8_t* __restrict__ sampled_mask, int N, int* __restrict__ rank,
                                                      float* __restrict__ consts) {
  __shared__ int carry;
  __shared__ int wsum[4];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < N; base += 256) {
    const int n = base + threadIdx.x;
    const int v = (n < N && sampled_mask[n]) ? 1 : 0;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if ((int)(threadIdx.x & 63) >= o) x += y; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
    __syncthreads();
    int add = carry;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) add += wsum[w];
    if (n < N) rank[n] = v ? add + x - 1 : -1;
    __syncthreads();
    if (threadIdx.x == 255) carry = add + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) consts[2] = (float)carry;
}

// eight lanes per sampled pixel: lane l walks the masks n = l, l + 8, ... (scattered one-byte reads), the partial
// membership words / size sums are combined with shuffles; four channels of the feature column per lane
__global__ __launch_bounds__(256) void ph_gather_kernel(const float* __restrict__ feats, long long HW,
                                                        const uint8_t* __restrict__ masks, int N, const int* __restrict__ rank,
                                                        const uint32_t* __restrict__ mask_size, const int32_t* __restrict__ pix,
                                                        int S_cap, const int* __restrict__ S_ptr, float* __restrict__ fn,
                                                        float* __restrict__ rinv, float* __restrict__ a, uint32_t* __restrict__ bits) {
  const int S = S_ptr ? min(*S_ptr, S_cap) : S_cap;            // device-side count (sync-free head) or the host's
  if (S <= 0) return;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int s = gid >> 3, l = gid & 7;
  const long long p = pix[min(s, S - 1)];
  float x[4];
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { x[e] = feats[(size_t)(4 * l + e) * HW + p]; ss = fmaf(x[e], x[e], ss); }
  uint32_t b[PH_MAXW];
#pragma unroll
  for (int k = 0; k < PH_MAXW; ++k) b[k] = 0u;
  unsigned long long tot = 0ull;
  uint32_t cnt = 0u;
  for (int n0 = l; n0 < N; n0 += 32) {
    uint8_t in4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) in4[e] = (n0 + 8 * e < N) ? masks[(size_t)(n0 + 8 * e) * HW + p] : (uint8_t)0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + 8 * e;
      if (n < N && in4[e]) {
        tot += mask_size[n]; ++cnt;
        const int k = rank[n];
        if (k >= 0) {
#pragma unroll
          for (int q = 0; q < PH_MAXW; ++q) if (q == (k >> 5)) b[q] |= 1u << (k & 31);
        }
      }
    }
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) {
    ss += __shfl_xor(ss, o);
    tot += __shfl_xor(tot, o);
    cnt += __shfl_xor((int)cnt, o);
#pragma unroll
    for (int q = 0; q < PH_MAXW; ++q) b[q] |= (uint32_t)__shfl_xor((int)b[q], o);
  }
  if (s >= S) return;
  const float r = 1.0f / fmaxf(sqrtf(ss), 1e-12f);                       // F.normalize(p=2, eps=1e-12)
  *reinterpret_cast<float4*>(fn + (size_t)s * PH_F + 4 * l) = make_float4(x[0] * r, x[1] * r, x[2] * r, x[3] * r);
  bits[(size_t)s * PH_MAXW + l] = b[l == 0 ? 0 : l == 1 ? 1 : l == 2 ? 2 : l == 3 ? 3 : l == 4 ? 4 : l == 5 ? 5 : l == 6 ? 6 : 7];
  if (l == 0) {
    rinv[s] = r;
    // per_pixel_mean_mask_size = sum(size of covering masks) / (cover count + 1e-9)   (utils/feature_utils.py:30-31)
    a[s] = (float)(long long)tot / ((float)cnt + 1e-9f);
  }
}

// ptp_max = (max a)^2; w_max = max(1, ptp_max / (min non-zero a)^2); the matrix minimum of the clamped ratio is exactly 1
// (the entry of the two largest a, ptp_max / ptp_max, or any zero product replaced by 1e10)
__global__ __launch_bounds__(256) void ph_consts_kernel(const float* __restrict__ a, int S_cap, const int* __restrict__ S_ptr,
                                                        float* __restrict__ consts) {
  const int S = S_ptr ? min(*S_ptr, S_cap) : S_cap;            // device-side count (sync-free head) or the host's
  __shared__ float smax[256], smin[256];
  float mx = 0.f, mn = INFINITY;
  for (int s = threadIdx.x; s < S; s += 256) { const float v = a[s]; mx = fmaxf(mx, v); if (v > 0.f) mn = fminf(mn, v); }
  smax[threadIdx.x] = mx; smin[threadIdx.x] = mn;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + o]); smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + o]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float P = smax[0] * smax[0];
    float wmax = 1.0f;
    if (smin[0] < INFINITY) wmax = fmaxf(P / (smin[0] * smin[0]), 1.0f);
    consts[0] = P; consts[1] = wmax;
  }
}

__device__ __forceinline__ float pair_weight(float ai, float aj, float P, float wmax, int use_w) {
  if (!use_w) return 1.0f;
  float pm = ai * aj;
  if (pm == 0.0f) pm = 1e10f;
  const float w = fmaxf(P / pm, 1.0f);
  return (w - 1.0f) / (wmax - 1.0f) * 9.0f + 1.0f;        // (w - w.min()) / (w.max() - w.min()) * 9. + 1.
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(4))) const f32x2 cfloat2;      // constant address space: wave-uniform rows come through
typedef __attribute__((address_space(4))) const uint32_t cuint;     // the scalar cache (s_load) and feed the VALU as SGPR pairs

__device__ __forceinline__ bool share_mask(const uint32_t* bj, const uint32_t* bi_) {
  cuint* bi = (cuint*)bi_;
  uint32_t x = 0u;
#pragma unroll
  for (int q = 0; q < PH_MAXW; ++q) x |= bj[q] & bi[q];
  return x != 0u;
}

// <f_i, f_j> with channel pairs on v_pk_fma_f32 (two partial sums, added at the end)
__device__ __forceinline__ float dot32(const f32x2* fj, const float* fi_) {
  cfloat2* fi = (cfloat2*)fi_;
  f32x2 d = {0.f, 0.f};
#pragma unroll
  for (int c = 0; c < PH_F / 2; ++c) d = __builtin_elementwise_fma(fi[c], fj[c], d);
  return d.x + d.y;
}

// kind: bit 1-2 = mode (0 soft, 1 all, 2 hard)
__global__ __launch_bounds__(256) void ph_flags_kernel(const float* __restrict__ fn, const uint32_t* __restrict__ bits, int S_cap,
                                                       const int* __restrict__ S_ptr, float pth, float nth, int mode,
                                                       int* __restrict__ colP, int* __restrict__ colN, double* __restrict__ partial) {
  __shared__ double red[4][4];
  const int S = S_ptr ? min(*S_ptr, S_cap) : S_cap;            // device-side count (sync-free head) or the host's
  if ((int)(blockIdx.y * PH_ROWS) >= S || (int)(blockIdx.x * 256) >= S) {     // nothing of this tile exists (grids are sized for S_cap)
    if (threadIdx.x < 4) partial[8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + 4 + threadIdx.x] = 0.0;
    return;
  }
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * PH_ROWS, i1 = min(i0 + PH_ROWS, S);
  f32x2 fj[PH_F / 2]; uint32_t bj[PH_MAXW];
  const int jj = min(j, S - 1);
#pragma unroll
  for (int c = 0; c < PH_F / 2; ++c) fj[c] = *reinterpret_cast<const f32x2*>(fn + (size_t)jj * PH_F + 2 * c);
#pragma unroll
  for (int q = 0; q < PH_MAXW; ++q) bj[q] = bits[(size_t)jj * PH_MAXW + q];
  bool anyP = false, anyN = false;
  float sp = 0.f, sn = 0.f, np_ = 0.f, nn_ = 0.f;           // similarity sums over this thread's <= 64 rows
  for (int i = i0; i < i1; ++i) {
    const float f = dot32(fj, fn + (size_t)i * PH_F);
    const bool c = share_mask(bj, bits + (size_t)i * PH_MAXW);
    if (c) { sp += f; np_ += 1.f; anyP |= (mode == 1) || f < pth; }
    else   { sn += f; nn_ += 1.f; anyN |= (mode == 1) || f > nth; }
  }
  if (j >= S) { sp = sn = np_ = nn_ = 0.f; anyP = anyN = false; }
  if (mode != 2) { if (anyP) atomicOr(&colP[j], 1); if (anyN) atomicOr(&colN[j], 1); }
  double v[4] = {(double)sp, (double)np_, (double)sn, (double)nn_};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    partial[8 * b + 4 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
  }
}

__global__ __launch_bounds__(256) void ph_sum_kernel(const float* __restrict__ fn, const uint32_t* __restrict__ bits,
                                                     const float* __restrict__ a, const float* __restrict__ consts, int S_cap,
                                                     const int* __restrict__ S_ptr, float pth, float nth, int mode, int use_w,
                                                     const int* __restrict__ colP, const int* __restrict__ colN,
                                                     double* __restrict__ partial) {
  __shared__ double red[4][4];
  const int S = S_ptr ? min(*S_ptr, S_cap) : S_cap;            // device-side count (sync-free head) or the host's
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * PH_ROWS, i1 = min(i0 + PH_ROWS, S);
  if (i0 >= min(S, (int)(blockIdx.x * 256 + 256))) {        // the whole tile is on or below the diagonal: nothing to add
    if (threadIdx.x < 4) partial[8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + threadIdx.x] = 0.0;
    return;
  }
  const int jj = min(j, S - 1);
  f32x2 fj[PH_F / 2]; uint32_t bj[PH_MAXW];
#pragma unroll
  for (int c = 0; c < PH_F / 2; ++c) fj[c] = *reinterpret_cast<const f32x2*>(fn + (size_t)jj * PH_F + 2 * c);
#pragma unroll
  for (int q = 0; q < PH_MAXW; ++q) bj[q] = bits[(size_t)jj * PH_MAXW + q];
  const float aj = a[jj], P = consts[0], wmax = consts[1];
  const bool hard = mode == 2;
  const bool cp = hard || colP[jj] != 0, cn = hard || colN[jj] != 0;
  float accP = 0.f, cntP = 0.f, accN = 0.f, cntN = 0.f;
  const int ie = (j < S) ? min(i1, j) : i0;                 // strictly upper triangle
  for (int i = i0; i < ie; ++i) {
    const float f = dot32(fj, fn + (size_t)i * PH_F);
    const bool c = share_mask(bj, bits + (size_t)i * PH_MAXW);
    const float w = pair_weight(a[i], aj, P, wmax, use_w);
    if (c) { if (cp && (!hard || f < pth)) { accP -= w * f; cntP += 1.f; } }
    else if (cn && (!hard || f > nth)) { accN += w * fmaxf(f, 0.f); cntN += 1.f; }
  }
  double v[4] = {(double)accP, (double)cntP, (double)accN, (double)cntN};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    partial[8 * b + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
  }
}

// out8 = {loss_pos, N_pos, loss_neg, N_neg, pos_similarity, neg_similarity, S, sampled masks}
__global__ __launch_bounds__(256) void ph_final_kernel(const double* __restrict__ partial, int nblk, const int* __restrict__ colP,
                                                       const int* __restrict__ colN, int S_cap, const int* __restrict__ S_ptr,
                                                       int mode, const float* __restrict__ consts, float* __restrict__ out8) {
  __shared__ double sh[10][256];
  const int S = S_ptr ? min(*S_ptr, S_cap) : S_cap;            // device-side count (sync-free head) or the host's

  double v[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) v[k] = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += partial[8 * (size_t)b + k];
  }
  if (mode != 2) for (int j = threadIdx.x; j < S; j += 256) { v[8] += colP[j] ? (double)j : 0.0; v[9] += colN[j] ? (double)j : 0.0; }
#pragma unroll
  for (int k = 0; k < 10; ++k) sh[k][threadIdx.x] = v[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
#pragma unroll
      for (int k = 0; k < 10; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double Np = (mode == 2) ? sh[1][0] : sh[8][0], Nn = (mode == 2) ? sh[3][0] : sh[9][0];
    out8[0] = (float)(Np > 0.0 ? sh[0][0] / Np : 0.0); out8[1] = (float)Np;
    out8[2] = (float)(Nn > 0.0 ? sh[2][0] / Nn : 0.0); out8[3] = (float)Nn;
    out8[4] = (float)(sh[4][0] / sh[5][0]);                  // C_F[C == 1].mean(): nan when empty, like torch
    out8[5] = (float)(sh[6][0] / sh[7][0]);
    out8[6] = (float)S; out8[7] = consts[2];
  }
}

// d f_t += coeff(min(t,u), max(t,u)) * f_u over one chunk of u; coeff = dL/dC_F of the pair
__global__ __launch_bounds__(256) void ph_bwd_kernel(const float* __restrict__ fn, const uint32_t* __restrict__ bits,
                                                     const float* __restrict__ a, const float* __restrict__ consts, int S_cap,
                                                     const int* __restrict__ S_ptr, float pth, float nth, int mode, int use_w,
                                                     const int* __restrict__ colP, const int* __restrict__ colN,
                                                     const float* __restrict__ out8, const float* __restrict__ g2,
                                                     float* __restrict__ dpart) {
  const int S = S_ptr ? min(*S_ptr, S_cap) : S_cap;            // device-side count (sync-free head) or the host's
  if ((int)(blockIdx.x * 256) >= S) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int per = (S + PH_CHUNKS - 1) / PH_CHUNKS;
  const int u0 = blockIdx.y * per, u1 = min(u0 + per, S);
  const int tt = min(t, S - 1);
  f32x2 ft[PH_F / 2], acc[PH_F / 2]; uint32_t bt[PH_MAXW];
#pragma unroll
  for (int c = 0; c < PH_F / 2; ++c) { ft[c] = *reinterpret_cast<const f32x2*>(fn + (size_t)tt * PH_F + 2 * c); acc[c] = f32x2{0.f, 0.f}; }
#pragma unroll
  for (int q = 0; q < PH_MAXW; ++q) bt[q] = bits[(size_t)tt * PH_MAXW + q];
  const float at = a[tt], P = consts[0], wmax = consts[1];
  const bool hard = mode == 2;
  const float Np = out8[1], Nn = out8[3];
  const float kp = (Np > 0.f) ? -g2[0] / Np : 0.f, kn = (Nn > 0.f) ? g2[1] / Nn : 0.f;
  const bool cpt = hard || colP[tt] != 0, cnt_ = hard || colN[tt] != 0;
  for (int u = u0; u < u1; ++u) {
    if (u == tt) continue;
    const float* fu = fn + (size_t)u * PH_F;
    const float f = dot32(ft, fu);
    const bool c = share_mask(bt, bits + (size_t)u * PH_MAXW);
    // the column flag belongs to the larger index of the pair
    const bool cp = (u > tt) ? (hard || colP[u] != 0) : cpt, cn = (u > tt) ? (hard || colN[u] != 0) : cnt_;
    float d = 0.f;
    if (c) { if (cp && (!hard || f < pth)) d = kp; }
    else if (cn && f > 0.f && (!hard || f > nth)) d = kn;
    if (d != 0.f) {
      d *= pair_weight(a[u], at, P, wmax, use_w);
      const f32x2 d2 = {d, d};
      cfloat2* fu2 = (cfloat2*)fu;
#pragma unroll
      for (int c2 = 0; c2 < PH_F / 2; ++c2) acc[c2] = __builtin_elementwise_fma(d2, fu2[c2], acc[c2]);
    }
  }
  if (t < S) {
    float* o = dpart + ((size_t)blockIdx.y * S + t) * PH_F;
#pragma unroll
    for (int c = 0; c < PH_F / 2; c += 2) *reinterpret_cast<float4*>(o + 2 * c) = make_float4(acc[c].x, acc[c].y, acc[c + 1].x, acc[c + 1].y);
  }
}

// reduce the chunks in order, back through x -> x / max(|x|, eps), scatter into the gradient image.
// One thread per (sampled pixel, channel): a 32-lane half wave owns a pixel and reduces <fn, d> with shuffles.
__global__ __launch_bounds__(256) void ph_scatter_kernel(const float* __restrict__ dpart, const float* __restrict__ fn,
                                                         const float* __restrict__ rinv, const int32_t* __restrict__ pix, int S_cap,
                                                         const int* __restrict__ S_ptr, long long HW, int accumulate,
                                                         float* __restrict__ dfeats) {
  const int S = S_ptr ? min(*S_ptr, S_cap) : S_cap;            // device-side count (sync-free head) or the host's
  if ((int)((blockIdx.x * 256) >> 5) >= S) return;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int s = gid >> 5, c = gid & 31;
  const int ss = min(s, S - 1);
  float d = 0.f;
  for (int k = 0; k < PH_CHUNKS; ++k) d += dpart[((size_t)k * S + ss) * PH_F + c];
  const float r = rinv[ss], f = fn[(size_t)ss * PH_F + c];
  float proj = f * d;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) proj += __shfl_xor(proj, o);
  if (!(r < 1e12f)) proj = 0.f;                             // |x| <= eps: the clamp is active, d x = r d
  if (s >= S) return;
  const float v = r * (d - f * proj);
  float* o = dfeats + (size_t)c * HW + pix[ss];            // sampled pixels are distinct: no atomics
  *o = accumulate ? *o + v : v;
}

// ---- feature-norm regulariser ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void featnorm_fwd_kernel(const float* __restrict__ feats, long long HW, int F,
                                                           double* __restrict__ partial) {
  __shared__ double red[4];
  double acc = 0.0;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long long)gridDim.x * 256) {
    float ss = 0.f;
    int c = 0;
    for (; c + 8 <= F; c += 8) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = feats[(size_t)(c + e) * HW + p];
#pragma unroll
      for (int e = 0; e < 8; ++e) ss = fmaf(x[e], x[e], ss);
    }
    for (; c < F; ++c) { const float x = feats[(size_t)c * HW + p]; ss = fmaf(x, x, ss); }
    acc += (double)sqrtf(ss);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void featnorm_final_kernel(const double* __restrict__ partial, int nblk, long long HW,
                                                             float* __restrict__ out2) {
  __shared__ double sh[256];
  double v = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) v += partial[b];
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const float mean = (float)(sh[0] / (double)HW);
    out2[0] = (1.0f - mean) * (1.0f - mean);                 // (1 - rendered_feature_norm) ** 2
    out2[1] = mean;
  }
}

// d/dF of (1 - mean)^2 = -2 (1 - mean) / HW * F_p / |F_p|   (zero where |F_p| == 0, torch's norm subgradient)
__global__ __launch_bounds__(256) void featnorm_bwd_kernel(const float* __restrict__ feats, long long HW, int F,
                                                           const float* __restrict__ out2, const float* __restrict__ g,
                                                           float* __restrict__ dfeats) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const float coef = g[0] * (-2.0f * (1.0f - out2[1])) / (float)HW;
  if (F == PH_F) {                                          // the rendered 32-d features: one read, held in registers
    float x[PH_F];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < PH_F; ++c) x[c] = feats[(size_t)c * HW + p];
#pragma unroll
    for (int c = 0; c < PH_F; ++c) ss = fmaf(x[c], x[c], ss);
    const float n = sqrtf(ss);
    const float k = (n > 0.f) ? coef / n : 0.f;
#pragma unroll
    for (int c = 0; c < PH_F; ++c) dfeats[(size_t)c * HW + p] = k * x[c];
    return;
  }
  float ss = 0.f;
  for (int c = 0; c < F; ++c) { const float x = feats[(size_t)c * HW + p]; ss = fmaf(x, x, ss); }
  const float n = sqrtf(ss);
  const float k = (n > 0.f) ? coef / n : 0.f;
  for (int c = 0; c < F; ++c) dfeats[(size_t)c * HW + p] = k * feats[(size_t)c * HW + p];
}

}  // namespace trase

using namespace trase;

extern "C" {

int trase_mask_stats(const uint8_t* sam_masks, int32_t N, int64_t HW, int32_t* cover_count, uint32_t* mask_size, int32_t device,
                     trase_stream_t stream_) {
  if (!sam_masks || !cover_count || !mask_size || N < 1 || N > 8192 || HW < 1) { set_error("trase_mask_stats: bad arguments"); return TRASE_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  launch_zero_bytes(mask_size, sizeof(uint32_t) * (size_t)N, stream);
  {
    ProfScope ps("mask_stats", stream);
    if ((HW & 15) == 0 && (((size_t)sam_masks | (size_t)cover_count) & 15) == 0)
      hipLaunchKernelGGL(mask_stats16_kernel, dim3((unsigned)((HW + 4095) / 4096)), dim3(256), sizeof(uint32_t) * (size_t)N, stream,
                         sam_masks, N, (long long)HW, cover_count, mask_size);
    else
      hipLaunchKernelGGL(mask_stats_kernel, dim3((unsigned)((HW + 1023) / 1024)), dim3(256), sizeof(uint32_t) * (size_t)N, stream,
                         sam_masks, N, (long long)HW, cover_count, mask_size);
  }
  TRASE_POST_LAUNCH("mask_stats", stream, 0);
  return TRASE_OK;
}

int trase_compact_pixels_sizes(int64_t HW, size_t* ws_bytes) {
  if (!ws_bytes || HW < 1) { set_error("trase_compact_pixels_sizes: bad arguments"); return TRASE_ERR_INVALID; }
  *ws_bytes = align_up(sizeof(uint32_t) * (size_t)((HW + CP_TILE - 1) / CP_TILE));
  return TRASE_OK;
}

int trase_compact_pixels(const uint8_t* flags, int64_t HW, int32_t* pix, int32_t cap, int32_t* count2, void* ws, size_t ws_bytes,
                         int32_t device, trase_stream_t stream_) {
  if (!flags || !pix || !count2 || HW < 1 || cap < 1) { set_error("trase_compact_pixels: bad arguments"); return TRASE_ERR_INVALID; }
  const int nblocks = (int)((HW + CP_TILE - 1) / CP_TILE);
  if (!ws || ws_bytes < align_up(sizeof(uint32_t) * (size_t)nblocks)) { set_error("trase_compact_pixels: workspace too small"); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  {
    ProfScope ps("compact_pixels", stream);
    hipLaunchKernelGGL(cp_count_kernel, dim3(nblocks), dim3(256), 0, stream, flags, (long long)HW, (uint32_t*)ws);
    hipLaunchKernelGGL(cp_scatter_kernel, dim3(nblocks), dim3(256), 0, stream, flags, (long long)HW, (const uint32_t*)ws, nblocks, pix, cap,
                       count2);
  }
  TRASE_POST_LAUNCH("compact_pixels", stream, 0);
  return TRASE_OK;
}

int trase_pairhead_sizes(int32_t S, size_t* ws_bytes) {
  if (!ws_bytes || S < 1) { set_error("trase_pairhead_sizes: bad arguments"); return TRASE_ERR_INVALID; }
  *ws_bytes = pair_ws_carve(S, nullptr, nullptr);
  return TRASE_OK;
}

int trase_pairhead_forward(const float* feats, int32_t F, int64_t HW, const uint8_t* sam_masks, int32_t N,
                           const uint8_t* sampled_mask, int32_t n_sampled_masks, const uint32_t* mask_size, const int32_t* pix,
                           int32_t S, int32_t mode, float positive_th, float negative_th, int32_t use_weights, float* out8,
                           void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream_) {
  return trase_pairhead_forward_n(feats, F, HW, sam_masks, N, sampled_mask, n_sampled_masks, mask_size, pix, S, nullptr, mode, positive_th,
                                  negative_th, use_weights, out8, ws, ws_bytes, device, stream_);
}

int trase_pairhead_forward_n(const float* feats, int32_t F, int64_t HW, const uint8_t* sam_masks, int32_t N,
                             const uint8_t* sampled_mask, int32_t n_sampled_masks, const uint32_t* mask_size, const int32_t* pix,
                             int32_t S, const int32_t* S_dev, int32_t mode, float positive_th, float negative_th, int32_t use_weights,
                             float* out8, void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream_) {
  if (!feats || !sam_masks || !sampled_mask || !mask_size || !pix || !out8 || S < 1 || N < 1 || N > PH_MAXN || HW < 1 || mode < 0 || mode > 2) {
    set_error("trase_pairhead_forward: bad arguments"); return TRASE_ERR_INVALID;
  }
  if (F != PH_F) { set_error("trase_pairhead_forward: %d feature channels (compiled for %d)", F, PH_F); return TRASE_ERR_INVALID; }
  if (n_sampled_masks < 0 || n_sampled_masks > 32 * PH_MAXW) {
    set_error("trase_pairhead_forward: %d sampled masks (at most %d)", n_sampled_masks, 32 * PH_MAXW); return TRASE_ERR_INVALID;
  }
  PairWs w;
  if (!ws || ws_bytes < pair_ws_carve(S, &w, ws)) { set_error("trase_pairhead_forward: workspace too small"); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  // colP and colN are neighbours in the workspace (pair_ws_carve): one fill for both
  launch_zero_bytes(w.colP, (size_t)((char*)w.colN - (char*)w.colP) + sizeof(int) * (size_t)S, stream);
  const dim3 grid((S + 255) / 256, (S + PH_ROWS - 1) / PH_ROWS);
  {
    ProfScope ps("pairhead_fwd", stream);
    hipLaunchKernelGGL(ph_rank_kernel, dim3(1), dim3(256), 0, stream, sampled_mask, N, w.rank, w.consts);
    hipLaunchKernelGGL(ph_gather_kernel, dim3((S * 8 + 255) / 256), dim3(256), 0, stream, feats, (long long)HW, sam_masks, N, w.rank,
                       mask_size, pix, S, (const int*)S_dev, w.fn, w.rinv, w.a, w.bits);
    hipLaunchKernelGGL(ph_consts_kernel, dim3(1), dim3(256), 0, stream, w.a, S, (const int*)S_dev, w.consts);
    hipLaunchKernelGGL(ph_flags_kernel, grid, dim3(256), 0, stream, w.fn, w.bits, S, (const int*)S_dev, positive_th, negative_th, mode, w.colP,
                       w.colN, w.partial);
    hipLaunchKernelGGL(ph_sum_kernel, grid, dim3(256), 0, stream, w.fn, w.bits, w.a, w.consts, S, (const int*)S_dev, positive_th, negative_th,
                       mode, use_weights, w.colP, w.colN, w.partial);
    hipLaunchKernelGGL(ph_final_kernel, dim3(1), dim3(256), 0, stream, w.partial, w.nblk, w.colP, w.colN, S, (const int*)S_dev, mode, w.consts,
                       out8);
  }
  TRASE_POST_LAUNCH("pairhead_fwd", stream, 0);
  return TRASE_OK;
}

int trase_pairhead_backward(int32_t F, int64_t HW, const int32_t* pix, int32_t S, int32_t mode, float positive_th,
                            float negative_th, int32_t use_weights, const float* out8, const float* g2, const void* ws,
                            size_t ws_bytes, int32_t accumulate, float* dL_dfeats, int32_t device, trase_stream_t stream_) {
  return trase_pairhead_backward_n(F, HW, pix, S, nullptr, mode, positive_th, negative_th, use_weights, out8, g2, ws, ws_bytes, accumulate,
                                   dL_dfeats, device, stream_);
}

int trase_pairhead_backward_n(int32_t F, int64_t HW, const int32_t* pix, int32_t S, const int32_t* S_dev, int32_t mode, float positive_th,
                              float negative_th, int32_t use_weights, const float* out8, const float* g2, const void* ws,
                              size_t ws_bytes, int32_t accumulate, float* dL_dfeats, int32_t device, trase_stream_t stream_) {
  if (!pix || !out8 || !g2 || !dL_dfeats || S < 1 || HW < 1 || mode < 0 || mode > 2 || F != PH_F) {
    set_error("trase_pairhead_backward: bad arguments"); return TRASE_ERR_INVALID;
  }
  PairWs w;
  if (!ws || ws_bytes < pair_ws_carve(S, &w, const_cast<void*>(ws))) { set_error("trase_pairhead_backward: workspace too small"); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  if (!accumulate) launch_zero_bytes(dL_dfeats, sizeof(float) * (size_t)F * (size_t)HW, stream);
  {
    ProfScope ps("pairhead_bwd", stream);
    hipLaunchKernelGGL(ph_bwd_kernel, dim3((S + 255) / 256, PH_CHUNKS), dim3(256), 0, stream, w.fn, w.bits, w.a, w.consts, S, (const int*)S_dev,
                       positive_th, negative_th, mode, use_weights, w.colP, w.colN, out8, g2, w.dpart);
    hipLaunchKernelGGL(ph_scatter_kernel, dim3((S * PH_F + 255) / 256), dim3(256), 0, stream, w.dpart, w.fn, w.rinv, pix, S, (const int*)S_dev,
                       (long long)HW, accumulate, dL_dfeats);
  }
  TRASE_POST_LAUNCH("pairhead_bwd", stream, 0);
  return TRASE_OK;
}

int trase_featnorm_sizes(int64_t HW, size_t* ws_bytes) {
  if (!ws_bytes || HW < 1) { set_error("trase_featnorm_sizes: bad arguments"); return TRASE_ERR_INVALID; }
  *ws_bytes = align_up(sizeof(double) * 1024);
  return TRASE_OK;
}

int trase_featnorm_forward(const float* feats, int32_t F, int64_t HW, float* out2, void* ws, size_t ws_bytes, int32_t device,
                           trase_stream_t stream_) {
  if (!feats || !out2 || F < 1 || HW < 1) { set_error("trase_featnorm_forward: bad arguments"); return TRASE_ERR_INVALID; }
  if (!ws || ws_bytes < align_up(sizeof(double) * 1024)) { set_error("trase_featnorm_forward: workspace too small"); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  const int nblk = (int)std::min<long long>(1024, (HW + 255) / 256);
  {
    ProfScope ps("featnorm_fwd", stream);
    hipLaunchKernelGGL(featnorm_fwd_kernel, dim3(nblk), dim3(256), 0, stream, feats, (long long)HW, F, (double*)ws);
    hipLaunchKernelGGL(featnorm_final_kernel, dim3(1), dim3(256), 0, stream, (const double*)ws, nblk, (long long)HW, out2);
  }
  TRASE_POST_LAUNCH("featnorm_fwd", stream, 0);
  return TRASE_OK;
}

int trase_featnorm_backward(const float* feats, int32_t F, int64_t HW, const float* out2, const float* g, float* dL_dfeats,
                            int32_t device, trase_stream_t stream_) {
  if (!feats || !out2 || !g || !dL_dfeats || F < 1 || HW < 1) { set_error("trase_featnorm_backward: bad arguments"); return TRASE_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  {
    ProfScope ps("featnorm_bwd", stream);
    hipLaunchKernelGGL(featnorm_bwd_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, stream, feats, (long long)HW, F, out2, g,
                       dL_dfeats);
  }
  TRASE_POST_LAUNCH("featnorm_bwd", stream, 0);
  return TRASE_OK;
}

}  // extern "C"
