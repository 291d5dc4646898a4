// nnfm.hip -- nearest-neighbour feature matching style loss (third loss head of the harness):
//   loss_nnfm_style(feat1 (C,N1), feats2 (C,N2)) = mean_i min_j (1 - <f1_i, f2_j> / (|f1_i| |f2_j|))
// utils/loss_utils.py:223-228, called at train_style_transfer_nnfm.py:201-203 on VGG conv features of the rendered frame
// (N1 = H/8 * W/8 positions, C = 512) and of the style image.  The reference materialises the N1 x N2 cosine matrix
// (32 400^2 floats = 4.2 GB for a 1080p frame); here it never exists:
//   1. nnfm_prep: columns L2-normalised, converted to bf16 and transposed to [position][channel] (K contiguous);
//   2. nnfm_match: bf16 MFMA GEMM of the two normalised matrices with a running row-maximum: a wave keeps the A panel
//      of its 32 rows in registers (C <= 512), the workgroup streams 32-column B tiles through LDS, every lane tracks the
//      best TWO (value, column) of its 16 accumulator rows, lanes are merged once at the end -> two candidate columns per row;
//   3. nnfm_finish: both candidates' cosines are re-evaluated in fp32 from the original data and the larger one is the match
//      (the bf16 product only SHORT-LISTS: two neighbours closer than the bf16 rounding of a C-term product -- ~1 % of the rows
//      of a 1000 x 777 problem -- would otherwise be decided by that rounding, and the row's gradient would point at the other
//      neighbour), 1 - cos summed in a fixed order -> deterministic scalar;
//   4. nnfm_bwd: d loss / d feat1 through the arg-min (feats2, the style reference, takes no gradient:
//      train_style_transfer_nnfm.py:199 evaluates it without a graph to the Gaussians).
#include "common.h"

namespace trase {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NN_MAXC = 512;

// ---- 1. normalise + transpose: in (C, N) fp32 -> out [N][C] bf16, inv_norm[N] ---------------------------------------
// one workgroup of 256 threads per 64 positions: coalesced reads along N, LDS transpose, 16-byte writes along C
__global__ __launch_bounds__(256) void nnfm_prep_kernel(const float* __restrict__ in, int C, int N, __bf16* __restrict__ out,
                                                        float* __restrict__ inv_norm) {
  __shared__ float s_t[64][NN_MAXC / 8 + 1];          // one 64 x 64 channel slab at a time
  __shared__ float s_n2[4][64];
  __shared__ float s_inv[64];
  const int n0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // tx: position, ty: 0..3
  const int n = n0 + tx;
  // pass 1: squared norms
  float acc = 0.f;
  if (n < N)
    for (int c = ty; c < C; c += 4) { const float v = in[(size_t)c * N + n]; acc = fmaf(v, v, acc); }
  s_n2[ty][tx] = acc;
  __syncthreads();
  if (ty == 0) {
    const float n2 = (s_n2[0][tx] + s_n2[1][tx]) + (s_n2[2][tx] + s_n2[3][tx]);
    const float inv = 1.0f / sqrtf(n2);               // the reference divides by the norm without an epsilon
    s_inv[tx] = inv;
    if (n < N) inv_norm[n] = inv;
  }
  __syncthreads();
  // pass 2: 64-channel slabs through LDS
  for (int c0 = 0; c0 < C; c0 += 64) {
    for (int cc = ty; cc < 64; cc += 4) {
      const int c = c0 + cc;
      s_t[tx][cc] = (n < N && c < C) ? in[(size_t)c * N + n] * s_inv[tx] : 0.f;
    }
    __syncthreads();
    // 64 positions x 64 channels -> each thread writes 16 channels of one position (two 16-byte stores)
    const int p = threadIdx.x >> 2, seg = threadIdx.x & 3;
    if (n0 + p < N) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (__bf16)s_t[p][16 * seg + 8 * half + e];
        const int c = c0 + 16 * seg + 8 * half;
        if (c < C) *reinterpret_cast<bf16x8*>(out + (size_t)(n0 + p) * C + c) = v;
      }
    }
    __syncthreads();
  }
}

// ---- 2. GEMM with running row maximum -------------------------------------------------------------------------------
// a > b ? if_true : if_false as one v_cmp + one v_cndmask
__device__ __forceinline__ uint32_t sel_gt(float a, float b, uint32_t if_false, uint32_t if_true) {
  uint32_t r;
  asm("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %4, vcc" : "=v"(r) : "v"(a), "v"(b), "v"(if_false), "v"(if_true) : "vcc");
  return r;
}

constexpr int NN_WPB = 4;                               // waves per workgroup, 32 rows each
constexpr int NN_BT = 32;                               // columns (style positions) per B tile

template <int KS>                                       // KS = C / 16 K-steps (<= 32)
__global__ __launch_bounds__(NN_WPB* WAVE) __attribute__((amdgpu_waves_per_eu(2))) void nnfm_match_kernel(const __bf16* __restrict__ A, int N1, const __bf16* __restrict__ B,
                                                                  int N2, int32_t* __restrict__ best_j, int32_t* __restrict__ second_j) {
  constexpr int C = KS * 16;
  constexpr int LDB = C + 8;                            // padded row pitch of the B tile (bf16): conflict-free 16-byte reads
  __shared__ __attribute__((aligned(16))) __bf16 s_b[2][NN_BT * LDB];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  const int row0 = (blockIdx.x * NN_WPB + wave) * 32;
  // A panel: row row0 + m, all K-steps, this lane's half of each (8 channels)
  bf16x8 ap[KS];
  {
    const int r = min(row0 + m, N1 - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ap[ks] = *reinterpret_cast<const bf16x8*>(A + (size_t)r * C + ks * 16 + 8 * h);
  }
  float bestv[16], secv[16];                            // the two largest products of each accumulator row ...
  uint32_t cols[16];                                    // ... and the TILES they came from: best | second << 16 (the column inside
#pragma unroll                                          // a tile is the lane's own m; one register instead of two keeps C = 512
  for (int r = 0; r < 16; ++r) { bestv[r] = -INFINITY; secv[r] = -INFINITY; cols[r] = 0u; }   // at two waves per SIMD)
  const int ntile = (N2 + NN_BT - 1) / NN_BT;
  // cooperative staging: the tile is NN_BT rows of C bf16 = C / 8 16-byte pieces per row
  constexpr int PIECES = NN_BT * (C / 8);
  auto stage = [&](int t, int buf) {
    for (int p = threadIdx.x; p < PIECES; p += NN_WPB * WAVE) {
      const int jr = p / (C / 8), pc = p % (C / 8);
      const int j = min(t * NN_BT + jr, N2 - 1);
      *reinterpret_cast<bf16x8*>(&s_b[buf][jr * LDB + 8 * pc]) = *reinterpret_cast<const bf16x8*>(B + (size_t)j * C + 8 * pc);
    }
  };
  stage(0, 0);
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntile) stage(t + 1, buf ^ 1);
    f32x16 D;
#pragma unroll
    for (int r = 0; r < 16; ++r) D[r] = 0.f;
    const __bf16* bt = &s_b[buf][m * LDB + 8 * h];       // B fragment: column (style position) m of the tile
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 bf = *reinterpret_cast<const bf16x8*>(bt + ks * 16);
      D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[ks], bf, D, 0, 0, 0);
    }
    // D[i][j]: lane (j = m, h) holds rows i = 8q + 4h + r; column index of this lane in the whole matrix:
    const int j = t * NN_BT + m;
    if (j < N2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {                                 // strict: the first (lowest) column wins ties inside a lane
        // a VALU copy of the accumulator element: the hazard recogniser inserts the wait states an MFMA result needs before the
        // compiler's own first read, not before an inline-asm read -- sel_gt must see a register a VALU instruction wrote
        const float d = __builtin_canonicalizef(D[r]);
        // (written as compare + v_cndmask: left to itself the compiler turns the two selects into exec-masked branches,
        // ~10 scalar instructions per accumulator element)
        const uint32_t c_ns = sel_gt(d, secv[r], cols[r], (cols[r] & 0xffffu) | ((uint32_t)t << 16));
        cols[r] = sel_gt(d, bestv[r], c_ns, (cols[r] << 16) | (uint32_t)t);
        secv[r] = __builtin_amdgcn_fmed3f(bestv[r], secv[r], d);     // second largest of {best, second, d} (second <= best)
        bestv[r] = __builtin_amdgcn_fmed3f(bestv[r], d, INFINITY);    // max(best, d)
      }
    }
    __syncthreads();
  }
  // merge the two-entry lists over the 32 lanes of each half (columns); ties -> lowest column, like torch.amin's value (the
  // index is ours).  The lists of two lanes hold disjoint columns: the merged second is the loser of the two firsts or the
  // winner's own second.
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = bestv[r], sv = secv[r];
    int c = (int)(cols[r] & 0xffffu) * NN_BT + m, sc = (int)(cols[r] >> 16) * NN_BT + m;
    c = min(c, N2 - 1); sc = min(sc, N2 - 1);            // a lane whose columns all lie beyond N2 never updated: keep it in range
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(v, o), sv2 = __shfl_xor(sv, o);
      const int c2 = __shfl_xor(c, o), sc2 = __shfl_xor(sc, o);
      const bool other = v2 > v || (v2 == v && c2 < c);          // the other lane's first wins
      const float lv = other ? v : v2, wsv = other ? sv2 : sv;     // loser of the firsts, winner's second
      const int lc = other ? c : c2, wsc = other ? sc2 : sc;
      const bool ls = lv > wsv || (lv == wsv && lc < wsc);
      sv = ls ? lv : wsv; sc = ls ? lc : wsc;
      v = other ? v2 : v; c = other ? c2 : c;
    }
    const int i = row0 + 8 * (r >> 2) + 4 * h + (r & 3);
    if (m == 0 && i < N1) { best_j[i] = c; second_j[i] = sv == -INFINITY ? c : sc; }    // N2 = 1: no second candidate
  }
}

// ---- 3. exact fp32 cosine of the matched pairs + deterministic mean ------------------------------------------------------
__global__ __launch_bounds__(256) void nnfm_finish_kernel(const float* __restrict__ f1, const float* __restrict__ f2, int C, int N1,
                                                          int N2, const float* __restrict__ inv1, const float* __restrict__ inv2,
                                                          int32_t* __restrict__ best_j, const int32_t* __restrict__ second_j,
                                                          float* __restrict__ cosv, float* __restrict__ partial) {
  __shared__ float s_red[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float d = 0.f;
  if (i < N1) {
    const int ja = best_j[i], jb = second_j[i];
    float da = 0.f, db = 0.f;
    for (int c = 0; c < C; ++c) {
      const float a = f1[(size_t)c * N1 + i];
      da = fmaf(a, f2[(size_t)c * N2 + ja], da);
      db = fmaf(a, f2[(size_t)c * N2 + jb], db);
    }
    const float ca = da * inv1[i] * inv2[ja], cb = db * inv1[i] * inv2[jb];
    const bool second = cb > ca || (cb == ca && jb < ja);          // the fp32 cosines decide; ties -> lowest column
    const float cs = second ? cb : ca;
    if (second) best_j[i] = jb;                                    // the backward follows the match
    cosv[i] = cs;
    d = 1.0f - cs;
  }
  s_red[threadIdx.x] = d;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = s_red[0];
}

__global__ __launch_bounds__(256) void nnfm_sum_kernel(const float* __restrict__ partial, int nb, int N1, float* __restrict__ loss) {
  __shared__ double s_red[256];
  double acc = 0.0;
  for (int k = threadIdx.x; k < nb; k += 256) acc += (double)partial[k];
  s_red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(s_red[0] / (double)N1);
}

// ---- 4. backward through the arg-min -----------------------------------------------------------------------------------
// d(1 - cos_i)/d f1_i = -( f2_j inv2_j inv1_i - cos_i f1_i inv1_i^2 ),  times g / N1
__global__ __launch_bounds__(256) void nnfm_bwd_kernel(const float* __restrict__ f1, const float* __restrict__ f2, int C, int N1, int N2,
                                                       const float* __restrict__ inv1, const float* __restrict__ inv2,
                                                       const int32_t* __restrict__ best_j, const float* __restrict__ cosv,
                                                       const float* __restrict__ g, float* __restrict__ d_f1) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N1) return;
  const int j = best_j[i];
  const float s = g[0] / (float)N1, i1 = inv1[i], i2 = inv2[j], cs = cosv[i];
  const float ka = -s * i2 * i1, kb = s * cs * i1 * i1;
  for (int c = blockIdx.y; c < C; c += gridDim.y)
    d_f1[(size_t)c * N1 + i] = fmaf(ka, f2[(size_t)c * N2 + j], kb * f1[(size_t)c * N1 + i]);
}

}  // namespace trase

using namespace trase;

extern "C" {

// workspace: A bf16 [N1][C] | B bf16 [N2][C] | inv1 [N1] | inv2 [N2] | best_j [N1] | cos [N1] | partial [ceil(N1/256)] | second_j [N1]
static size_t nn_off(int C, int N1, int N2, size_t off[8]) {
  size_t o = 0;
  off[0] = o; o += align_up(sizeof(__bf16) * (size_t)N1 * C);
  off[1] = o; o += align_up(sizeof(__bf16) * (size_t)N2 * C);
  off[2] = o; o += align_up(sizeof(float) * (size_t)N1);
  off[3] = o; o += align_up(sizeof(float) * (size_t)N2);
  off[4] = o; o += align_up(sizeof(int32_t) * (size_t)N1);
  off[5] = o; o += align_up(sizeof(float) * (size_t)N1);
  off[6] = o; o += align_up(sizeof(float) * (size_t)((N1 + 255) / 256));
  off[7] = o; o += align_up(sizeof(int32_t) * (size_t)N1);
  return o;
}

static int nn_check(int C, int N1, int N2) {
  if (C < 16 || C > NN_MAXC || (C % 64) != 0) { set_error("nnfm: C = %d channels (supported: multiples of 64 up to %d)", C, NN_MAXC); return TRASE_ERR_UNSUPPORTED; }
  if (N1 < 1 || N2 < 1) { set_error("nnfm: empty feature maps"); return TRASE_ERR_INVALID; }
  if (N2 > 65536 * NN_BT) { set_error("nnfm: N2 = %d style positions (supported: up to %d)", N2, 65536 * NN_BT); return TRASE_ERR_UNSUPPORTED; }
  return TRASE_OK;
}

int trase_nnfm_sizes(int32_t C, int32_t N1, int32_t N2, size_t* ws_bytes) {
  int rc = nn_check(C, N1, N2);
  if (rc) return rc;
  if (!ws_bytes) { set_error("nnfm_sizes: null"); return TRASE_ERR_INVALID; }
  size_t off[8];
  *ws_bytes = nn_off(C, N1, N2, off);
  return TRASE_OK;
}

int trase_nnfm_forward(const float* feat1, const float* feats2, int32_t C, int32_t N1, int32_t N2, float* loss, void* ws,
                       size_t ws_bytes, int32_t device, trase_stream_t stream_) {
  int rc = nn_check(C, N1, N2);
  if (rc) return rc;
  size_t off[8];
  if (!feat1 || !feats2 || !loss || !ws || ws_bytes < nn_off(C, N1, N2, off)) { set_error("nnfm_forward: bad arguments / workspace too small"); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  LaunchCtx c{stream, 0, 0};
  char* w = (char*)ws;
  __bf16* A = (__bf16*)(w + off[0]); __bf16* B = (__bf16*)(w + off[1]);
  float* inv1 = (float*)(w + off[2]); float* inv2 = (float*)(w + off[3]);
  int32_t* bj = (int32_t*)(w + off[4]); float* cosv = (float*)(w + off[5]); float* partial = (float*)(w + off[6]);
  int32_t* sj = (int32_t*)(w + off[7]);
  { ProfScope ps("nnfm_prep", stream);
    hipLaunchKernelGGL(nnfm_prep_kernel, dim3((N1 + 63) / 64), dim3(256), 0, stream, feat1, C, N1, A, inv1);
    hipLaunchKernelGGL(nnfm_prep_kernel, dim3((N2 + 63) / 64), dim3(256), 0, stream, feats2, C, N2, B, inv2); }
  TRASE_POST_LAUNCH("nnfm_prep", stream, c.debug);
  { ProfScope ps("nnfm_match", stream);
    const dim3 grid((N1 + 32 * NN_WPB - 1) / (32 * NN_WPB)), block(NN_WPB * WAVE);
    switch (C / 64) {
      case 1: hipLaunchKernelGGL(nnfm_match_kernel<4>, grid, block, 0, stream, A, N1, B, N2, bj, sj); break;
      case 2: hipLaunchKernelGGL(nnfm_match_kernel<8>, grid, block, 0, stream, A, N1, B, N2, bj, sj); break;
      case 3: hipLaunchKernelGGL(nnfm_match_kernel<12>, grid, block, 0, stream, A, N1, B, N2, bj, sj); break;
      case 4: hipLaunchKernelGGL(nnfm_match_kernel<16>, grid, block, 0, stream, A, N1, B, N2, bj, sj); break;
      case 5: hipLaunchKernelGGL(nnfm_match_kernel<20>, grid, block, 0, stream, A, N1, B, N2, bj, sj); break;
      case 6: hipLaunchKernelGGL(nnfm_match_kernel<24>, grid, block, 0, stream, A, N1, B, N2, bj, sj); break;
      case 7: hipLaunchKernelGGL(nnfm_match_kernel<28>, grid, block, 0, stream, A, N1, B, N2, bj, sj); break;
      default: hipLaunchKernelGGL(nnfm_match_kernel<32>, grid, block, 0, stream, A, N1, B, N2, bj, sj); break;
    } }
  TRASE_POST_LAUNCH("nnfm_match", stream, c.debug);
  const int nb = (N1 + 255) / 256;
  { ProfScope ps("nnfm_finish", stream);
    hipLaunchKernelGGL(nnfm_finish_kernel, dim3(nb), dim3(256), 0, stream, feat1, feats2, C, N1, N2, inv1, inv2, bj, sj, cosv, partial);
    hipLaunchKernelGGL(nnfm_sum_kernel, dim3(1), dim3(256), 0, stream, partial, nb, N1, loss); }
  TRASE_POST_LAUNCH("nnfm_finish", stream, c.debug);
  return TRASE_OK;
}

int trase_nnfm_backward(const float* feat1, const float* feats2, int32_t C, int32_t N1, int32_t N2, const float* g_loss,
                        const void* ws, size_t ws_bytes, float* dL_dfeat1, int32_t device, trase_stream_t stream_) {
  int rc = nn_check(C, N1, N2);
  if (rc) return rc;
  size_t off[8];
  if (!feat1 || !feats2 || !g_loss || !dL_dfeat1 || !ws || ws_bytes < nn_off(C, N1, N2, off)) { set_error("nnfm_backward: bad arguments / workspace too small"); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  const char* w = (const char*)ws;
  { ProfScope ps("nnfm_bwd", stream);
    hipLaunchKernelGGL(nnfm_bwd_kernel, dim3((N1 + 255) / 256, 8), dim3(256), 0, stream, feat1, feats2, C, N1, N2,
                       (const float*)(w + off[2]), (const float*)(w + off[3]), (const int32_t*)(w + off[4]), (const float*)(w + off[5]),
                       g_loss, dL_dfeat1); }
  TRASE_POST_LAUNCH("nnfm_bwd", stream, 0);
  return TRASE_OK;
}

}  // extern "C"
