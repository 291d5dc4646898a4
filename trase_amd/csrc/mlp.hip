// mlp.hip -- the per-Gaussian deformation MLP (utils/time_utils.py:60-131 DeformNetwork, called through
// scene/deform_model.py:34-35 at train.py:202-204, render.py:195, gui.py:965) as ONE fused forward kernel
// on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
// Network (default TRASE config D=8, W=256, multires=10, t_multires=10, not blender, not 6dof):
//   PE(x) 63 | PE(t) 21  -> 84 -> [Linear 256 + ReLU] x 8, layer 5 sees cat(PE, h) = 340 -> heads 3 | 4 | 3.
// Fusion: a workgroup of 4 waves owns 128 Gaussians, each wave 32 rows.  Activations never leave the CU:
// they live in LDS as bf16 (64 KiB, XOR-swizzled 16-byte chunks so that the row-per-lane fragment reads are
// conflict free), the positional encoding is regenerated in registers whenever a layer consumes it (layer 0
// and the skip layer), weights stream from L2 (1 MB of bf16 for the whole net), bias+ReLU+bf16 happen in the
// MFMA epilogue.  Waves never synchronise with each other.
// Forward only in this round (the FEATURE state, style transfer, render.py and the GUIs call the MLP under
// torch.no_grad()); the training backward is listed under "next" in DESIGN.md.
#include "common.h"

namespace trase {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MW = 256;          // hidden width
constexpr int MD = 8;            // hidden layers
constexpr int EMB = 84;          // 63 + 21
constexpr int EMBP = 96;         // padded to a multiple of 16
constexpr int SKIP = 5;          // layer whose input is cat(PE, h)
constexpr int MROWS = 32;        // rows per wave
constexpr int MWAVES = 4;
constexpr int HEADP = 32;        // 10 head outputs padded to one 32-wide MFMA block

struct MlpNet {
  const __bf16* w[MD];   // K-slice-major [Kp/16][256][16], Kp = 96 (layer 0), 352 (skip layer), 256 otherwise
  const float* b[MD];
  const __bf16* w_head;  // [256/16][32][16]: rows 0-2 warp, 3-6 rotation, 7-9 scaling, rest 0
  const float* b_head;   // [32]
};

__device__ __forceinline__ int mlp_kp(int l) { return l == 0 ? EMBP : (l == SKIP ? EMBP + MW : MW); }

// ---- weight packing: fp32 nn.Linear parameters -> bf16, padded / re-ordered -----------------------
struct MlpPackArgs {
  const float* w[MD]; const float* b[MD];
  const float* w_warp; const float* b_warp; const float* w_rot; const float* b_rot; const float* w_scale; const float* b_scale;
  __bf16* out_w[MD]; float* out_b[MD]; __bf16* out_wh; float* out_bh;
};

__global__ __launch_bounds__(256) void mlp_pack_kernel(MlpPackArgs a) {
  const int l = blockIdx.y;                    // 0..7 hidden layers, 8 = heads
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (l < MD) {
    const int kp = l == 0 ? EMBP : (l == SKIP ? EMBP + MW : MW);
    const int kin = l == 0 ? EMB : (l == SKIP ? EMB + MW : MW);
    if (idx < MW * kp) {
      const int n = idx / kp, k = idx % kp;
      float v = 0.f;
      if (l == 0) { if (k < EMB) v = a.w[l][n * kin + k]; }
      else if (l == SKIP) { if (k < EMB) v = a.w[l][n * kin + k]; else if (k >= EMBP) v = a.w[l][n * kin + EMB + (k - EMBP)]; }
      else v = a.w[l][n * kin + k];
      // K-slice-major: [k/16][n][k%16] -- a wave's fragment load for one K-step is then 1 KiB contiguous
      a.out_w[l][((size_t)(k >> 4) * MW + n) * 16 + (k & 15)] = (__bf16)v;
    }
    if (idx < MW) a.out_b[l][idx] = a.b[l][idx];
  } else {
    if (idx < HEADP * MW) {
      const int n = idx / MW, k = idx % MW;
      float v = 0.f;
      if (n < 3) v = a.w_warp[n * MW + k];
      else if (n < 7) v = a.w_rot[(n - 3) * MW + k];
      else if (n < 10) v = a.w_scale[(n - 7) * MW + k];
      a.out_wh[((size_t)(k >> 4) * HEADP + n) * 16 + (k & 15)] = (__bf16)v;
    }
    if (idx < HEADP) {
      float v = 0.f;
      if (idx < 3) v = a.b_warp[idx]; else if (idx < 7) v = a.b_rot[idx - 3]; else if (idx < 10) v = a.b_scale[idx - 7];
      a.out_bh[idx] = v;
    }
  }
}

// ---- positional encoding, generated straight into an MFMA A fragment ---------------------------------
// column order of cat(PE(x), PE(t)) as built by Embedder.embed (utils/time_utils.py:26-57):
//   x(3), then per frequency 2^f: sin(x 2^f)(3), cos(x 2^f)(3);  t, then per frequency: sin(t 2^f), cos(t 2^f)
__device__ __forceinline__ float pe_value(int c, float x0, float x1, float x2, float t) {
  if (c < 3) return c == 0 ? x0 : (c == 1 ? x1 : x2);
  if (c < 63) {
    const int q = c - 3, f = q / 6, r = q % 6, d = r % 3;
    const float v = (d == 0 ? x0 : (d == 1 ? x1 : x2)) * (float)(1 << f);
    return r < 3 ? __sinf(v) : __cosf(v);
  }
  if (c == 63) return t;
  if (c < EMB) {
    const int q = c - 64, f = q >> 1;
    const float v = t * (float)(1 << f);
    return (q & 1) ? __cosf(v) : __sinf(v);
  }
  return 0.f;
}

__device__ __forceinline__ bf16x8 pe_fragment(int c0, float x0, float x1, float x2, float t) {
  bf16x8 a;
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = (__bf16)pe_value(c0 + j, x0, x1, x2, t);
  return a;
}

// LDS activation tile of one wave: 32 rows x 256 bf16, 16-byte chunks XOR-swizzled by the row
__device__ __forceinline__ int act_off(int m, int k) {   // element offset of (row m, column k)
  const int chunk = (k >> 3) ^ (m & 15);
  return m * MW + (chunk << 3) + (k & 7);
}

__global__ __launch_bounds__(MWAVES* WAVE) void mlp_fwd_kernel(MlpNet net, const float* __restrict__ x,
                                                              const float* __restrict__ t, int t_stride, int N,
                                                              float* __restrict__ d_xyz, float* __restrict__ d_rot,
                                                              float* __restrict__ d_scale) {
  __shared__ __attribute__((aligned(16))) __bf16 s_act[MWAVES][MROWS * MW];   // 64 KiB
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  const int row0 = (blockIdx.x * MWAVES + wave) * MROWS;
  if (row0 >= N) return;
  const int gm = min(row0 + m, N - 1);
  const float x0 = x[3 * gm], x1 = x[3 * gm + 1], x2 = x[3 * gm + 2];
  const float tt = t[(size_t)gm * t_stride];
  __bf16* act = s_act[wave];
  for (int l = 0; l < MD; ++l) {
    const int kp = mlp_kp(l);
    const int emb_steps = (l == 0 || l == SKIP) ? EMBP / 16 : 0;
    const int steps = kp / 16;
    const __bf16* __restrict__ W = net.w[l];
    f32x16 acc[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    for (int ks = 0; ks < steps; ++ks) {
      bf16x8 a;
      if (ks < emb_steps) a = pe_fragment(ks * 16 + 8 * h, x0, x1, x2, tt);
      else a = *reinterpret_cast<const bf16x8*>(act + act_off(m, (ks - emb_steps) * 16 + 8 * h));
      const __bf16* wk = W + ((size_t)ks * MW + m) * 16 + 8 * h;    // column n = nb*32 + m of this N block
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(wk + (size_t)nb * 32 * 16);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nb], 0, 0, 0);
      }
    }
    // every row of this wave has been fully consumed: overwrite the tile with relu(acc + bias) as bf16
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const float* __restrict__ B = net.b[l];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const int col = nb * 32 + m;
      const float bias = B[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        act[act_off(row, col)] = (__bf16)fmaxf(acc[nb][r] + bias, 0.f);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  // heads: one 32-wide block, fp32 out
  f32x16 hacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
  for (int ks = 0; ks < MW / 16; ++ks) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(act + act_off(m, ks * 16 + 8 * h));
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(net.w_head + ((size_t)ks * HEADP + m) * 16 + 8 * h);
    hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, hacc, 0, 0, 0);
  }
  const int col = m;
  if (col < 10) {
    const float bias = net.b_head[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < N) {
        const float v = hacc[r] + bias;
        if (col < 3) d_xyz[(size_t)row * 3 + col] = v;
        else if (col < 7) d_rot[(size_t)row * 4 + (col - 3)] = v;
        else d_scale[(size_t)row * 3 + (col - 7)] = v;
      }
    }
  }
}

// ---- v3 (default) --------------------------------------------------------------------------------------
// What limited v1 (ISA + counters): 405 registers => one wave per SIMD with un-prefetched weight loads, and an
// epilogue of 128 two-byte LDS writes per layer.  v3:
//  * operand roles swapped: D^T = W_frag * act_frag^T, so a lane owns ONE batch row and its 16 results per block
//    are 4 runs of 4 consecutive features -> bias+ReLU+bf16 pack -> one 8-byte LDS write per run (32 per layer);
//  * the 256 outputs are produced in two halves of 128 (64 accumulator registers instead of 128; the first
//    half waits in 32 packed registers until the second has consumed the inputs) => 2 waves per SIMD;
//  * weight fragments are double-buffered in registers: the loads of K-step ks+1 are in flight during the
//    MFMAs of K-step ks;
//  * the positional encoding is generated branch-free (both lane halves' columns are compile-time constants).
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float pe_const(int c, float x0, float x1, float x2, float t) {   // c is a compile-time constant
  if (c < 3) return c == 0 ? x0 : (c == 1 ? x1 : x2);
  if (c < 63) {
    const int q = c - 3, f = q / 6, r = q % 6, d = r % 3;
    const float v = (d == 0 ? x0 : (d == 1 ? x1 : x2)) * (float)(1 << f);
    return r < 3 ? __sinf(v) : __cosf(v);
  }
  if (c == 63) return t;
  if (c < EMB) {
    const int q = c - 64, f = q >> 1;
    const float v = t * (float)(1 << f);
    return (q & 1) ? __cosf(v) : __sinf(v);
  }
  return 0.f;
}

template <int KS>
__device__ __forceinline__ bf16x8 pe_fragment_ct(int h, float x0, float x1, float x2, float t) {
  bf16x8 a;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float lo = pe_const(KS * 16 + j, x0, x1, x2, t);
    const float hi = pe_const(KS * 16 + 8 + j, x0, x1, x2, t);
    a[j] = (__bf16)(h ? hi : lo);
  }
  return a;
}

__device__ __forceinline__ short bf16_bits(float v) { return __builtin_bit_cast(short, (__bf16)v); }

// one K-step of a 128-wide output half: 4 MFMAs, lane = batch row
__device__ __forceinline__ void mma4(f32x16 (&acc)[4], const bf16x8 (&w)[4], const bf16x8& a) {
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nb], a, acc[nb], 0, 0, 0);
}
// fragment of K-step `kstep` (K-slice-major weights): rows row_base + nb*32 + m, columns 8h..8h+7 of the slice
__device__ __forceinline__ void load_w4(bf16x8 (&w)[4], const __bf16* __restrict__ W, int row_base, int m, int h, int kstep) {
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
    w[nb] = *reinterpret_cast<const bf16x8*>(W + ((size_t)kstep * MW + row_base + nb * 32 + m) * 16 + 8 * h);
}

template <int KS>
__device__ __forceinline__ void emb_steps_rec(f32x16 (&acc)[4], const __bf16* __restrict__ W, int row_base, int m, int h,
                                              float x0, float x1, float x2, float tt) {
  if constexpr (KS < EMBP / 16) {
    bf16x8 w[4];
    load_w4(w, W, row_base, m, h, KS);
    const bf16x8 a = pe_fragment_ct<KS>(h, x0, x1, x2, tt);
    mma4(acc, w, a);
    __builtin_amdgcn_sched_barrier(0);      // keep the 6 unrolled steps from hoisting all their loads (register blow-up)
    emb_steps_rec<KS + 1>(acc, W, row_base, m, h, x0, x1, x2, tt);
  }
}

__global__ __launch_bounds__(MWAVES* WAVE) void mlp_fwd_kernel_v3(MlpNet net, const float* __restrict__ x,
                                                                    const float* __restrict__ t, int t_stride, int N,
                                                                    float* __restrict__ d_xyz, float* __restrict__ d_rot,
                                                                    float* __restrict__ d_scale) {
  __shared__ __attribute__((aligned(16))) __bf16 s_act[MWAVES][MROWS * MW];   // 64 KiB
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  const int row0 = (blockIdx.x * MWAVES + wave) * MROWS;
  if (row0 >= N) return;
  const int grow = row0 + m;
  const int gm = min(grow, N - 1);
  const float x0 = x[3 * gm], x1 = x[3 * gm + 1], x2 = x[3 * gm + 2];
  const float tt = t[(size_t)gm * t_stride];
  __bf16* act = s_act[wave];
  for (int l = 0; l < MD; ++l) {
    const bool has_emb = (l == 0 || l == SKIP);
    const int hid_steps = (l == 0) ? 0 : MW / 16;
    const int kst0 = has_emb ? EMBP / 16 : 0;             // first K-step of the hidden columns
    const __bf16* __restrict__ W = net.w[l];
    const float* __restrict__ B = net.b[l];
    s16x4 held[4][4];                                      // first half, packed bf16: [block][run of 4 features]
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int row_base = half * 128;
      __builtin_amdgcn_sched_barrier(0);                  // the two halves must not be interleaved (register budget)
      f32x16 acc[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
      if (has_emb) emb_steps_rec<0>(acc, W, row_base, m, h, x0, x1, x2, tt);
      if (hid_steps) {
        bf16x8 w0[4], w1[4];
        load_w4(w0, W, row_base, m, h, kst0);
        for (int ks = 0; ks < hid_steps; ks += 2) {       // unrolled by two: static register double buffer
          load_w4(w1, W, row_base, m, h, kst0 + ks + 1);
          const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(act + act_off(m, ks * 16 + 8 * h));
          mma4(acc, w0, a0);
          if (ks + 2 < hid_steps) load_w4(w0, W, row_base, m, h, kst0 + ks + 2);
          const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(act + act_off(m, (ks + 1) * 16 + 8 * h));
          mma4(acc, w1, a1);
        }
      }
      // epilogue of this half: lane = batch row m; register r of block nb is feature row_base + nb*32 + 8(r/4) + 4h + r%4
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f0 = row_base + nb * 32 + 8 * q + 4 * h;
          const float4 bias = *reinterpret_cast<const float4*>(B + f0);
          s16x4 pk;
          pk[0] = bf16_bits(fmaxf(acc[nb][4 * q + 0] + bias.x, 0.f));
          pk[1] = bf16_bits(fmaxf(acc[nb][4 * q + 1] + bias.y, 0.f));
          pk[2] = bf16_bits(fmaxf(acc[nb][4 * q + 2] + bias.z, 0.f));
          pk[3] = bf16_bits(fmaxf(acc[nb][4 * q + 3] + bias.w, 0.f));
          if (half == 0) held[nb][q] = pk;                 // inputs are still needed by the second half
          else *reinterpret_cast<s16x4*>(act + act_off(m, f0)) = pk;
        }
      }
    }
    // both halves have consumed the old tile: now the first half may land
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<s16x4*>(act + act_off(m, nb * 32 + 8 * q + 4 * h)) = held[nb][q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  // heads: D^T[o][row], one block; lane = batch row, register r is output 8(r/4) + 4h + r%4
  f32x16 hacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
  for (int ks = 0; ks < MW / 16; ++ks) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(act + act_off(m, ks * 16 + 8 * h));
    const bf16x8 w = *reinterpret_cast<const bf16x8*>(net.w_head + ((size_t)ks * HEADP + m) * 16 + 8 * h);
    hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, hacc, 0, 0, 0);
  }
  if (grow < N) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {                           // outputs 0..15 live in registers 0..7 (q = 0,1)
      const int o = 8 * (r >> 2) + 4 * h + (r & 3);
      if (o < 10) {
        const float v = hacc[r] + net.b_head[o];
        if (o < 3) d_xyz[(size_t)grow * 3 + o] = v;
        else if (o < 7) d_rot[(size_t)grow * 4 + (o - 3)] = v;
        else d_scale[(size_t)grow * 3 + (o - 7)] = v;
      }
    }
  }
}

static size_t mlp_ws_bytes() {
  size_t b = 0;
  for (int l = 0; l < MD; ++l) {
    const int kp = l == 0 ? EMBP : (l == SKIP ? EMBP + MW : MW);
    b += align_up(sizeof(__bf16) * (size_t)MW * kp) + align_up(sizeof(float) * MW);
  }
  b += align_up(sizeof(__bf16) * (size_t)HEADP * MW) + align_up(sizeof(float) * HEADP);
  return b;
}

}  // namespace trase

using namespace trase;

extern "C" {

int trase_mlp_sizes(size_t* ws_bytes) {
  if (!ws_bytes) { set_error("trase_mlp_sizes: null"); return TRASE_ERR_INVALID; }
  *ws_bytes = mlp_ws_bytes();
  return TRASE_OK;
}

int trase_mlp_forward(const TraseMlpWeights* w, const float* x, const float* t, int32_t t_stride, int32_t N,
                      float* d_xyz, float* d_rotation, float* d_scaling, void* ws, size_t ws_bytes, int32_t device,
                      trase_stream_t stream_) {
  if (!w || N < 0) { set_error("trase_mlp_forward: bad arguments"); return TRASE_ERR_INVALID; }
  if (w->D != MD || w->W != MW || w->xyz_multires != 10 || w->t_multires != 10 || w->is_blender || w->is_6dof) {
    set_error("trase_mlp_forward: only the default DeformNetwork (D=8, W=256, multires=10, t_multires=10, "
              "not blender, not 6dof) is compiled in");
    return TRASE_ERR_UNSUPPORTED;
  }
  if (N == 0) return TRASE_OK;
  if (!x || !t || !d_xyz || !d_rotation || !d_scaling) { set_error("trase_mlp_forward: null pointer"); return TRASE_ERR_INVALID; }
  for (int l = 0; l < MD; ++l)
    if (!w->weight[l] || !w->bias[l]) { set_error("trase_mlp_forward: null layer %d", l); return TRASE_ERR_INVALID; }
  if (!w->w_warp || !w->b_warp || !w->w_rotation || !w->b_rotation || !w->w_scaling || !w->b_scaling) {
    set_error("trase_mlp_forward: null head"); return TRASE_ERR_INVALID;
  }
  if (!ws || ws_bytes < mlp_ws_bytes()) { set_error("trase_mlp_forward: workspace too small"); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  MlpPackArgs pa;
  MlpNet net;
  char* c = (char*)ws;
  for (int l = 0; l < MD; ++l) {
    const int kp = l == 0 ? EMBP : (l == SKIP ? EMBP + MW : MW);
    pa.w[l] = w->weight[l]; pa.b[l] = w->bias[l];
    pa.out_w[l] = (__bf16*)c; net.w[l] = (const __bf16*)c; c += align_up(sizeof(__bf16) * (size_t)MW * kp);
    pa.out_b[l] = (float*)c; net.b[l] = (const float*)c; c += align_up(sizeof(float) * MW);
  }
  pa.out_wh = (__bf16*)c; net.w_head = (const __bf16*)c; c += align_up(sizeof(__bf16) * (size_t)HEADP * MW);
  pa.out_bh = (float*)c; net.b_head = (const float*)c;
  pa.w_warp = w->w_warp; pa.b_warp = w->b_warp; pa.w_rot = w->w_rotation; pa.b_rot = w->b_rotation;
  pa.w_scale = w->w_scaling; pa.b_scale = w->b_scaling;
  {
    ProfScope ps("mlp_pack", stream);
    hipLaunchKernelGGL(mlp_pack_kernel, dim3((MW * (EMBP + MW) + 255) / 256, MD + 1), dim3(256), 0, stream, pa);
  }
  TRASE_POST_LAUNCH("mlp_pack", stream, 0);
  const int rows_per_block = MWAVES * MROWS;
  {
    ProfScope ps("mlp_fwd", stream);
    const dim3 grid((N + rows_per_block - 1) / rows_per_block), block(MWAVES * WAVE);
    if (w->variant & 1)       // v1: first-generation kernel (kept for A/B)
      hipLaunchKernelGGL(mlp_fwd_kernel, grid, block, 0, stream, net, x, t, t_stride, N, d_xyz, d_rotation, d_scaling);
    else
      hipLaunchKernelGGL(mlp_fwd_kernel_v3, grid, block, 0, stream, net, x, t, t_stride, N, d_xyz, d_rotation, d_scaling);
  }
  TRASE_POST_LAUNCH("mlp_fwd", stream, 0);
  return TRASE_OK;
}

}  // extern "C"
