// mlp.hip -- the per-Gaussian deformation MLP (utils/time_utils.py:60-131 DeformNetwork, called through
// scene/deform_model.py:34-35 at train.py:202-204, render.py:195, gui.py:965) as ONE fused forward kernel
// on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
// Network (default TRASE config D=8, W=256, multires=10, t_multires=10, not blender, not 6dof):
//   PE(x) 63 | PE(t) 21  -> 84 -> [Linear 256 + ReLU] x 8, layer 5 sees cat(PE, h) = 340 -> heads 3 | 4 | 3.
// Fusion: a workgroup of 4 waves owns 128 Gaussians, each wave 32 rows.  Activations never leave the CU:
// they live in LDS as bf16 (64 KiB, XOR-swizzled 16-byte chunks so that the row-per-lane fragment reads are
// conflict free), the positional encoding is generated once per wave into six register fragments, weights
// stream from L2 (1 MB of bf16 for the whole net), bias (as the accumulator's initial value) + ReLU + bf16
// happen in the MFMA epilogue.  Waves never synchronise with each other; two waves share a SIMD.
// Training (GAUSSIAN state, train.py:202-204 / :299): the same forward additionally saves the activations as
// "transposed images" and the ReLU gates as bits; the backward is a fused data chain of the same shape plus
// split-N MFMA GEMMs for the parameter gradients (see "training backward" below).
#include "common.h"
#include <type_traits>

namespace trase {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MW = 256;          // hidden width
constexpr int MD = 8;            // hidden layers
constexpr int EMB_T = 84;        // default network: PE(x) 63 + PE(t) 21 (t_multires = 10)
constexpr int EMB_B = 93;        // is_blender (D-NeRF): PE(x) 63 + timenet output 30 (utils/time_utils.py:74-86)
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
  const float* temb;     // is_blender: the 30 timenet outputs shared by all rows (columns 63..92); else nullptr
};

// ---- weight packing: fp32 nn.Linear parameters -> bf16, padded / re-ordered -----------------------
struct MlpPackArgs {
  const float* w[MD]; const float* b[MD];
  const float* w_warp; const float* b_warp; const float* w_rot; const float* b_rot; const float* w_scale; const float* b_scale;
  __bf16* out_w[MD]; float* out_b[MD]; __bf16* out_wh; float* out_bh;
  int emb;               // input columns of layer 0: EMB_T or EMB_B
};

__global__ __launch_bounds__(256) void mlp_pack_kernel(MlpPackArgs a) {
  const int l = blockIdx.y;                    // 0..7 hidden layers, 8 = heads
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (l < MD) {
    const int kp = l == 0 ? EMBP : (l == SKIP ? EMBP + MW : MW);
    const int EMB = a.emb;
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
//   is_blender: the time block is the timenet output instead (30 columns, the same for every row)
// LDS activation tile of one wave: 32 rows x 256 bf16, 16-byte chunks XOR-swizzled by the row
__device__ __forceinline__ int act_off(int m, int k) {   // element offset of (row m, column k)
  const int chunk = (k >> 3) ^ (m & 15);
  return m * MW + (chunk << 3) + (k & 7);
}

// ---- helpers shared by the block-GEMM kernels ------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float pe_const(int c, float x0, float x1, float x2, float t, const float* __restrict__ temb) {   // c is a compile-time constant
  if (c < 3) return c == 0 ? x0 : (c == 1 ? x1 : x2);
  if (c < 63) {
    const int q = c - 3, f = q / 6, r = q % 6, d = r % 3;
    const float v = (d == 0 ? x0 : (d == 1 ? x1 : x2)) * (float)(1 << f);
    return r < 3 ? __sinf(v) : __cosf(v);
  }
  if (temb) return c < EMB_B ? temb[c - 63] : 0.f;       // wave-uniform pointer and address: scalar loads
  if (c == 63) return t;
  if (c < EMB_T) {
    const int q = c - 64, f = q >> 1;
    const float v = t * (float)(1 << f);
    return (q & 1) ? __cosf(v) : __sinf(v);
  }
  return 0.f;
}

// ---- saved state of the training forward ---------------------------------------------------------------
// "Transposed image" of an [N][M] bf16 matrix: [tile = row/32][half = (row%32)/16][column][row%16].  A 32x32x16 MFMA
// whose reduction index is the ROW (the weight-gradient GEMMs dZ^T . input) reads its operand fragments -- one column, 8
// consecutive rows per lane -- as 16-byte loads, and the 64 lanes of a fragment load (32 columns x 2 row groups of one
// K-step) cover ONE CONTIGUOUS KILOBYTE.  (Round 1 kept the 32 rows of a column together: a fragment load then took 32
// bytes out of every 64 -- the texture-data unit spent ~87 cycles per load instruction instead of 16 and was 89 % busy
// while the matrix cores idled at 18 %, and half-used lines were fetched again: 3.2 GB of HBM reads for 2.15 GB of
// operands.)  The chain kernels hold lane = row, registers = columns: a 4x4 transpose inside every lane quad (two DPP
// exchange rounds) turns "4 columns of my row" into "4 rows of my column"; one store instruction of the wave covers
// 8 columns x 32 rows = two contiguous runs of 256 bytes.
// ReLU + bf16 of four accumulator values as packed 16-bit work: convert two floats per instruction, then max(x, 0) on the bf16
// patterns read as signed 16-bit integers (negative floats and -0 are negative integers; positive patterns are unchanged) --
// the same values as bf16(max(x, 0)), 4 instructions instead of 6, and never a -0
__device__ __forceinline__ s16x4 relu_bf16x4(float a, float b, float c, float d) {
  unsigned lo, hi;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(c), "v"(d));
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(lo) : "v"(lo));
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(hi) : "v"(hi));
  typedef unsigned u2v_ __attribute__((ext_vector_type(2)));
  const u2v_ r = {lo, hi};
  return __builtin_bit_cast(s16x4, r);
}
// the four ReLU gates (value != 0; relu_bf16x4 leaves no -0) of such a group as bits 0..3: a packed unsigned minimum with 1
// turns each half-word into 0 / 1, two shifts and ors gather the four -- 6 instructions instead of ~16
__device__ __forceinline__ unsigned gate_bits4(s16x4 pk) {
  typedef unsigned u2v_ __attribute__((ext_vector_type(2)));
  const u2v_ u = __builtin_bit_cast(u2v_, pk);
  unsigned lo, hi;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(lo) : "v"(u.x), "s"(0x00010001u));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(hi) : "v"(u.y), "s"(0x00010001u));
  unsigned x = lo | (hi << 2);                    // bits 0, 16 (e0, e1), 2, 18 (e2, e3)
  x |= x >> 15;                                   // bit 1 = e1, bit 3 = e3
  return x & 15u;
}
// dZ = dH . gate for four values: two packed conversions, then an AND with half-word masks expanded from the four gate bits
// (sign-extended one-bit fields) -- the same values as selecting per element
__device__ __forceinline__ s16x4 gated_bf16x4(float a, float b, float c, float d, unsigned bits) {
  unsigned lo, hi;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(c), "v"(d));
  const unsigned m0 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 0, 1), m1 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 1, 1);
  const unsigned m2 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 2, 1), m3 = (unsigned)__builtin_amdgcn_sbfe((int)bits, 3, 1);
  lo &= (m0 & 0xffffu) | (m1 & 0xffff0000u);
  hi &= (m2 & 0xffffu) | (m3 & 0xffff0000u);
  typedef unsigned u2v_ __attribute__((ext_vector_type(2)));
  const u2v_ r = {lo, hi};
  return __builtin_bit_cast(s16x4, r);
}
__device__ __forceinline__ size_t timg_off(int M, int col, int row) {
  return (size_t)(row >> 4) * (size_t)(M * 16) + (size_t)col * 16 + (size_t)(row & 15);
}
// The same image from the LDS activation tile (block kernels): the tile holds [row][column] with 8-byte runs of four columns;
// gfx950's transposing read (ds_read_b64_tr_b16: inside a 16-lane group lane 4 jj + cc hands in the address of four columns
// -- chunk cc -- of row jj and lane c receives column c of the four rows) IS the 4 x 4 transpose that the DPP rounds above do
// with ~14 VALU instructions per piece -- measured: those, not the store traffic, were what kept the training forward at
// 0.56 ms against 0.33 for inference.  One call moves a 16-row x 16-column patch: lane (grp, i) ends up with column c0 + i of
// the rows r0 + 4 grp .. + 3, i.e. 8 contiguous image bytes, and the wave's store covers one 512-byte run.
// rows_valid: rows of the patch that exist (< 16 only in the last tile of the batch: the rest is written as zeros).
struct PatchLane {        // lane constants of store_patch_transposed (wave block at tile rows lrow0 ..)
  int lds_elem;           // (lrow0 + 4 grp + jj) * MW + 4 (cc & 1): tile offset without the swizzled chunk term
  int sw, cch;            // swizzle key of the lane's row, chunk half of the lane's column group
  int img_elem;           // i * 16 + 4 grp: image offset inside a 16-row x 16-column patch
  int row4;               // 4 grp
};
__device__ __forceinline__ PatchLane patch_lane(int lrow0) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, i = lane & 15, jj = i >> 2, cc = i & 3;
  PatchLane p;
  p.lds_elem = (lrow0 + 4 * grp + jj) * MW + 4 * (cc & 1);
  p.sw = (4 * grp + jj) & 15;                   // lrow0 and r0 are multiples of 16
  p.cch = cc >> 1;
  p.img_elem = i * 16 + 4 * grp;
  p.row4 = 4 * grp;
  return p;
}
// r0, c0: patch origin inside the wave's block (uniform); img_patch: UNIFORM address of the patch's first column in the image
// (tile base + (row_in_tile >> 4) * MW * 16 + c0 * 16): the store then takes it as a scalar base and the lane offset as is
typedef short s16x4p __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4p patch_read(const __bf16* __restrict__ act, const PatchLane& pl, int r0, int c0) {
  typedef __attribute__((address_space(3))) s16x4p lds_s16x4;
  const int chunk = ((c0 >> 3) + pl.cch) ^ pl.sw;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(act + (pl.lds_elem + r0 * MW + (chunk << 3))));
}
__device__ __forceinline__ void patch_store(s16x4p v, const PatchLane& pl, __bf16* __restrict__ img_patch, int rows_valid) {
  if (rows_valid < 16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (pl.row4 + e < rows_valid) ? v[e] : (short)0;
  }
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  const u2v o2 = __builtin_bit_cast(u2v, v);
#ifdef TRASE_MLP_PLAIN_STORES
  *reinterpret_cast<u2v*>(img_patch + pl.img_elem) = o2;
#else
  __builtin_nontemporal_store(o2, reinterpret_cast<u2v*>(img_patch + pl.img_elem));
#endif
}

// ---- inference forward, block-GEMM organisation ---------------------------------------------------------------------
// The 32-row chain kernel makes every wave stream every weight slab from L2 on its own (16 MAC per byte through the vector
// memory path) and hides the round trip with a single K-step of lookahead; PMC shows the matrix pipe busy 36 %.  Here
// the WORKGROUP owns 128 rows and stages each 256 x 16 weight slab (8 KiB, contiguous in the K-slice-major packing) in
// LDS exactly once, double-buffered, one workgroup barrier per K-step.  Wave (wr, wc) owns rows wr*64.. (two 32-row
// groups) x output columns wc*128.. : 8 MFMAs per K-step from 4 weight + 2 activation fragments out of LDS, 128
// accumulator registers, two workgroups per CU (80 KiB of LDS each: 64 KiB activations shared by the four waves + two
// slabs).  Slab layout in LDS: [k/8][n][k%8] (two 4 KiB planes): the fragment reads of a lane half are 16-byte-contiguous
// (conflict-free), as are the staging writes.  No `held` half: the tile is rewritten after the K loop, behind a barrier.
constexpr int BRG = 2;                       // 32-row groups per wave
constexpr int BROWS = 2 * BRG * MROWS;       // rows per workgroup (128)

// encoding fragment of K-step KS for the block kernel: the time block is selected between PE(t) and the (pre-loaded,
// wave-uniform) timenet outputs without branches
template <int KS>
__device__ __forceinline__ bf16x8 blk_pe_fragment(int h, const float (&p)[4], bool blender, const float (&tb)[32]) {
  bf16x8 a;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int c = KS * 16 + 8 * hh + j;
      if (c < 63) v[hh] = pe_const(c, p[0], p[1], p[2], p[3], nullptr);
      else {
        const float pv = pe_const(c, p[0], p[1], p[2], p[3], nullptr);
        const float tv = (c < EMB_B) ? tb[c - 63] : 0.f;
        v[hh] = blender ? tv : pv;
      }
    }
    a[j] = (__bf16)(h ? v[1] : v[0]);
  }
  return a;
}

// workgroup barrier that orders LDS traffic only: __syncthreads() is a release/acquire fence and drains vmcnt too, which
// would force the slab loads that are deliberately in flight across the barrier to complete at every K-step
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0); vmcnt and expcnt left at their maxima
  __builtin_amdgcn_s_barrier();
}

// SAVE: training forward -- additionally writes, per layer, the transposed image of the post-ReLU activations and the
// ReLU gates (same saved-state layout as the chain kernel, so the backward does not care which forward produced it)
// FULL: every row of the workgroup's 128 exists (all workgroups but the last): the image stores are then unconditional --
// inside a branch the compiler's wait counts cannot rely on them having been issued, and a wait for an older weight slab then
// includes them
template <bool SAVE, bool FULL>
__device__ __forceinline__ void mlp_fwd_blk_body(__bf16* __restrict__ act, __bf16 (*s_w)[2][MW * 8], const MlpNet& net,
                                                 const float* __restrict__ x, const float* __restrict__ t, int t_stride, int N,
                                                 float* __restrict__ d_xyz, float* __restrict__ d_rot,
                                                 float* __restrict__ d_scale, __bf16* __restrict__ actsT,
                                                 uint32_t* __restrict__ gates, const int* __restrict__ ro = nullptr,
                                                 __bf16* __restrict__ peT = nullptr) {
  // ro (row order, training only): batch row r evaluates Gaussian ro[r] -- inputs are gathered and the ten outputs scattered
  // through it; everything saved for the backward (images, gates) is in BATCH order (see "dead rows" below)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  const int wr = wave & 1, wc = wave >> 1;
  const int lrow0 = wr * (BRG * MROWS);        // first row of this wave inside the tile
  const int row0 = blockIdx.x * BROWS + lrow0;
  float px[BRG][4];                            // x, y, z, t of this lane's row in each group: the encoding is generated
#pragma unroll                                 // on the fly in the six encoding K-steps of layers 0 and 5
  for (int g = 0; g < BRG; ++g) {
    int gm = min(row0 + 32 * g + m, N - 1);
    if (ro) gm = ro[gm];
    px[g][0] = x[3 * gm]; px[g][1] = x[3 * gm + 1]; px[g][2] = x[3 * gm + 2]; px[g][3] = t[(size_t)gm * t_stride];
  }
  const bool blender = net.temb != nullptr;
  float tb[32];                                // is_blender: the 30 shared timenet outputs (scalar loads); else unused zeros
  {
    const float* tp = blender ? net.temb : net.b_head;     // always a valid address: the loads need no branch
#pragma unroll
    for (int i = 0; i < 32; ++i) tb[i] = (i < EMB_B - 63) ? tp[i] : 0.f;
  }
  const int sn = threadIdx.x;                  // thread i stages output column n = i of a slab (two 16-byte pieces)
  uint4 sa0, sa1, sb0, sb1;                    // slab registers -- sa: odd slabs, sb: even slabs (plain scalars: registers)
  {                                            // slabs 0 and 1 of the first layer; later layers get theirs from the
    const __bf16* src = net.w[0] + (size_t)sn * 16;          // last K-step pair of the layer before
    sb0 = *reinterpret_cast<const uint4*>(src); sb1 = *reinterpret_cast<const uint4*>(src + 8);
    sa0 = *reinterpret_cast<const uint4*>(src + (size_t)MW * 16); sa1 = *reinterpret_cast<const uint4*>(src + (size_t)MW * 16 + 8);
  }
  // SAVE: the transposed image of a layer's activations is NOT stored in that layer's epilogue -- loads and stores share
  // vmcnt, in order, so the burst of 32 stores per wave stalled the next layer's second weight slab until it had drained
  // (0.56 ms against 0.33 for inference = exactly the store time).  The tile stays in LDS as the next layer's A operand;
  // the wave re-reads its own 32 pieces from there, four per K-step pair, while the next layer multiplies.
  const PatchLane pl = patch_lane(lrow0);
  const int u_row0 = __builtin_amdgcn_readfirstlane(row0), u_wc = __builtin_amdgcn_readfirstlane(wc);   // wave-uniform: scalar registers
  // (reads and stores of a group of pieces are issued apart -- the reads before a K-step pair, the stores after it: a store
  // right behind its read waits out the LDS latency 32 times per layer)
  auto drain_read = [&](int piece) {
    return patch_read(act, pl, (piece >> 3) * 16, u_wc * 128 + (piece & 7) * 16);
  };
  auto drain_store = [&](int l_prev, int piece, s16x4p v) {
    const int r0 = (piece >> 3) * 16, c0 = u_wc * 128 + (piece & 7) * 16;    // 16-row x 16-column patch of this wave's block
    const int growb = u_row0 + r0;
    __bf16* const patch = actsT + ((size_t)l_prev * ((N + 31) >> 5) + (growb >> 5)) * (MW * 32) +
                          (size_t)((growb >> 4) & 1) * (MW * 16) + (size_t)c0 * 16;
    if constexpr (FULL) patch_store(v, pl, patch, 16);
    else if ((growb & ~31) < N) patch_store(v, pl, patch, min(max(N - growb, 0), 16));   // the 32-row image tile exists
  };
  if constexpr (SAVE) {
    // The encoding as a transposed image (operand of the weight-gradient GEMMs of layers 0 and 5; a kernel of its own cost 0.04 ms
    // per step for this).  Done HERE, before any accumulator is live (inside layer 0's K-steps the same code spilled): the two
    // waves of a row block share the six K-steps (wc = 0: columns 0..47, wc = 1: 48..95), park their fragments in the idle
    // activation tile -- each wave inside the half of the tile that only its own epilogue overwrites -- and take them out
    // through the same transposing patch reads as every other image.
    auto park = [&](auto ks_c) {
      constexpr int KS = decltype(ks_c)::value;
#pragma unroll
      for (int g = 0; g < BRG; ++g)
        *reinterpret_cast<bf16x8*>(act + act_off(lrow0 + 32 * g + m, (KS % 3) * 16 + 8 * h + 128 * (KS / 3))) =
            blk_pe_fragment<KS>(h, px[g], blender, tb);
    };
    if (u_wc == 0) { park(std::integral_constant<int, 0>{}); park(std::integral_constant<int, 1>{}); park(std::integral_constant<int, 2>{}); }
    else { park(std::integral_constant<int, 3>{}); park(std::integral_constant<int, 4>{}); park(std::integral_constant<int, 5>{}); }
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // (every wave reads back only what it wrote itself)
    constexpr int PP = BRG * 2 * 3;                          // 16-row x 16-column patches per wave: 4 row pieces x 3 column pieces
#pragma unroll
    for (int j0 = 0; j0 < PP; j0 += 6) {
      s16x4p pv[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) pv[j] = patch_read(act, pl, ((j0 + j) / 3) * 16, u_wc * 128 + ((j0 + j) % 3) * 16);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int growb = u_row0 + ((j0 + j) / 3) * 16;
        __bf16* const patch = peT + (size_t)(growb >> 5) * (EMBP * 32) + (size_t)((growb >> 4) & 1) * (EMBP * 16) +
                              (size_t)(u_wc * 48 + ((j0 + j) % 3) * 16) * 16;
        if constexpr (FULL) patch_store(pv[j], pl, patch, 16);
        else if ((growb & ~31) < N) patch_store(pv[j], pl, patch, min(max(N - growb, 0), 16));
      }
    }
#pragma unroll
    for (int g = 0; g < BRG; ++g) asm volatile("" : "+v"(px[g][0]), "+v"(px[g][1]), "+v"(px[g][2]), "+v"(px[g][3]));   // (no reuse of these values below)
  }
  for (int l = 0; l < MD; ++l) {
    const bool has_emb = (l == 0 || l == SKIP);
    const int emb_k = has_emb ? EMBP / 16 : 0;
    const int steps = emb_k + ((l == 0) ? 0 : MW / 16);
    const __bf16* __restrict__ W = net.w[l];
    const __bf16* __restrict__ Wn = net.w[min(l + 1, MD - 1)];
    const float* __restrict__ B = net.b[l];
    f32x16 acc[BRG][4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bias = *reinterpret_cast<const float4*>(B + wc * 128 + nb * 32 + 8 * q + 4 * h);
#pragma unroll
        for (int g = 0; g < BRG; ++g) {
          acc[g][nb][4 * q + 0] = bias.x; acc[g][nb][4 * q + 1] = bias.y; acc[g][nb][4 * q + 2] = bias.z; acc[g][nb][4 * q + 3] = bias.w;
        }
      }
    // slab pipeline: slab ks is in LDS buffer ks&1 during K-step ks; its global loads were issued TWO K-steps earlier
    // (registers sa / sb alternate), it was parked in LDS at the end of K-step ks-1, behind that step's barrier
    // slab ks of this layer; past the end: slabs 0 / 1 of the next layer (every layer has an even number of K-steps)
    auto slab_load = [&](int ks, uint4& r0, uint4& r1) {
      const __bf16* src = (ks < steps ? W + (size_t)ks * MW * 16 : Wn + (size_t)(ks - steps) * MW * 16) + (size_t)sn * 16;
      r0 = *reinterpret_cast<const uint4*>(src); r1 = *reinterpret_cast<const uint4*>(src + 8);
    };
    auto slab_park = [&](int buf, const uint4& r0, const uint4& r1) {
      *reinterpret_cast<uint4*>(&s_w[buf][0][sn * 8]) = r0;
      *reinterpret_cast<uint4*>(&s_w[buf][1][sn * 8]) = r1;
    };
    slab_park(0, sb0, sb1);                                  // slab 0 (requested during the previous layer's last K-steps)
    lds_barrier();
    auto mma = [&](int buf, const bf16x8 (&a)[BRG]) {
      const __bf16* wl = &s_w[buf][h][(wc * 128) * 8];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(wl + (nb * 32 + m) * 8);
#pragma unroll
        for (int g = 0; g < BRG; ++g) acc[g][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a[g], acc[g][nb], 0, 0, 0);
      }
    };
    // a pair of K-steps (ks even): the loop is written two steps at a time so that sa / sb are static registers
    auto kpair = [&](int ks, const bf16x8 (&a0)[BRG], const bf16x8 (&a1)[BRG], auto&& between) {
      slab_load(ks + 2, sb0, sb1);
      mma(0, a0);
      between();                                             // (SAVE: image stores -- early in the pair, so that the slab
      slab_park(1, sa0, sa1);                                // loads issued after them have ~1.5 pairs to get past them)
      lds_barrier();
      slab_load(ks + 3, sa0, sa1);
      mma(1, a1);
      if (ks + 2 < steps) slab_park(0, sb0, sb1);            // slab ks+2
      lds_barrier();
    };
    auto pe_frag = [&](int ks, int g) -> bf16x8 {
      switch (ks) {                                          // folds after unrolling: the columns are compile-time constants
        case 0: return blk_pe_fragment<0>(h, px[g], blender, tb);
        case 1: return blk_pe_fragment<1>(h, px[g], blender, tb);
        case 2: return blk_pe_fragment<2>(h, px[g], blender, tb);
        case 3: return blk_pe_fragment<3>(h, px[g], blender, tb);
        case 4: return blk_pe_fragment<4>(h, px[g], blender, tb);
        default: return blk_pe_fragment<5>(h, px[g], blender, tb);
      }
    };
    if (has_emb) {
      // opaque to loop-invariant code motion: otherwise the 2 x 96 encoding values of layers 0 and 5 are computed once
      // before the layer loop and kept (= spilled) across the hidden layers
#pragma unroll
      for (int g = 0; g < BRG; ++g) asm volatile("" : "+v"(px[g][0]), "+v"(px[g][1]), "+v"(px[g][2]), "+v"(px[g][3]));
#pragma unroll
      for (int ks = 0; ks < EMBP / 16; ks += 2) {
        const bf16x8 a0[BRG] = {pe_frag(ks, 0), pe_frag(ks, 1)}, a1[BRG] = {pe_frag(ks + 1, 0), pe_frag(ks + 1, 1)};
        kpair(ks, a0, a1, [] {});
      }
    }
    for (int ks = emb_k; ks < steps; ks += 2) {
      bf16x8 a0[BRG], a1[BRG];
#pragma unroll
      for (int g = 0; g < BRG; ++g) {
        a0[g] = *reinterpret_cast<const bf16x8*>(act + act_off(lrow0 + 32 * g + m, (ks - emb_k) * 16 + 8 * h));
        a1[g] = *reinterpret_cast<const bf16x8*>(act + act_off(lrow0 + 32 * g + m, (ks + 1 - emb_k) * 16 + 8 * h));
      }
      constexpr int DP = BRG * 16 / (MW / 32);             // image pieces of the previous layer per K-step pair (4)
      const int p0 = ((ks - emb_k) >> 1) * DP;
      s16x4p dv[DP];
      if constexpr (SAVE) {                                  // (l >= 1 here: layer 0 has no K-steps of this kind)
#pragma unroll
        for (int j = 0; j < DP; ++j) dv[j] = drain_read(p0 + j);
      }
      kpair(ks, a0, a1, [&] {
        if constexpr (SAVE) {
#pragma unroll
          for (int j = 0; j < DP; ++j) drain_store(l - 1, p0 + j, dv[j]);
        }
      });
    }
    // epilogue: ReLU, bf16, this wave's 64 x 128 block of the tile (all reads of the old tile are behind the last barrier)
#pragma unroll
    for (int g = 0; g < BRG; ++g) {
      const int grow = row0 + 32 * g + m;
      unsigned gate[2] = {0u, 0u};                           // SAVE: this lane's 64 ReLU gates of the layer
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f0 = wc * 128 + nb * 32 + 8 * q + 4 * h;
          const s16x4 pk = relu_bf16x4(acc[g][nb][4 * q], acc[g][nb][4 * q + 1], acc[g][nb][4 * q + 2], acc[g][nb][4 * q + 3]);
          *reinterpret_cast<s16x4*>(act + act_off(lrow0 + 32 * g + m, f0)) = pk;
          if constexpr (SAVE) gate[nb >> 1] |= gate_bits4(pk) << ((nb & 1) * 16 + q * 4);
        }
      if constexpr (SAVE)      // the row's 256 gate bits are two uint4 (h = 0 / 1); this wave owns words 2 wc, 2 wc + 1 of each
        if (FULL || grow < N) *reinterpret_cast<uint2*>(gates + (((size_t)l * N + grow) * 2 + h) * 4 + wc * 2) = uint2{gate[0], gate[1]};
    }
    lds_barrier();             // the tile is complete; the image stores and the next layer's slab loads stay in flight
  }
  if constexpr (SAVE) {
    for (int p0 = 0; p0 < BRG * 16; p0 += 8) {
      s16x4p dv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) dv[j] = drain_read(p0 + j);
#pragma unroll
      for (int j = 0; j < 8; ++j) drain_store(MD - 1, p0 + j, dv[j]);
    }
  }
  if (wc != 0) return;                                       // heads: one wave per 64 rows
  f32x16 hacc[BRG];
#pragma unroll
  for (int g = 0; g < BRG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) hacc[g][r] = 0.f;
  for (int ks = 0; ks < MW / 16; ++ks) {
    const bf16x8 w = *reinterpret_cast<const bf16x8*>(net.w_head + ((size_t)ks * HEADP + m) * 16 + 8 * h);
#pragma unroll
    for (int g = 0; g < BRG; ++g) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(act + act_off(lrow0 + 32 * g + m, ks * 16 + 8 * h));
      hacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, hacc[g], 0, 0, 0);
    }
  }
  // the ten outputs of a row sit in two lanes (h = 0: outputs 0-3, 8, 9; h = 1: 4-7): one cross-half exchange of register 3
  // (v_permlane32_swap) gives lane h = 0 the rows of d_xyz and d_scaling and lane h = 1 the row of d_rotation -- three wide
  // stores per row instead of ten dwords (with a row order every dword store touched a line of its own)
#pragma unroll
  for (int g = 0; g < BRG; ++g) {
    int grow = row0 + 32 * g + m;
    const bool ok = grow < N;
    if (ok && ro) grow = ro[grow];
    float o[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) o[r] = hacc[g][r] + net.b_head[8 * (r >> 2) + 4 * h + (r & 3)];
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[3]), __float_as_uint(o[3]), false, false);
    const float other3 = __uint_as_float(h ? sw[0] : sw[1]);       // h = 0 receives output 7, h = 1 receives output 3
    if (ok) {
      if (h == 0) {
        float* px3 = d_xyz + (size_t)grow * 3;
        px3[0] = o[0]; px3[1] = o[1]; px3[2] = o[2];
        float* ps3 = d_scale + (size_t)grow * 3;
        ps3[0] = other3; ps3[1] = o[4]; ps3[2] = o[5];
      } else {
        *reinterpret_cast<float4*>(d_rot + (size_t)grow * 4) = make_float4(other3, o[0], o[1], o[2]);
      }
    }
  }
}

__global__ __launch_bounds__(MWAVES* WAVE) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mlp_fwd_kernel_blk(MlpNet net, const float* __restrict__ x, const float* __restrict__ t, int t_stride, int N,
                        float* __restrict__ d_xyz, float* __restrict__ d_rot, float* __restrict__ d_scale) {
  __shared__ __attribute__((aligned(16))) __bf16 act[BROWS * MW];                   // 64 KiB
  __shared__ __attribute__((aligned(16))) __bf16 s_w[2][2][MW * 8];                 // [buffer][k half][n][8]: 16 KiB
  mlp_fwd_blk_body<false, false>(act, s_w, net, x, t, t_stride, N, d_xyz, d_rot, d_scale, nullptr, nullptr);
}

__global__ __launch_bounds__(MWAVES* WAVE) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mlp_fwd_train_kernel_blk(MlpNet net, const float* __restrict__ x, const float* __restrict__ t, int t_stride, int N,
                              float* __restrict__ d_xyz, float* __restrict__ d_rot, float* __restrict__ d_scale,
                              __bf16* __restrict__ actsT, uint4* __restrict__ gates, const int* __restrict__ ro,
                              __bf16* __restrict__ peT) {
  __shared__ __attribute__((aligned(16))) __bf16 act[BROWS * MW];
  __shared__ __attribute__((aligned(16))) __bf16 s_w[2][2][MW * 8];
  if ((int)(blockIdx.x + 1) * BROWS <= N)
    mlp_fwd_blk_body<true, true>(act, s_w, net, x, t, t_stride, N, d_xyz, d_rot, d_scale, actsT, reinterpret_cast<uint32_t*>(gates), ro, peT);
  else
    mlp_fwd_blk_body<true, false>(act, s_w, net, x, t, t_stride, N, d_xyz, d_rot, d_scale, actsT, reinterpret_cast<uint32_t*>(gates), ro, peT);
}

// ---- training backward ------------------------------------------------------------------------------------
// (1) data chain:  dZ_l = dH_l * [h_l > 0];  dH_{l-1} = dZ_l . W_l[:, hidden columns]  (x and t are detached at the
//     call site, train.py:196-204 `deform.step(gaussians.get_xyz.detach(), time_input)`: nothing flows into the PE).
//     Same fused structure as the forward: a wave owns 32 rows, the dZ tile lives in swizzled LDS, the TRANSPOSED
//     weights stream K-slice-major from L2, lane = batch row.  Every dZ_l leaves as a transposed image.
// (2) parameter gradients: dW_l = dZ_l^T . input_l, a GEMM whose reduction runs over the N rows.  A workgroup owns a
//     contiguous range of row tiles and the whole M x NK output (accumulators never leave the registers), reads
//     both operands as transposed images, and writes one fp32 partial; db_l comes from one extra MFMA against a
//     fragment of ones.  (3) a small kernel sums the partials into the caller's gradient tensors.
struct MlpNetT {
  const __bf16* wt[MD];   // l >= 1: [256/16][256][16], wt[l][k'=feature f][n=input column j] = W_l[f][hidden_off + j]
  const __bf16* wt_head;  // [1][256][16]: [n = j][k' = o] = W_head[o][j], o < 10
};

struct MlpPackTArgs {
  const float* w[MD];
  const float* w_warp; const float* w_rot; const float* w_scale;
  __bf16* out_wt[MD]; __bf16* out_wth;
  int emb;
};

__global__ __launch_bounds__(256) void mlp_pack_t_kernel(MlpPackTArgs a) {
  const int l = blockIdx.y;                    // 1..7 hidden layers, 0 = heads
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= 1) {
    if (idx < MW * MW) {
      const int f = idx / MW, j = idx % MW;
      const int kin = l == SKIP ? a.emb + MW : MW, off = l == SKIP ? a.emb : 0;
      a.out_wt[l][((size_t)(f >> 4) * MW + j) * 16 + (f & 15)] = (__bf16)a.w[l][f * kin + off + j];
    }
  } else if (idx < MW * 16) {
    const int j = idx >> 4, o = idx & 15;
    float v = 0.f;
    if (o < 3) v = a.w_warp[o * MW + j];
    else if (o < 7) v = a.w_rot[(o - 3) * MW + j];
    else if (o < 10) v = a.w_scale[(o - 7) * MW + j];
    a.out_wth[j * 16 + o] = (__bf16)v;
  }
}

// Dead rows.  A Gaussian that the view culled (radii == 0: a quarter of them on the S4 orbit, more with a camera inside the cloud)
// hands the network an exactly-zero cotangent: its dZ rows are zero in every layer and it adds nothing to any parameter gradient.
// The backward therefore works on the LIVE 32-row tiles only: mlp_tile_flags / mlp_tile_list compact the ids of the tiles that
// hold at least one row with a non-zero cotangent (ascending, deterministic); a chain workgroup takes four consecutive list
// entries (its four 32-row groups need not be neighbours), the weight-gradient GEMMs walk the same list, and the dZ images of
// dead tiles are neither written nor read.  Skipping works on whole tiles, so it pays when dead rows come in runs: the training
// forward can evaluate the rows in a caller-given order (`ro`, trase_amd.deform sorts the Gaussians along a Morton curve: culling
// is spatially coherent) -- batch row r is Gaussian ro[r], the saved state is in batch order, cotangents are gathered through ro.
__global__ __launch_bounds__(256) void mlp_tile_flags_kernel(const float* __restrict__ g_xyz, const float* __restrict__ g_rot,
                                                             const float* __restrict__ g_scale, int N,
                                                             const int* __restrict__ ro, int* __restrict__ flags) {
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (tile >= ((N + 31) >> 5)) return;
  const int row = tile * 32 + (lane & 31);
  bool nz = false;
  if (row < N) {
    const size_t sr = ro ? ro[row] : row;
    auto live = [](float v) { return (__float_as_uint(v) & 0x7fffffffu) != 0u; };      // -0 is zero; NaN / Inf are kept
    if (lane < 32) {
      if (g_xyz) nz = live(g_xyz[3 * sr]) || live(g_xyz[3 * sr + 1]) || live(g_xyz[3 * sr + 2]);
      if (g_scale) nz = nz || live(g_scale[3 * sr]) || live(g_scale[3 * sr + 1]) || live(g_scale[3 * sr + 2]);
    } else if (g_rot) nz = live(g_rot[4 * sr]) || live(g_rot[4 * sr + 1]) || live(g_rot[4 * sr + 2]) || live(g_rot[4 * sr + 3]);
  }
  const bool any = __ballot(nz) != 0ull;
  if (lane == 0) flags[tile] = any ? 1 : 0;
}

// one workgroup: ascending list of the live tile ids + their count.  Thread i owns the contiguous chunk [i c, (i + 1) c) of the
// flags (all of its loads in flight at once), one workgroup-wide scan of the chunk counts places the chunks.
__global__ __launch_bounds__(1024) void mlp_tile_list_kernel(const int* __restrict__ flags, int tiles, int* __restrict__ live,
                                                             int* __restrict__ n_live) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int CH = 16;                                     // flags per thread and pass: 16 K tiles = 524 k Gaussians per pass
  int base = 0;
  for (int t0 = 0; t0 < tiles; t0 += 1024 * CH) {
    const int first = t0 + tid * CH;
    unsigned bits = 0;
#pragma unroll
    for (int k = 0; k < CH; ++k) bits |= (first + k < tiles && flags[first + k] != 0) ? (1u << k) : 0u;
    const int cnt = __popc(bits);
    int incl = cnt;                                          // inclusive scan over the wave (DPP row shifts + broadcasts)
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int pre = base, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { if (w < wave) pre += wsum[w]; total += wsum[w]; }
    int at = pre + incl - cnt;
#pragma unroll
    for (int k = 0; k < CH; ++k)
      if (bits & (1u << k)) live[at++] = first + k;
    base += total;
    __syncthreads();
  }
  if (tid == 0) *n_live = base;
}

// backward data chain on the block-GEMM organisation of mlp_fwd_blk_body: the workgroup owns four live 32-row tiles and stages
// every slab of the TRANSPOSED weights in LDS once; wave (wr, wc) owns the groups 2 wr, 2 wr + 1 x input columns wc*128..; the dZ
// tile lives in the shared 64 KiB activation tile; every dZ_l leaves as a transposed image.  The head stage is a single K-step
// whose four weight fragments each wave reads straight from L2.
__global__ __launch_bounds__(MWAVES* WAVE) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mlp_bwd_data_kernel_blk(MlpNetT net, const float* __restrict__ g_xyz, const float* __restrict__ g_rot,
                             const float* __restrict__ g_scale, int N, const uint32_t* __restrict__ gates,
                             __bf16* __restrict__ dzT, __bf16* __restrict__ gT, const int* __restrict__ ro,
                             const int* __restrict__ live_list, const int* __restrict__ n_live_ptr) {
  __shared__ __attribute__((aligned(16))) __bf16 act[BROWS * MW];                   // 64 KiB
  __shared__ __attribute__((aligned(16))) __bf16 s_w[2][2][MW * 8];                 // 16 KiB
  const int n_live = *n_live_ptr;
  if ((int)blockIdx.x * (BROWS / 32) >= n_live) return;                             // (the grid covers the all-live case)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  const int wr = wave & 1, wc = wave >> 1;
  const int lrow0 = wr * (BRG * MROWS);
  const PatchLane pl = patch_lane(lrow0);
  const int u_wr = __builtin_amdgcn_readfirstlane(wr), u_wc = __builtin_amdgcn_readfirstlane(wave >> 1);
  const int tiles = (N + 31) >> 5;
  const int sn = threadIdx.x;
  int tg[BRG];                                                // tile of each 32-row group of this wave (wave-uniform), -1 = none
#pragma unroll
  for (int gi = 0; gi < BRG; ++gi) {
    const int k = (int)blockIdx.x * (BROWS / 32) + BRG * u_wr + gi;
    tg[gi] = k < n_live ? live_list[k] : -1;
  }
  // slabs 0 and 1 of the first hidden stage (l = MD - 1) are requested before the head stage
  uint4 sa0, sa1, sb0, sb1;
  {
    const __bf16* src = net.wt[MD - 1] + (size_t)sn * 16;
    sb0 = *reinterpret_cast<const uint4*>(src); sb1 = *reinterpret_cast<const uint4*>(src + 8);
    sa0 = *reinterpret_cast<const uint4*>(src + (size_t)MW * 16); sa1 = *reinterpret_cast<const uint4*>(src + (size_t)MW * 16 + 8);
  }
  // cotangent of the ten head outputs as one 16-wide K-step: columns 0-2 d_xyz, 3-6 rotation, 7-9 scaling
  bf16x8 g8[BRG];
  bool live[BRG];
  int gmr[BRG];
#pragma unroll
  for (int gi = 0; gi < BRG; ++gi) {
    const int grow = max(tg[gi], 0) * 32 + m;                 // batch row
    const int gb = min(grow, N - 1);
    live[gi] = tg[gi] >= 0 && grow < N; gmr[gi] = gb;
    const size_t gm = ro ? ro[gb] : gb;                       // the Gaussian behind it
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = 0.f;
    if (live[gi]) {
      if (h == 0) {
        if (g_xyz) { g[0] = g_xyz[3 * gm]; g[1] = g_xyz[3 * gm + 1]; g[2] = g_xyz[3 * gm + 2]; }
        if (g_rot) { g[3] = g_rot[4 * gm]; g[4] = g_rot[4 * gm + 1]; g[5] = g_rot[4 * gm + 2]; g[6] = g_rot[4 * gm + 3]; }
        if (g_scale) g[7] = g_scale[3 * gm];
      } else if (g_scale) { g[0] = g_scale[3 * gm + 1]; g[1] = g_scale[3 * gm + 2]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) g8[gi][j] = (__bf16)g[j];
    // transposed image of the cotangent, [tile][2][32 columns][16 rows] (columns 10..31 zero): operand of the head GEMM
    if (wc == 0 && tg[gi] >= 0) {
      __bf16* gt = gT + (size_t)tg[gi] * (HEADP * 32);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        gt[timg_off(HEADP, 8 * h + j, m)] = g8[gi][j];
        gt[timg_off(HEADP, 16 + 8 * h + j, m)] = (__bf16)0.f;
      }
    }
  }
  auto dz_read = [&](int piece) { return patch_read(act, pl, (piece >> 3) * 16, u_wc * 128 + (piece & 7) * 16); };
  auto dz_store = [&](int img, int piece, s16x4p v) {
    const int r0 = (piece >> 3) * 16, c0 = u_wc * 128 + (piece & 7) * 16;
    const int tile = tg[r0 >> 5], half = (r0 >> 4) & 1;
    if (tile >= 0)
      patch_store(v, pl, dzT + ((size_t)img * tiles + tile) * (MW * 32) + (size_t)half * (MW * 16) + (size_t)c0 * 16,
                  min(max(N - (tile * 32 + half * 16), 0), 16));
  };
  for (int l = MD; l >= 1; --l) {                         // produces dZ_{l-1}; l == MD is the head stage
    // this stage's ReLU gates (recorded by the forward) are requested now and used in the epilogue
    uint2 gv[BRG];
#pragma unroll
    for (int gi = 0; gi < BRG; ++gi)
      gv[gi] = *reinterpret_cast<const uint2*>(gates + (((size_t)(l - 1) * N + gmr[gi]) * 2 + h) * 4 + wc * 2);
    f32x16 acc[BRG][4];
#pragma unroll
    for (int gi = 0; gi < BRG; ++gi)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[gi][nb][r] = 0.f;
    if (l == MD) {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(net.wt_head + (size_t)(wc * 128 + nb * 32 + m) * 16 + 8 * h);
#pragma unroll
        for (int gi = 0; gi < BRG; ++gi) acc[gi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, g8[gi], acc[gi][nb], 0, 0, 0);
      }
    } else {
      const __bf16* __restrict__ W = net.wt[l];
      const __bf16* __restrict__ Wn = net.wt[max(l - 1, 1)];
      constexpr int steps = MW / 16;
      auto slab_load = [&](int ks, uint4& r0, uint4& r1) {   // past the end: slabs 0 / 1 of the next stage
        const __bf16* src = (ks < steps ? W + (size_t)ks * MW * 16 : Wn + (size_t)(ks - steps) * MW * 16) + (size_t)sn * 16;
        r0 = *reinterpret_cast<const uint4*>(src); r1 = *reinterpret_cast<const uint4*>(src + 8);
      };
      auto slab_park = [&](int buf, const uint4& r0, const uint4& r1) {
        *reinterpret_cast<uint4*>(&s_w[buf][0][sn * 8]) = r0;
        *reinterpret_cast<uint4*>(&s_w[buf][1][sn * 8]) = r1;
      };
      slab_park(0, sb0, sb1);
      lds_barrier();
      auto mma = [&](int buf, const bf16x8 (&a)[BRG]) {
        const __bf16* wl = &s_w[buf][h][(wc * 128) * 8];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const bf16x8 w = *reinterpret_cast<const bf16x8*>(wl + (nb * 32 + m) * 8);
#pragma unroll
          for (int gi = 0; gi < BRG; ++gi) acc[gi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a[gi], acc[gi][nb], 0, 0, 0);
        }
      };
      for (int ks = 0; ks < steps; ks += 2) {
        bf16x8 a0[BRG], a1[BRG];
#pragma unroll
        for (int gi = 0; gi < BRG; ++gi) {
          a0[gi] = *reinterpret_cast<const bf16x8*>(act + act_off(lrow0 + 32 * gi + m, ks * 16 + 8 * h));
          a1[gi] = *reinterpret_cast<const bf16x8*>(act + act_off(lrow0 + 32 * gi + m, (ks + 1) * 16 + 8 * h));
        }
        slab_load(ks + 2, sb0, sb1);
        mma(0, a0);
        slab_park(1, sa0, sa1);
        lds_barrier();
        slab_load(ks + 3, sa0, sa1);
        mma(1, a1);
        if (ks + 2 < steps) slab_park(0, sb0, sb1);
        lds_barrier();
      }
    }
    // epilogue: ReLU gate recorded by the forward, bf16, LDS tile for the next stage + transposed image
#pragma unroll
    for (int gi = 0; gi < BRG; ++gi) {
      const unsigned gate[2] = {gv[gi].x, gv[gi].y};
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f0 = wc * 128 + nb * 32 + 8 * q + 4 * h;
          const unsigned bits = live[gi] ? gate[nb >> 1] >> ((nb & 1) * 16 + q * 4) : 0u;
          const s16x4 pk = gated_bf16x4(acc[gi][nb][4 * q], acc[gi][nb][4 * q + 1], acc[gi][nb][4 * q + 2], acc[gi][nb][4 * q + 3], bits);
          *reinterpret_cast<s16x4*>(act + act_off(lrow0 + 32 * gi + m, f0)) = pk;
        }
    }
    // the image of this wave's 64 x 128 block, from the tile it has just written (its own region: an LDS wait is enough).
    // (Spreading these stores over the next stage's K-steps, as the training forward does, needs registers this kernel does not
    // have: 272 bytes of scratch.)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int p0 = 0; p0 < 32; p0 += 8) {                     // eight reads in flight, then their stores
      s16x4p dv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) dv[j] = dz_read(p0 + j);
#pragma unroll
      for (int j = 0; j < 8; ++j) dz_store(l - 1, p0 + j, dv[j]);
    }
    lds_barrier();
  }
}

// ---- parameter-gradient GEMM: out[f][k] = sum over rows of A[row][f] * B[row][k] ------------------------------
struct WgradJob {
  const __bf16* A;     // transposed image, M columns
  const __bf16* B;     // transposed image, NK columns
  float* partial;      // [G][M][NK]
  float* bias_partial; // [G][M] or null
};
constexpr int WG_MAX_JOBS = 8;
// distance between the partial planes of consecutive row groups, in floats: NOT the bare M * NK -- the reduction reads the
// same offset of every plane, and planes a power of two apart put all of those reads on the same memory channels
#ifndef WG_PLANE_PAD
#define WG_PLANE_PAD 1088
#endif
__host__ __device__ constexpr size_t wg_plane(int M, int NK) { return (size_t)M * NK + WG_PLANE_PAD; }
struct WgradJobs { WgradJob j[WG_MAX_JOBS]; };

// ---- the hidden-layer weight-gradient GEMM with its operand stream parked in LDS ----------------------------------
// The GEMM is bound by the HBM stream of its two images (128 flop per byte against the chip's ~400), and what limits the stream
// is the bytes it can keep in flight.  The round-2..4 kernel kept its operand fragments in registers: 480 accumulator + fragment
// registers left a ring of three K-steps (48 KB unique per CU, every fragment requested twice -- by the two waves that share
// it), 0.469 ms for the hidden layers (profiles/r5_ab_experiments.txt; removed in round 5).  Here the fragments go
// global -> LDS directly (global_load_lds_dwordx4: no registers; a fragment load of the transposed image is ONE contiguous
// kilobyte, and the LDS-DMA writes lane L's 16 bytes at base + 16 L -- exactly the 16 bytes lane L wants back), each fragment
// is requested ONCE per workgroup (wave w loads fragments 4 w .. 4 w + 3 of the K-step's sixteen), and the ring is NST
// K-steps deep (16 KB each): 144 KB in flight per CU.  One workgroup barrier per K-step; LDS-DMA completion is counted by hand
// (hipcc does not see the asm loads): s_waitcnt vmcnt(4 (NST - 2)) = "my four loads of the next K-step have landed".
// STREAM: non-temporal policy -- per operand, for an image that the launch reads exactly once: both images of the hidden layers'
// jobs (0.404 -> 0.354 ms, 6.07 TB/s) and the dZ image of the encoding jobs (0.086 -> 0.069); the encoding image (shared by both
// encoding jobs) and the short head job lose with it
template <bool STREAM>
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  if constexpr (STREAM)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// MT / NT: 32-column blocks of the A / B image (M = 32 MT output rows, NK = 32 NT output columns); WM x WN waves, each owning
// (MT / WM) x (NT / WN) blocks; NST ring slots of one K-step (LPW = ceil((MT + NT) / 4) fragments per wave: the surplus ones
// re-request the last fragment into a spare kilobyte, so that every wave counts the same number of loads per K-step).
// Schedule of K-step s: wait for MY loads of K-step s + 1, barrier (everybody's have landed; everybody holds K-step s in
// registers, so slot s is free), request K-step s + NST into slot s, read K-step s + 1 from LDS into the second register set,
// multiply K-step s -- the LDS latency of the next fragments hides under this step's MFMAs (one wave per SIMD: nobody else would).
template <int MT, int NT, int WM, int WN, int NST, bool STREAM_A, bool STREAM_B>
__global__ __launch_bounds__(256) void mlp_wgrad_lds_kernel(WgradJobs jobs, const int* __restrict__ live_list,
                                                            const int* __restrict__ n_live_ptr, int G) {
  static_assert(WM * WN == 4 && MT % WM == 0 && NT % WN == 0, "four waves");
  constexpr int M = MT * 32, NK = NT * 32, MB = MT / WM, NB = NT / WN, F = MT + NT, LPW = (F + 3) / 4, SLOT = LPW * 4 * 1024;
  static_assert(NST * SLOT <= 160 * 1024 && LPW * (NST - 2) < 64, "LDS ring / vmcnt range");
  __shared__ __attribute__((aligned(1024))) unsigned char ring[NST * SLOT];
  const WgradJob job = jobs.j[blockIdx.y];
  const int g = blockIdx.x;
  const int tiles = *n_live_ptr;                           // the reduction runs over the LIVE row tiles (see "dead rows")
  const int t_begin = (int)((long long)tiles * g / G), t_end = (int)((long long)tiles * (g + 1) / G);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const bool do_bias = job.bias_partial != nullptr && wn == 0;
  // MB x NB blocks of 32 x 32 (the hidden layers: 16 blocks = all 256 accumulation registers).  The bias sums (db_l = column
  // sums of dZ_l) do NOT take another MFMA block per row of blocks as in the register-fragment kernel of rounds 2-4 (64 more accumulators: the allocator
  // then rotates ~100 registers between the two register files in every K-step): a lane holds 8 rows of one column per
  // fragment -- four packed bf16 dot products against (1, 1) add them up in fp32; the two row halves meet in the epilogue.
  f32x16 acc[MB][NB];
  float bsum[MB];
#pragma unroll
  for (int a = 0; a < MB; ++a) {
    bsum[a] = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  }
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 one2 = {(__bf16)1.0f, (__bf16)1.0f};
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const unsigned lds0 = (unsigned)(size_t)(lds_u8*)ring;
  // loader role: wave w brings fragments w LPW .. w LPW + LPW - 1 of the K-step's F (A image first, then B)
  const __bf16* lsrc[LPW];
  size_t ltile[LPW], lhalf[LPW];
  bool l_is_a[LPW];
#pragma unroll
  for (int b = 0; b < LPW; ++b) {
    const int fid = min(wave * LPW + b, F - 1);
    const bool isA = fid < MT;
    l_is_a[b] = isA;
    lsrc[b] = (isA ? job.A + (size_t)fid * 512 : job.B + (size_t)(fid - MT) * 512) + (size_t)i * 16 + 8 * h;
    ltile[b] = isA ? (size_t)M * 32 : (size_t)NK * 32;
    lhalf[b] = isA ? (size_t)M * 16 : (size_t)NK * 16;
  }
  const unsigned my_slot_off = (unsigned)wave * (unsigned)(LPW * 1024);
  const int s_begin = 2 * t_begin, s_end = 2 * t_end;    // always an even number of steps
  auto issue = [&](int slot, int step, int tile_id) {    // this wave's fragments of the K-step -> ring slot
    const size_t tile = (size_t)tile_id;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)slot * (unsigned)SLOT + my_slot_off);
#pragma unroll
    for (int b = 0; b < LPW; ++b) {
      const void* src = lsrc[b] + tile * ltile[b] + (size_t)(step & 1) * lhalf[b];
      if constexpr (STREAM_A == STREAM_B) glds16<STREAM_A>(src, dst + b * 1024u);
      else if (l_is_a[b] ? STREAM_A : STREAM_B) glds16<true>(src, dst + b * 1024u);      // (wave-uniform)
      else glds16<false>(src, dst + b * 1024u);
    }
  };
  if (s_begin < s_end) {
    // (steps past the end re-request the last one: the count of loads in flight is then the same in every K-step)
    for (int k = 0; k < NST; ++k) { const int st = min(s_begin + k, s_end - 1); issue(k, st, live_list[st >> 1]); }
    // the tile id of the NEXT request is fetched a K-step ahead (a scalar load in front of its use costs its round trip per step)
    int tile_nx = live_list[min(s_begin + NST, s_end - 1) >> 1];
    const unsigned char* rd = ring + lane * 16;
    bf16x8 fa[2][MB], fb[2][NB];
    auto frags = [&](int set, int slot) {
      const unsigned char* st = rd + slot * SLOT;
#pragma unroll
      for (int a = 0; a < MB; ++a) fa[set][a] = *reinterpret_cast<const bf16x8*>(st + (MB * wm + a) * 1024);
#pragma unroll
      for (int b = 0; b < NB; ++b) fb[set][b] = *reinterpret_cast<const bf16x8*>(st + (MT + NB * wn + b) * 1024);
    };
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(LPW * (NST - 1)) : "memory");     // K-step s_begin is in LDS
    frags(0, 0);
    int slot = 0;                                        // ring slot of K-step s
    auto step = [&](int s, int cur) {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(LPW * (NST - 2)) : "memory");
      issue(slot, min(s + NST, s_end - 1), tile_nx);
      tile_nx = live_list[min(s + 1 + NST, s_end - 1) >> 1];
      slot = slot == NST - 1 ? 0 : slot + 1;
      frags(cur ^ 1, slot);
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][a], fb[cur][b], acc[a][b], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const bf16x2 v = {fa[cur][a][e], fa[cur][a][e + 1]};
          bsum[a] = __builtin_amdgcn_fdot2_f32_bf16(v, one2, bsum[a], false);
        }
    };
#pragma nounroll
    for (int s = s_begin; s < s_end; s += 2) {
      step(s, 0);
      step(s + 1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the surplus requests of the last K-steps must not outlive the workgroup's LDS
  }
  // D[f_local][k_local]: lane = k_local (+32 for the odd f quads), register r = f_local%4 + 4*(f_local/8)
  float* out = job.partial + (size_t)g * wg_plane(M, NK);
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = (wm * MB + a) * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        out[(size_t)f * NK + (wn * NB + b) * 32 + i] = acc[a][b][r];
      }
#pragma unroll
  for (int a = 0; a < MB; ++a) {                         // column f = 32 (wm MB + a) + i: rows 8 h .. of every K-step in this lane
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(bsum[a]), __float_as_uint(bsum[a]), false, false);
    const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);     // (half 0) + (half 1), the same order in both lanes
    if (do_bias && h == 0) job.bias_partial[(size_t)g * M + (wm * MB + a) * 32 + i] = tot;
  }
}

struct WreduceJob {
  const float* partial; const float* bias_partial;   // [G][M][NK], [G][M] (or null)
  float* out; float* bias_out;                        // out[f * stride + col_off + k], k < k_valid, f < m_valid
  int M, NK, stride, col_off, k_valid, m_valid, G, head;
  float* head_w[3]; float* head_b[3];                 // head job: rows 0-2 / 3-6 / 7-9 go to three tensors
};
constexpr int WR_MAX_JOBS = 12;
struct WreduceJobs { WreduceJob j[WR_MAX_JOBS]; };

__global__ __launch_bounds__(256) void mlp_wreduce_kernel(WreduceJobs jobs) {
  const WreduceJob job = jobs.j[blockIdx.y];
  const int total = job.M * job.NK;                      // a multiple of 4 (NK is a multiple of 32)
  const size_t plane = wg_plane(job.M, job.NK);
  // lane = four consecutive outputs (one 16-byte load per partial plane), wave w = the planes g = w (mod 4): the narrow jobs
  // have 128 planes and few outputs -- with one thread walking all planes of an output the kernel's run time was that chain
  // of dependent round trips.  Up to twelve planes in flight per thread; the four waves' sums meet in LDS and are added in a
  // fixed order -> bit-reproducible.
  __shared__ float4 sh[3][64];
  const int q = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int idx = (blockIdx.x * 64 + q) * 4;
  const int f = idx / job.NK, k = idx % job.NK;
  const bool mine = idx < total && f < job.m_valid && k < job.k_valid;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (mine) {
    int g = seg;
    for (; g + 44 < job.G; g += 48) {
      float4 v[12];
#pragma unroll
      for (int u = 0; u < 12; ++u) v[u] = *reinterpret_cast<const float4*>(job.partial + (size_t)(g + 4 * u) * plane + idx);
#pragma unroll
      for (int u = 0; u < 12; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    float4 v[12];
#pragma unroll
    for (int u = 0; u < 12; ++u)
      v[u] = (g + 4 * u < job.G) ? *reinterpret_cast<const float4*>(job.partial + (size_t)(g + 4 * u) * plane + idx)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 12; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  if (seg > 0) sh[seg - 1][q] = s;
  __syncthreads();
  if (seg == 0 && mine) {
    const float4 s1 = sh[0][q], s2 = sh[1][q], s3 = sh[2][q];
    const float o[4] = {(s.x + s1.x) + (s2.x + s3.x), (s.y + s1.y) + (s2.y + s3.y),
                        (s.z + s1.z) + (s2.z + s3.z), (s.w + s1.w) + (s2.w + s3.w)};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (k + e >= job.k_valid) break;
      if (job.head) {
        // (selects, not head_w[sg]: a dynamically indexed member of the by-value job record is copied to scratch)
        const int base = f < 3 ? 0 : (f < 7 ? 3 : 7);
        float* const hw = f < 3 ? job.head_w[0] : (f < 7 ? job.head_w[1] : job.head_w[2]);
        if (hw) hw[(f - base) * job.stride + k + e] = o[e];
      } else if (job.out) job.out[(size_t)f * job.stride + job.col_off + k + e] = o[e];
    }
  }
  const int bidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (job.bias_partial && bidx < job.m_valid) {
    // sixteen partials in flight (one thread walking up to 128 planes one dependent load at a time WAS this kernel's run
    // time: 90 us); four interleaved sums, fixed order
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < job.G; g += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = (g + u < job.G) ? job.bias_partial[(size_t)(g + u) * job.M + bidx] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) a4[u & 3] += v[u];
    }
    const float b = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    if (job.head) {
      const int base = bidx < 3 ? 0 : (bidx < 7 ? 3 : 7);
      float* const hb = bidx < 3 ? job.head_b[0] : (bidx < 7 ? job.head_b[1] : job.head_b[2]);
      if (hb) hb[bidx - base] = b;
    } else if (job.bias_out) job.bias_out[bidx] = b;
  }
}

// =====================================================================================================================
// Round 6: the chain kernels in a register-chained, OUTPUT-stationary organisation ("RC").
//
// Why: the block kernels above are bound by LDS throughput, not by the matrix pipe.  Per layer and CU they move 768 KB of
// fragment reads (256 B/clk), 256 KB of weight-slab stores and 128 KB of activation stores (ds_write_b128 / _b64: ~80 B/clk,
// MI355X_MICROARCH.md "LDS") = ~7 900 LDS cycles against 8 192 matrix-pipe cycles: every imperfection of the overlap is lost time
// (measured: 39 % of the bf16 peak).  Here
//   * a WAVE owns 64 rows (two 32-row groups) and ALL 256 columns of a layer; the workgroup (4 waves, one per SIMD, 256 rows)
//     shares nothing but the weights;
//   * activations never touch LDS: the fp32 accumulators of an output block, converted to bf16 in place, ARE two B fragments of
//     the next layer (the weights' K order is permuted to the accumulator layout: k = 32 (ks/2) + 16 (ks%2) + 8 (e/4) + 4 h + e%4
//     for element e of half h in K-step ks);
//   * the loop is output-stationary: for each 32-column output block all K-steps run into two accumulators (one per row group),
//     so one 1-KB weight fragment read from LDS feeds two MFMAs (0.5 KB / MFMA instead of 0.75 + the activation traffic), the
//     live accumulators are 32 registers instead of 256, and a block's epilogue (bias is the accumulator's initial value; ReLU +
//     bf16 = 2 VALU per 4 values) issues in the shadow of the next block's MFMAs;
//   * the weights are ONE linear stream of 1-KB fragments in consumption order (1008 of them for the forward), staged once per
//     256 rows through a three-slot ring of 16-fragment groups: per group one workgroup barrier, placed in the middle of the
//     previous group so that the fragment reads run ahead across group boundaries.
// LDS per layer and CU: 512 KB of reads + 128 KB of stores = ~3 700 cycles against the same 8 192 matrix-pipe cycles.
constexpr int RC_ROWS = 256;                 // rows per workgroup (4 waves x 64)
constexpr int RC_G = 16;                     // fragments per staged group (16 KB)
constexpr int RC_SLOTS = 3;                  // ring slots (groups)
constexpr int RC_PE_KS = EMBP / 16;          // 6 encoding K-steps
constexpr int RC_F_L0 = 8 * RC_PE_KS;        // fragments of layer 0 (48)
constexpr int RC_F_HID = 8 * 16;             // fragments of a hidden layer (128)
constexpr int RC_F_SKIP = 8 * (RC_PE_KS + 16);   // fragments of the skip layer (176)
constexpr int RC_F_HEAD = 16;
constexpr int RC_FWD_FRAGS = RC_F_L0 + 6 * RC_F_HID + RC_F_SKIP + RC_F_HEAD;   // 1008 = 63 groups
static_assert(RC_FWD_FRAGS % RC_G == 0, "whole groups");

// input column of element e (0..7) of lane half h in hidden K-step ks of the permuted order
__host__ __device__ constexpr int rc_kperm(int ks, int h, int e) { return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (e >> 2) + 4 * h + (e & 3); }
__host__ __device__ constexpr int rc_layer_frag0(int l) {        // first fragment of layer l in the forward stream (l == MD: heads)
  return l == 0 ? 0 : (l <= SKIP ? RC_F_L0 + (l - 1) * RC_F_HID : RC_F_L0 + (l - 2) * RC_F_HID + RC_F_SKIP);
}

struct RcNet {
  const __bf16* wstream;   // [RC_FWD_FRAGS][64 lanes][8]: MFMA A fragments (lane = h * 32 + output column of the block) in consumption order
  const float* bias;       // [MD][256] then [32] (heads)
  const float* temb;       // is_blender: the 30 shared timenet outputs; else nullptr
};

// weight packing for the RC kernels: one thread per fragment element
__global__ __launch_bounds__(256) void mlp_pack_rc_kernel(MlpPackArgs a, __bf16* __restrict__ ws, float* __restrict__ bias) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < MD * MW) bias[idx] = a.b[idx / MW][idx % MW];
  if (idx < HEADP) {
    float v = 0.f;
    if (idx < 3) v = a.b_warp[idx]; else if (idx < 7) v = a.b_rot[idx - 3]; else if (idx < 10) v = a.b_scale[idx - 7];
    bias[MD * MW + idx] = v;
  }
  if (idx >= RC_FWD_FRAGS * 512) return;
  const int frag = idx >> 9, lane = (idx >> 3) & 63, e = idx & 7, n = lane & 31, h = lane >> 5;
  const int EMB = a.emb;
  float v = 0.f;
  int l = MD;
  for (int k = MD - 1; k >= 0; --k) if (frag < rc_layer_frag0(k + 1)) l = k;
  const int f = frag - rc_layer_frag0(l);
  if (l == MD) {                                   // heads: one block, 16 hidden K-steps
    const int k = rc_kperm(f, h, e);
    if (n < 3) v = a.w_warp[n * MW + k]; else if (n < 7) v = a.w_rot[(n - 3) * MW + k]; else if (n < 10) v = a.w_scale[(n - 7) * MW + k];
  } else {
    const int pe_ks = (l == 0 || l == SKIP) ? RC_PE_KS : 0;
    const int per_block = pe_ks + (l == 0 ? 0 : 16);
    const int nb = f / per_block, ks = f % per_block;
    const int row = nb * 32 + n;
    const int kin = l == 0 ? EMB : (l == SKIP ? EMB + MW : MW);
    if (ks < pe_ks) {
      const int c = 16 * ks + 8 * h + e;                               // encoding columns in natural order
      if (c < EMB) v = a.w[l][row * kin + c];
    } else {
      v = a.w[l][row * kin + (l == SKIP ? EMB : 0) + rc_kperm(ks - pe_ks, h, e)];
    }
  }
  ws[idx] = (__bf16)v;
}

// One wave's view of the weight stream: ring slot addresses are compile-time offsets from the lane's base
struct RcStream {
  const __bf16* g;          // global stream + this thread's staging offset
  unsigned char* ring;      // LDS ring base
  uint4 st[4];              // staging registers: this wave's four fragments of the group being fetched
};

// 8-byte slot permutation of the 64-byte rows of a wave's transposition scratch (SAVE): slot ^ rc_swz(row) makes both the
// epilogue's ds_write_b64 (16 lanes = 16 rows, one slot) and the transposing reads (32 lanes = 8 rows x 4 slots) conflict-free
__device__ __forceinline__ int rc_swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1) | (((row >> 2) & 1) << 2); }

// SAVE: training forward -- additionally writes, per layer, the transposed image of the post-ReLU activations, the ReLU gates as
// bits and the encoding image: the SAME saved state as mlp_fwd_blk_body (the backward does not care which forward produced it).
// FULL: every row of the workgroup's 256 exists (image stores unconditional).
// G: 32-row groups per wave -- 2: four waves per workgroup, one per SIMD (a weight fragment read feeds two MFMAs); 1: eight waves, two per
// SIMD (each wave half the registers; the second wave's VALU / LDS / VMEM instructions issue under the first one's MFMAs)
template <bool SAVE, bool FULL, int G = 2>
__device__ __forceinline__ void mlp_fwd_rc_body(unsigned char* __restrict__ ring, float* __restrict__ s_bias, const RcNet& net,
                                                const float* __restrict__ x, const float* __restrict__ t, int t_stride, int N,
                                                float* __restrict__ d_xyz, float* __restrict__ d_rot, float* __restrict__ d_scale,
                                                const int* __restrict__ ro, unsigned char* __restrict__ pe_store,
                                                unsigned char* __restrict__ scratch = nullptr,
                                                __bf16* __restrict__ actsT = nullptr, uint32_t* __restrict__ gates = nullptr,
                                                __bf16* __restrict__ peT = nullptr) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  constexpr int WROWS = 32 * G, NWV = RC_ROWS / WROWS, FPW = RC_G / NWV, NP = 2 * G;   // rows per wave, waves, staged fragments per wave and group, epilogue pieces
  const int row0 = blockIdx.x * RC_ROWS + wave * WROWS;
  // ---- biases into LDS (8.1 KB), inputs into registers --------------------------------------------------------------
  for (int i = threadIdx.x; i < MD * MW + HEADP; i += NWV * 64) s_bias[i] = net.bias[i];
  float px[G][4];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    int gm = min(row0 + 32 * g + m, N - 1);
    if (ro) gm = ro[gm];
    px[g][0] = x[3 * gm]; px[g][1] = x[3 * gm + 1]; px[g][2] = x[3 * gm + 2]; px[g][3] = t[(size_t)gm * t_stride];
  }
  const bool blender = net.temb != nullptr;
  float tb[32];
  {
    const float* tp = blender ? net.temb : net.bias;
#pragma unroll
    for (int i = 0; i < 32; ++i) tb[i] = (i < EMB_B - 63) ? tp[i] : 0.f;
  }
  // ---- weight stream -----------------------------------------------------------------------------------------------
  // group G holds fragments 16 G .. 16 G + 15; wave w stages fragments 4 w .. 4 w + 3 of a group (lane: its 16 bytes of each)
  const __bf16* gsrc = net.wstream + (size_t)(wave * FPW) * 512 + (size_t)lane * 8;
  unsigned char* const wr = ring + (wave * FPW) * 1024 + lane * 16;      // where this lane parks its pieces inside a slot
  const unsigned char* const rd = ring + lane * 16;                      // where it reads a fragment from
  uint4 st0, st1, st2, st3;                    // (named scalars: as an array the staging registers were promoted to LDS)
  auto stage_load = [&](int grp) {
    const __bf16* p = gsrc + (size_t)grp * RC_G * 512;
    st0 = *reinterpret_cast<const uint4*>(p); st1 = *reinterpret_cast<const uint4*>(p + 512);
    if constexpr (FPW == 4) { st2 = *reinterpret_cast<const uint4*>(p + 1024); st3 = *reinterpret_cast<const uint4*>(p + 1536); }
  };
  auto stage_park = [&](int grp) {
    unsigned char* q = wr + (grp % RC_SLOTS) * (RC_G * 1024);
    *reinterpret_cast<uint4*>(q) = st0; *reinterpret_cast<uint4*>(q + 1024) = st1;
    if constexpr (FPW == 4) { *reinterpret_cast<uint4*>(q + 2048) = st2; *reinterpret_cast<uint4*>(q + 3072) = st3; }
  };
  auto frag_read = [&](int f) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(rd + ((f / RC_G) % RC_SLOTS) * (RC_G * 1024) + (f % RC_G) * 1024);
  };
  constexpr int NGRP = RC_FWD_FRAGS / RC_G;
  stage_load(0);
  stage_park(0);
  stage_load(1);
  lds_barrier();
  // fragment look-ahead: three reads in flight
  constexpr int LA = 3;
  bf16x8 wf[LA + 1];
#pragma unroll
  for (int i = 0; i < LA; ++i) wf[i] = frag_read(i);
  // called once per fragment, BEFORE its MFMAs: the group bookkeeping of the stream position f, and the look-ahead read
  auto advance = [&](int f) {
    if (f % RC_G == 0) {                       // start of group f/16: park group +1 (fetched during the previous group), fetch group +2
      const int grp = f / RC_G;
#ifndef RC_DBG_NOSTAGE
      if (grp + 1 < NGRP) stage_park(grp + 1);
      if (grp + 2 < NGRP) stage_load(grp + 2);
#else
      if (grp == 0) { stage_park(1); stage_load(2); }
      if (grp == 1) stage_park(2);
#endif
    }
    if (f % RC_G == RC_G / 2) lds_barrier();   // everybody has parked group +1 (and finished reading group -1): +1 is readable
    if (f + LA < RC_FWD_FRAGS) wf[(f + LA) % (LA + 1)] = frag_read(f + LA);
  };

  // the activations of a layer as B fragments, [row group][K-step].  Two banks alternate as a layer's input and output; one lives
  // in the ARCHITECTURAL registers and one in the ACCUMULATION registers (MFMA reads its B operand from either file; the packed
  // conversion writes architectural registers only, so the odd layers' outputs are moved over by v_accvgpr_write) -- with both
  // banks in one file the allocator has no room left for the fragment look-ahead
  bf16x8 bankA[G][16], bankB[G][16];
  // the encoding's fragments (layers 0 and 5): generated once, parked in a wave-private 12-KB piece of LDS (lane-linear, read back
  // like a weight fragment) -- 48 registers that the hidden layers would otherwise carry for nothing
  unsigned char* const pe_lds = pe_store + wave * (G * RC_PE_KS * 1024) + lane * 16;
  bf16x8 pe[G][RC_PE_KS];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    pe[g][0] = blk_pe_fragment<0>(h, px[g], blender, tb); pe[g][1] = blk_pe_fragment<1>(h, px[g], blender, tb);
    pe[g][2] = blk_pe_fragment<2>(h, px[g], blender, tb); pe[g][3] = blk_pe_fragment<3>(h, px[g], blender, tb);
    pe[g][4] = blk_pe_fragment<4>(h, px[g], blender, tb); pe[g][5] = blk_pe_fragment<5>(h, px[g], blender, tb);
#pragma unroll
    for (int k = 0; k < RC_PE_KS; ++k) *reinterpret_cast<bf16x8*>(pe_lds + (g * RC_PE_KS + k) * 1024) = pe[g][k];
  }
  auto pe_frag = [&](int g, int k) -> bf16x8 { return *reinterpret_cast<const bf16x8*>(pe_lds + (g * RC_PE_KS + k) * 1024); };
  // ---- SAVE: the wave's transposition scratch (64 rows x 64 bytes, slots permuted by rc_swz) ----------------------------
  unsigned char* const scr = SAVE ? scratch + wave * (WROWS * 64) : nullptr;
  const int tiles = (N + 31) >> 5;
  const int u_row0 = __builtin_amdgcn_readfirstlane(row0);
  PatchLane pl;                                     // (only img_elem / row4 are used: patch_store)
  int tr_off = 0;                                   // this lane's byte offset for a transposing read of patch (r0 = 0, c0 = 0)
  int tr_sw = 0;
  if constexpr (SAVE) {
    const int grp = lane >> 4, i = lane & 15, jj = i >> 2, cc = i & 3;
    pl.img_elem = i * 16 + 4 * grp; pl.row4 = 4 * grp; pl.lds_elem = 0; pl.sw = 0; pl.cch = 0;
    tr_off = (4 * grp + jj) * 64; tr_sw = rc_swz(4 * grp + jj);
    tr_off += 0 * cc;
  }
  // 16-row x 16-column patch (r0 multiple of 16, c0 in {0, 16}) of the scratch, transposed: lane (grp, i) <- column c0 + i, rows r0 + 4 grp ..
  auto scr_patch_read = [&](int r0, int c0) -> s16x4p {
    typedef __attribute__((address_space(3))) s16x4p lds_s16x4;
    const int cc = lane & 3;
    const int slot = ((c0 >> 2) + cc) ^ tr_sw;      // (r0 is a multiple of 16: rc_swz(row) does not depend on it)
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(scr + r0 * 64 + tr_off + slot * 8));
  };
  // four bf16 (columns col4 .. col4 + 3 of the scratch block, col4 a multiple of 4) of row `r` (= 32 g + m)
  auto scr_write4 = [&](int r, int col4, s16x4 v) {
    *reinterpret_cast<s16x4*>(scr + r * 64 + (((col4 >> 2) ^ rc_swz(r)) * 8)) = v;
  };
  // the image patches of the block sitting in the scratch: `which` = 0..3 -> row patch `which` (16 rows), both column patches
  auto image_store = [&](__bf16* __restrict__ img, int img_cols, int col0, int which) {
#ifdef RC_DBG_NOIMG
    return;
#endif
    const s16x4p v0 = scr_patch_read(16 * which, 0), v1 = scr_patch_read(16 * which, 16);
#ifdef RC_DBG_NOSTORE
    asm volatile("" :: "v"(v0), "v"(v1));
    return;
#endif
    const int growb = u_row0 + 16 * which;
    __bf16* const base = img + (size_t)(growb >> 5) * ((size_t)img_cols * 32) + (size_t)((growb >> 4) & 1) * ((size_t)img_cols * 16) + (size_t)col0 * 16;
    if constexpr (FULL) { patch_store(v0, pl, base, 16); patch_store(v1, pl, base + 256, 16); }
    else if ((growb & ~31) < N) {
      const int rv = min(max(N - growb, 0), 16);
      patch_store(v0, pl, base, rv); patch_store(v1, pl, base + 256, rv);
    }
  };
  if constexpr (SAVE) {
    // the encoding as a transposed image [tile][half][96 columns][16 rows]: three passes of 32 columns through the scratch
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          const s16x8 f8 = __builtin_bit_cast(s16x8, pe[g][2 * pass + k2]);       // lane (m, h): columns 16 (2 pass + k2) + 8 h + 0..7
          const s16x4 a4 = {f8[0], f8[1], f8[2], f8[3]}, b4 = {f8[4], f8[5], f8[6], f8[7]};
          scr_write4(32 * g + m, 16 * k2 + 8 * h, a4);
          scr_write4(32 * g + m, 16 * k2 + 8 * h + 4, b4);
        }
#pragma unroll
      for (int which = 0; which < NP; ++which) image_store(peT, EMBP, 32 * pass, which);
    }
  }
  uint32_t gw[G][4];                                 // SAVE: this lane's ReLU gates of the layer, [row group][word]
#ifdef RC_L2
  int l2_off[4];                                     // element offset of this lane's 8-byte piece of column group q inside a 16-row image tile
  __bf16* const actsT_l2 = actsT;
#pragma unroll
  for (int q = 0; q < 4; ++q) l2_off[q] = (m >> 4) * 4096 + h * 64 + (((m & 15) + 4 * q) & 15) * 4;
#endif
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  auto to_acc_file = [](bf16x8 v) -> bf16x8 {
    typedef int i32x4v __attribute__((ext_vector_type(4)));
    const i32x4v s4 = __builtin_bit_cast(i32x4v, v);
    int a0, a1, a2, a3;
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(a0) : "v"(s4[0]));
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(a1) : "v"(s4[1]));
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(a2) : "v"(s4[2]));
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(a3) : "v"(s4[3]));
    const i32x4v r = {a0, a1, a2, a3};
    return __builtin_bit_cast(bf16x8, r);
  };
  // Accumulators are double-buffered per output block: the epilogue of block b (ReLU + bf16 -> two B fragments of the next layer
  // per row group) is issued in FOUR pieces behind the first K-steps of block b + 1 -- one wave per SIMD hides up to five
  // single-issue instructions per MFMA (MI355X_MICROARCH.md), and a piece is ~10-20 of them behind two MFMAs.  The last block of a
  // layer finishes inside the first block of the next layer (its fragments are K-steps 14 / 15 there: needed last).
  f32x16 acc[2][G];
  // piece p = 2 g + s2 of the epilogue of the block in acc[slot]: fragment (g, K-step 2 nb + s2) of `out`
  auto epi_piece = [&](auto out_acc_c, bf16x8 (&out)[G][16], int slot, int nb, int p, int l) {
    constexpr bool OUT_ACC = decltype(out_acc_c)::value;
    const int g = p >> 1, s2 = p & 1;
    const f32x16& a = acc[slot][g];
    const s16x4 lo = relu_bf16x4(a[8 * s2], a[8 * s2 + 1], a[8 * s2 + 2], a[8 * s2 + 3]);
    const s16x4 hi = relu_bf16x4(a[8 * s2 + 4], a[8 * s2 + 5], a[8 * s2 + 6], a[8 * s2 + 7]);
#ifdef RC_L2
    if constexpr (SAVE) {
      // TIMING EXPERIMENT (profiles/r6_ab_experiments.txt): image layout [row/16][col/4][rotated row%16][col%4] written straight from the
      // registers (two 8-byte stores per piece), gate bits by four packed minima + shifts
      typedef unsigned u2v_ __attribute__((ext_vector_type(2)));
      const u2v_ ul = __builtin_bit_cast(u2v_, lo), uh = __builtin_bit_cast(u2v_, hi);
      unsigned t0, t1, t2, t3;
      asm("v_pk_min_u16 %0, %1, %2" : "=v"(t0) : "v"(ul.x), "s"(0x00010001u));
      asm("v_pk_min_u16 %0, %1, %2" : "=v"(t1) : "v"(ul.y), "s"(0x00010001u));
      asm("v_pk_min_u16 %0, %1, %2" : "=v"(t2) : "v"(uh.x), "s"(0x00010001u));
      asm("v_pk_min_u16 %0, %1, %2" : "=v"(t3) : "v"(uh.y), "s"(0x00010001u));
      const uint32_t w8 = t0 | (t1 << 1) | (t2 << 2) | (t3 << 3);
      const int pp = 2 * (nb & 1) + s2;
      if (pp == 0) gw[g][nb >> 1] = w8; else gw[g][nb >> 1] |= w8 << (4 * pp);
      __bf16* const img = actsT_l2 + (size_t)l * tiles * (MW * 32) + (size_t)((u_row0 >> 4) + 2 * g) * 4096 + (size_t)(8 * nb + 4 * s2) * 64;
#if defined(RC_DBG_L2_NOSTORE)
      asm volatile("" :: "v"(ul), "v"(uh), "v"(img));
#elif defined(RC_DBG_L2_PLAIN)
      *reinterpret_cast<u2v_*>(img + l2_off[2 * s2]) = ul;
      *reinterpret_cast<u2v_*>(img + 128 + l2_off[2 * s2 + 1]) = uh;
#else
      __builtin_nontemporal_store(ul, reinterpret_cast<u2v_*>(img + l2_off[2 * s2]));
      __builtin_nontemporal_store(uh, reinterpret_cast<u2v_*>(img + 128 + l2_off[2 * s2 + 1]));
#endif
    }
#else
    if constexpr (SAVE) {
      // registers 4 q + j (q = 2 s2, 2 s2 + 1) <-> columns 32 nb + 8 q + 4 h + j: gate bits as in mlp_fwd_blk_body (word nb / 2, bit
      // (nb & 1) 16 + 4 q + j of the (row, h) record), and the block's [row][column] image for the transposing reads
      const uint32_t b8 = gate_bits4(lo) | (gate_bits4(hi) << 4);
      const uint32_t sh = b8 << ((nb & 1) * 16 + 8 * s2);
      if ((nb & 1) == 0 && s2 == 0) gw[g][nb >> 1] = sh; else gw[g][nb >> 1] |= sh;
      scr_write4(32 * g + m, 16 * s2 + 4 * h, lo);
      scr_write4(32 * g + m, 16 * s2 + 8 + 4 * h, hi);
    }
#endif
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    if constexpr (OUT_ACC) out[g][2 * nb + s2] = to_acc_file(__builtin_bit_cast(bf16x8, v));
    else out[g][2 * nb + s2] = __builtin_bit_cast(bf16x8, v);
  };
  // SAVE: what follows the four pieces of a block -- `w` = 0..3: the image patches of row patch w (the block is in the scratch);
  // after the last block of a layer also the layer's gate words
  auto epi_image = [&](int l, int nb, int w) {
    if constexpr (SAVE) {
#ifndef RC_L2
      image_store(actsT + (size_t)l * tiles * (MW * 32), MW, 32 * nb, w);
#endif
      if (nb == 7 && w == NP - 1) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int grow = row0 + 32 * g + m;
          if (FULL || grow < N)
            *reinterpret_cast<uint4*>(gates + (((size_t)l * N + grow) * 2 + h) * 4) = make_uint4(gw[g][0], gw[g][1], gw[g][2], gw[g][3]);
        }
      }
    }
  };
  // one layer: IN -> OUT.  PE_KS encoding K-steps first (layers 0 and 5), then HID hidden K-steps out of `in`.
  // `pending(step)`: the previous layer's last block (its output bank is THIS layer's input: K-steps 14, 15) -- steps 0..3 its
  // epilogue pieces, 4..7 (SAVE) its image patches
  auto layer = [&](auto l_c, auto out_acc_c, bf16x8 (&in)[G][16], bf16x8 (&out)[G][16], auto&& pending) {
    constexpr int L = decltype(l_c)::value;
    constexpr int PE_KS = (L == 0 || L == SKIP) ? RC_PE_KS : 0, HID = L == 0 ? 0 : 16, KS = PE_KS + HID;
    constexpr int F0 = rc_layer_frag0(L);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      // the block's biases: the accumulators' initial value (lane (m, h), register 4 q + j <-> column 32 nb + 8 q + 4 h + j)
      f32x16 binit;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4v b4 = *reinterpret_cast<const f32x4v*>(s_bias + L * MW + nb * 32 + 8 * q + 4 * h);
        binit[4 * q] = b4[0]; binit[4 * q + 1] = b4[1]; binit[4 * q + 2] = b4[2]; binit[4 * q + 3] = b4[3];
      }
      const int slot = nb & 1;
      auto prev = [&](int step) {                 // the block before this one
        if (nb == 0) pending(step);
        else if (step < NP) epi_piece(out_acc_c, out, slot ^ 1, nb - 1, step, L);
        else epi_image(L, nb - 1, step - NP);
      };
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int f = F0 + nb * KS + ks;
        advance(f);
        const bf16x8 w = wf[f % (LA + 1)];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const bf16x8 a = ks < PE_KS ? pe_frag(g, ks < PE_KS ? ks : 0) : in[g][ks >= PE_KS ? ks - PE_KS : 0];
          acc[slot][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, ks == 0 ? binit : acc[slot][g], 0, 0, 0);
        }
        // the previous block's epilogue: one piece per K-step 1..4; SAVE: its image patches behind K-steps 6..9 (a short
        // block -- layer 0: six K-steps -- takes all four behind its last one)
        if (ks >= 1 && ks <= NP) prev(ks - 1);
        if (SAVE && KS >= 10 && ks >= 6 && ks < 6 + NP) prev(NP + ks - 6);
        if (SAVE && KS < 10 && ks == KS - 1) {
          prev(NP); prev(NP + 1);
          if constexpr (G == 2) { prev(NP + 2); prev(NP + 3); }
        }
      }
    }
  };
  auto none = [](int) {};
  constexpr std::true_type ACC{};
  constexpr std::false_type ARCH{};
  // the last block (7, in acc[1]) of layer l, whose output bank is `out`
  auto last_of = [&](auto out_acc_c, bf16x8 (&out)[G][16], int l) {
    return [&, l, out_acc_c](int step) { if (step < NP) epi_piece(out_acc_c, out, 1, 7, step, l); else epi_image(l, 7, step - NP); };
  };
  // bankA: accumulation file, bankB: architectural file
  layer(std::integral_constant<int, 0>{}, ACC, bankB, bankA, none);
  layer(std::integral_constant<int, 1>{}, ARCH, bankA, bankB, last_of(ACC, bankA, 0));
  layer(std::integral_constant<int, 2>{}, ACC, bankB, bankA, last_of(ARCH, bankB, 1));
  layer(std::integral_constant<int, 3>{}, ARCH, bankA, bankB, last_of(ACC, bankA, 2));
  layer(std::integral_constant<int, 4>{}, ACC, bankB, bankA, last_of(ARCH, bankB, 3));
  layer(std::integral_constant<int, 5>{}, ARCH, bankA, bankB, last_of(ACC, bankA, 4));
  layer(std::integral_constant<int, 6>{}, ACC, bankB, bankA, last_of(ARCH, bankB, 5));
  layer(std::integral_constant<int, 7>{}, ARCH, bankA, bankB, last_of(ACC, bankA, 6));
  {
    auto fin = last_of(ARCH, bankB, 7);             // (the last layer's last block: nothing left to hide it behind)
#pragma unroll
    for (int st_ = 0; st_ < (SAVE ? 2 * NP : NP); ++st_) fin(st_);
  }
  // heads: one 32-wide output block (10 used) out of bankB
  f32x16 hacc[G];
  {
    f32x16 binit;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4v b4 = *reinterpret_cast<const f32x4v*>(s_bias + MD * MW + 8 * q + 4 * h);
      binit[4 * q] = b4[0]; binit[4 * q + 1] = b4[1]; binit[4 * q + 2] = b4[2]; binit[4 * q + 3] = b4[3];
    }
    constexpr int F0 = rc_layer_frag0(MD);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      advance(F0 + ks);
      const bf16x8 w = wf[(F0 + ks) % (LA + 1)];
#pragma unroll
      for (int g = 0; g < G; ++g) hacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, bankB[g][ks], ks == 0 ? binit : hacc[g], 0, 0, 0);
    }
  }
  // the ten outputs of a row sit in two lanes (h = 0: outputs 0-3, 8, 9; h = 1: 4-7): see mlp_fwd_blk_body
#pragma unroll
  for (int g = 0; g < G; ++g) {
    int grow = row0 + 32 * g + m;
    const bool ok = grow < N;
    if (ok && ro) grow = ro[grow];
    float o[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) o[r] = hacc[g][r];
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[3]), __float_as_uint(o[3]), false, false);
    const float other3 = __uint_as_float(h ? sw[0] : sw[1]);
    if (ok) {
      if (h == 0) {
        float* px3 = d_xyz + (size_t)grow * 3;
        px3[0] = o[0]; px3[1] = o[1]; px3[2] = o[2];
        float* ps3 = d_scale + (size_t)grow * 3;
        ps3[0] = other3; ps3[1] = o[4]; ps3[2] = o[5];
      } else {
        *reinterpret_cast<float4*>(d_rot + (size_t)grow * 4) = make_float4(other3, o[0], o[1], o[2]);
      }
    }
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_fwd_rc_kernel(RcNet net, const float* __restrict__ x, const float* __restrict__ t, int t_stride, int N,
                       float* __restrict__ d_xyz, float* __restrict__ d_rot, float* __restrict__ d_scale) {
  __shared__ __attribute__((aligned(1024))) unsigned char ring[RC_SLOTS * RC_G * 1024];      // 48 KiB
  __shared__ __attribute__((aligned(16))) float s_bias[MD * MW + HEADP];
  __shared__ __attribute__((aligned(1024))) unsigned char pe_store[4 * 2 * RC_PE_KS * 1024];  // per wave: 12 encoding fragments
  mlp_fwd_rc_body<false, false>(ring, s_bias, net, x, t, t_stride, N, d_xyz, d_rot, d_scale, nullptr, pe_store);
}


__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mlp_fwd_train_rc_kernel(RcNet net, const float* __restrict__ x, const float* __restrict__ t, int t_stride, int N,
                             float* __restrict__ d_xyz, float* __restrict__ d_rot, float* __restrict__ d_scale,
                             __bf16* __restrict__ actsT, uint4* __restrict__ gates, const int* __restrict__ ro, __bf16* __restrict__ peT) {
  __shared__ __attribute__((aligned(1024))) unsigned char ring[RC_SLOTS * RC_G * 1024];      // 48 KiB
  __shared__ __attribute__((aligned(16))) float s_bias[MD * MW + HEADP];
  __shared__ __attribute__((aligned(64))) unsigned char scratch[4 * 4096];                   // per wave: 64 rows x 64 bytes
  __shared__ __attribute__((aligned(1024))) unsigned char pe_store[4 * 2 * RC_PE_KS * 1024];
  if ((int)(blockIdx.x + 1) * RC_ROWS <= N)
    mlp_fwd_rc_body<true, true>(ring, s_bias, net, x, t, t_stride, N, d_xyz, d_rot, d_scale, ro, pe_store, scratch, actsT, reinterpret_cast<uint32_t*>(gates), peT);
  else
    mlp_fwd_rc_body<true, false>(ring, s_bias, net, x, t, t_stride, N, d_xyz, d_rot, d_scale, ro, pe_store, scratch, actsT, reinterpret_cast<uint32_t*>(gates), peT);
}


// G = 1: eight waves of 32 rows, two per SIMD
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mlp_fwd_rc1_kernel(RcNet net, const float* __restrict__ x, const float* __restrict__ t, int t_stride, int N,
                        float* __restrict__ d_xyz, float* __restrict__ d_rot, float* __restrict__ d_scale) {
  __shared__ __attribute__((aligned(1024))) unsigned char ring[RC_SLOTS * RC_G * 1024];
  __shared__ __attribute__((aligned(16))) float s_bias[MD * MW + HEADP];
  __shared__ __attribute__((aligned(1024))) unsigned char pe_store[8 * RC_PE_KS * 1024];
  mlp_fwd_rc_body<false, false, 1>(ring, s_bias, net, x, t, t_stride, N, d_xyz, d_rot, d_scale, nullptr, pe_store);
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mlp_fwd_train_rc1_kernel(RcNet net, const float* __restrict__ x, const float* __restrict__ t, int t_stride, int N,
                              float* __restrict__ d_xyz, float* __restrict__ d_rot, float* __restrict__ d_scale,
                              __bf16* __restrict__ actsT, uint4* __restrict__ gates, const int* __restrict__ ro, __bf16* __restrict__ peT) {
  __shared__ __attribute__((aligned(1024))) unsigned char ring[RC_SLOTS * RC_G * 1024];
  __shared__ __attribute__((aligned(16))) float s_bias[MD * MW + HEADP];
  __shared__ __attribute__((aligned(64))) unsigned char scratch[8 * 2048];                   // per wave: 32 rows x 64 bytes
  __shared__ __attribute__((aligned(1024))) unsigned char pe_store[8 * RC_PE_KS * 1024];
  if ((int)(blockIdx.x + 1) * RC_ROWS <= N)
    mlp_fwd_rc_body<true, true, 1>(ring, s_bias, net, x, t, t_stride, N, d_xyz, d_rot, d_scale, ro, pe_store, scratch, actsT, reinterpret_cast<uint32_t*>(gates), peT);
  else
    mlp_fwd_rc_body<true, false, 1>(ring, s_bias, net, x, t, t_stride, N, d_xyz, d_rot, d_scale, ro, pe_store, scratch, actsT, reinterpret_cast<uint32_t*>(gates), peT);
}

static size_t mlp_rc_ws_bytes() {
  return align_up(sizeof(__bf16) * (size_t)RC_FWD_FRAGS * 512) + align_up(sizeof(float) * (MD * MW + HEADP));
}
static size_t mlp_ws_bytes_blk() {
  size_t b = 0;
  for (int l = 0; l < MD; ++l) {
    const int kp = l == 0 ? EMBP : (l == SKIP ? EMBP + MW : MW);
    b += align_up(sizeof(__bf16) * (size_t)MW * kp) + align_up(sizeof(float) * MW);
  }
  b += align_up(sizeof(__bf16) * (size_t)HEADP * MW) + align_up(sizeof(float) * HEADP);
  return b;
}
// both organisations' packed weights fit (the block kernels' first, the RC stream behind them): TRASE_MLP_RC switches per call
static size_t mlp_ws_bytes() { return mlp_ws_bytes_blk() + mlp_rc_ws_bytes(); }
// TRASE_MLP_RC: 1 (default) = the register-chained INFERENCE kernel of round 6 (0.248 against 0.300 ms at 300k rows), 0 = the block
// kernel (cross-check / A/B baseline).  TRASE_MLP_RC_TRAIN: 1 = the register-chained TRAINING forward; default 0 -- with the saved
// state's extra issue slots (scratch stores, gate bits, transposing reads, image stores: ~250 single-issue instructions per 32
// MFMAs where ~160 hide) it measures 0.50 ms against the block kernel's 0.44 (profiles/r6_ab_experiments.txt)
static bool mlp_use_rc() { static const bool on = [] { const char* e = getenv("TRASE_MLP_RC"); return !e || atoi(e) != 0; }(); return on; }
// TRASE_MLP_RC_G: 32-row groups per wave of the RC kernels (2 = four waves, one per SIMD: the default; 1 = eight waves, two per SIMD:
// inference -1.6 %, training forward 0.51 -> 0.43 ms = the block kernel's time -- profiles/r6_ab_experiments.txt)
static int mlp_rc_groups() { static const int g = [] { const char* e = getenv("TRASE_MLP_RC_G"); return e ? atoi(e) : 2; }(); return g; }
static bool mlp_use_rc_train() { static const bool on = [] { const char* e = getenv("TRASE_MLP_RC_TRAIN"); return e && atoi(e) != 0; }(); return on; }

// ---- buffer plans of the training pair ---------------------------------------------------------------------
struct MlpSaved {            // written by the training forward, read by the backward
  __bf16* actsT; __bf16* peT; uint4* gates; size_t bytes;
};
static MlpSaved mlp_saved_plan(void* base, int N) {
  const size_t tiles = (size_t)(N + 31) / 32;
  char* c = (char*)base;
  MlpSaved p;
  p.actsT = (__bf16*)c; c += align_up(sizeof(__bf16) * MD * tiles * MW * 32);
  p.peT = (__bf16*)c;   c += align_up(sizeof(__bf16) * tiles * EMBP * 32);
  p.gates = (uint4*)c;  c += align_up(sizeof(uint4) * (size_t)MD * N * 2);
  p.bytes = (size_t)(c - (char*)base);
  return p;
}

struct MlpBwdPlan {
  __bf16* wt[MD]; __bf16* wt_head; __bf16* dzT; __bf16* gT;
  int* tile_flags; int* live_list; int* n_live;   // "dead rows": per 32-row tile flag, compacted ids, count
  float* part_hidden; float* bias_hidden;   // 7 jobs (layers 1..7): [7][Gh][256][256], [7][Gh][256]
  float* part_pe; float* bias_pe;           // 2 jobs (layers 0, 5):  [2][Gp][256][96],  [Gp][256] (layer 0 only)
  float* part_head; float* bias_head;       // 1 job:                 [Gd][32][256],     [Gd][32]
  int Gh, Gp, Gd; size_t bytes;
};
static MlpBwdPlan mlp_bwd_plan(void* base, int N) {
  const size_t tiles = (size_t)(N + 31) / 32;
  char* c = (char*)base;
  MlpBwdPlan p;
  p.wt[0] = nullptr;
  for (int l = 1; l < MD; ++l) { p.wt[l] = (__bf16*)c; c += align_up(sizeof(__bf16) * (size_t)MW * MW); }
  p.wt_head = (__bf16*)c; c += align_up(sizeof(__bf16) * (size_t)MW * 16);
  p.dzT = (__bf16*)c; c += align_up(sizeof(__bf16) * MD * tiles * MW * 32);
  p.gT = (__bf16*)c;  c += align_up(sizeof(__bf16) * tiles * HEADP * 32);
  p.tile_flags = (int*)c; c += align_up(sizeof(int) * (tiles + 1));
  p.live_list = (int*)c;  c += align_up(sizeof(int) * (tiles + 1));
  p.n_live = (int*)c;     c += align_up(sizeof(int) * 4);
  // one workgroup per CU for the big GEMMs (252 = 7 x 36), more and shorter ones for the narrow jobs
  const int gh_max = 36;
  p.Gh = (int)(tiles < (size_t)gh_max ? tiles : gh_max); p.Gp = (int)(tiles < 128 ? tiles : 128);
  p.Gd = p.Gp;
  if (p.Gh < 1) p.Gh = p.Gp = p.Gd = 1;
  p.part_hidden = (float*)c; c += align_up(sizeof(float) * 7 * (size_t)p.Gh * wg_plane(MW, MW));
  p.bias_hidden = (float*)c; c += align_up(sizeof(float) * 7 * (size_t)p.Gh * MW);
  p.part_pe = (float*)c;     c += align_up(sizeof(float) * 2 * (size_t)p.Gp * wg_plane(MW, EMBP));
  p.bias_pe = (float*)c;     c += align_up(sizeof(float) * (size_t)p.Gp * MW);
  p.part_head = (float*)c;   c += align_up(sizeof(float) * (size_t)p.Gd * wg_plane(HEADP, MW));
  p.bias_head = (float*)c;   c += align_up(sizeof(float) * (size_t)p.Gd * HEADP);
  p.bytes = (size_t)(c - (char*)base);
  return p;
}

static int mlp_check_weights(const TraseMlpWeights* w, const char* who) {
  if (!w) { set_error("%s: null weights", who); return TRASE_ERR_INVALID; }
  if (w->D != MD || w->W != MW || w->xyz_multires != 10 || w->is_6dof || (w->is_blender ? w->t_multires != 6 : w->t_multires != 10)) {
    set_error("%s: only DeformNetwork(D=8, W=256, multires=10, not 6dof) with t_multires=10 (default) or "
              "is_blender (t_multires=6, timenet) is compiled in", who);
    return TRASE_ERR_UNSUPPORTED;
  }
  for (int l = 0; l < MD; ++l)
    if (!w->weight[l] || !w->bias[l]) { set_error("%s: null layer %d", who, l); return TRASE_ERR_INVALID; }
  if (!w->w_warp || !w->b_warp || !w->w_rotation || !w->b_rotation || !w->w_scaling || !w->b_scaling) {
    set_error("%s: null head", who); return TRASE_ERR_INVALID;
  }
  return TRASE_OK;
}

// pack the fp32 parameters into the forward layout inside `ws`
static int mlp_pack_forward(const TraseMlpWeights* w, void* ws, MlpNet& net, hipStream_t stream) {
  MlpPackArgs pa;
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
  pa.emb = w->is_blender ? EMB_B : EMB_T;
  net.temb = nullptr;
  {
    ProfScope ps("mlp_pack", stream);
    hipLaunchKernelGGL(mlp_pack_kernel, dim3((MW * (EMBP + MW) + 255) / 256, MD + 1), dim3(256), 0, stream, pa);
  }
  TRASE_POST_LAUNCH("mlp_pack", stream, 0);
  return TRASE_OK;
}

// pack the fp32 parameters into the RC weight stream (behind the block kernels' region of `ws`)
static int mlp_pack_rc(const TraseMlpWeights* w, void* ws, RcNet& net, hipStream_t stream) {
  MlpPackArgs pa;
  for (int l = 0; l < MD; ++l) { pa.w[l] = w->weight[l]; pa.b[l] = w->bias[l]; pa.out_w[l] = nullptr; pa.out_b[l] = nullptr; }
  pa.out_wh = nullptr; pa.out_bh = nullptr;
  pa.w_warp = w->w_warp; pa.b_warp = w->b_warp; pa.w_rot = w->w_rotation; pa.b_rot = w->b_rotation;
  pa.w_scale = w->w_scaling; pa.b_scale = w->b_scaling;
  pa.emb = w->is_blender ? EMB_B : EMB_T;
  char* c = (char*)ws + mlp_ws_bytes_blk();
  __bf16* stream_w = (__bf16*)c; c += align_up(sizeof(__bf16) * (size_t)RC_FWD_FRAGS * 512);
  float* bias = (float*)c;
  net.wstream = stream_w; net.bias = bias; net.temb = nullptr;
  {
    ProfScope ps("mlp_pack", stream);
    hipLaunchKernelGGL(mlp_pack_rc_kernel, dim3((RC_FWD_FRAGS * 512 + 255) / 256), dim3(256), 0, stream, pa, stream_w, bias);
  }
  TRASE_POST_LAUNCH("mlp_pack", stream, 0);
  return TRASE_OK;
}

}  // namespace trase

using namespace trase;

extern "C" {

int trase_mlp_sizes(size_t* ws_bytes) {
  if (!ws_bytes) { set_error("trase_mlp_sizes: null"); return TRASE_ERR_INVALID; }
  *ws_bytes = mlp_ws_bytes();
  return TRASE_OK;
}

int trase_mlp_train_sizes(int32_t N, size_t* fwd_ws_bytes, size_t* saved_bytes, size_t* bwd_ws_bytes) {
  if (!fwd_ws_bytes || !saved_bytes || !bwd_ws_bytes || N < 0) { set_error("trase_mlp_train_sizes: bad arguments"); return TRASE_ERR_INVALID; }
  *fwd_ws_bytes = mlp_ws_bytes();
  *saved_bytes = mlp_saved_plan(nullptr, N).bytes;
  *bwd_ws_bytes = mlp_bwd_plan(nullptr, N).bytes;
  return TRASE_OK;
}

int trase_mlp_forward(const TraseMlpWeights* w, const float* x, const float* t, int32_t t_stride, int32_t N,
                      float* d_xyz, float* d_rotation, float* d_scaling, void* ws, size_t ws_bytes, int32_t device,
                      trase_stream_t stream_) {
  if (N < 0) { set_error("trase_mlp_forward: bad arguments"); return TRASE_ERR_INVALID; }
  if (int rc = mlp_check_weights(w, "trase_mlp_forward")) return rc;
  if (N == 0) return TRASE_OK;
  if (!x || !t || !d_xyz || !d_rotation || !d_scaling) { set_error("trase_mlp_forward: null pointer"); return TRASE_ERR_INVALID; }
  if (!ws || ws_bytes < mlp_ws_bytes()) { set_error("trase_mlp_forward: workspace too small"); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  if (w->is_blender && t_stride != 0) { set_error("trase_mlp_forward: is_blender takes the timenet output (30 floats) with t_stride 0"); return TRASE_ERR_INVALID; }
  if (mlp_use_rc()) {
    RcNet rnet;
    if (int rc = mlp_pack_rc(w, ws, rnet, stream)) return rc;
    if (w->is_blender) rnet.temb = t;
    {
      ProfScope ps("mlp_fwd", stream);
      if (mlp_rc_groups() == 1)
        hipLaunchKernelGGL(mlp_fwd_rc1_kernel, dim3((N + RC_ROWS - 1) / RC_ROWS), dim3(512), 0, stream, rnet, x, t, t_stride, N, d_xyz, d_rotation, d_scaling);
      else
        hipLaunchKernelGGL(mlp_fwd_rc_kernel, dim3((N + RC_ROWS - 1) / RC_ROWS), dim3(256), 0, stream, rnet, x, t, t_stride, N, d_xyz, d_rotation, d_scaling);
    }
    TRASE_POST_LAUNCH("mlp_fwd", stream, 0);
    return TRASE_OK;
  }
  MlpNet net;
  if (int rc = mlp_pack_forward(w, ws, net, stream)) return rc;
  if (w->is_blender) net.temb = t;                         // t = the 30 timenet outputs shared by all rows
  {
    ProfScope ps("mlp_fwd", stream);
    const dim3 block(MWAVES * WAVE);
    // block-GEMM organisation: weight slabs staged in LDS once per 128-row workgroup
    hipLaunchKernelGGL(mlp_fwd_kernel_blk, dim3((N + BROWS - 1) / BROWS), block, 0, stream, net, x, t, t_stride, N, d_xyz,
                       d_rotation, d_scaling);
  }
  TRASE_POST_LAUNCH("mlp_fwd", stream, 0);
  return TRASE_OK;
}

int trase_mlp_forward_train(const TraseMlpWeights* w, const float* x, const float* t, int32_t t_stride, int32_t N,
                            float* d_xyz, float* d_rotation, float* d_scaling, void* saved, size_t saved_bytes, void* ws,
                            size_t ws_bytes, int32_t device, trase_stream_t stream_) {
  return trase_mlp_forward_train_rows(w, x, t, t_stride, N, nullptr, d_xyz, d_rotation, d_scaling, saved, saved_bytes, ws, ws_bytes,
                                      device, stream_);
}

int trase_mlp_forward_train_rows(const TraseMlpWeights* w, const float* x, const float* t, int32_t t_stride, int32_t N,
                                 const int32_t* row_order, float* d_xyz, float* d_rotation, float* d_scaling, void* saved,
                                 size_t saved_bytes, void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream_) {
  if (N < 0) { set_error("trase_mlp_forward_train: bad arguments"); return TRASE_ERR_INVALID; }
  if (int rc = mlp_check_weights(w, "trase_mlp_forward_train")) return rc;
  if (N == 0) return TRASE_OK;
  if (!x || !t || !d_xyz || !d_rotation || !d_scaling) { set_error("trase_mlp_forward_train: null pointer"); return TRASE_ERR_INVALID; }
  if (!ws || ws_bytes < mlp_ws_bytes()) { set_error("trase_mlp_forward_train: workspace too small"); return TRASE_ERR_WORKSPACE; }
  const MlpSaved sv = mlp_saved_plan(saved, N);
  if (!saved || saved_bytes < sv.bytes) { set_error("trase_mlp_forward_train: saved-state buffer too small"); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  if (w->is_blender && t_stride != 0) { set_error("trase_mlp_forward_train: is_blender takes the timenet output (30 floats) with t_stride 0"); return TRASE_ERR_INVALID; }
  if (mlp_use_rc_train()) {
    RcNet rnet;
    if (int rc = mlp_pack_rc(w, ws, rnet, stream)) return rc;
    if (w->is_blender) rnet.temb = t;
    {
      ProfScope ps("mlp_fwd_train", stream);
      if (mlp_rc_groups() == 1)
        hipLaunchKernelGGL(mlp_fwd_train_rc1_kernel, dim3((N + RC_ROWS - 1) / RC_ROWS), dim3(512), 0, stream, rnet, x, t, t_stride, N,
                           d_xyz, d_rotation, d_scaling, sv.actsT, sv.gates, (const int*)row_order, sv.peT);
      else
        hipLaunchKernelGGL(mlp_fwd_train_rc_kernel, dim3((N + RC_ROWS - 1) / RC_ROWS), dim3(256), 0, stream, rnet, x, t, t_stride, N,
                           d_xyz, d_rotation, d_scaling, sv.actsT, sv.gates, (const int*)row_order, sv.peT);
    }
    TRASE_POST_LAUNCH("mlp_fwd_train", stream, 0);
    return TRASE_OK;
  }
  MlpNet net;
  if (int rc = mlp_pack_forward(w, ws, net, stream)) return rc;
  if (w->is_blender) net.temb = t;
  {
    ProfScope ps("mlp_fwd_train", stream);
    const dim3 block(MWAVES * WAVE);
    hipLaunchKernelGGL(mlp_fwd_train_kernel_blk, dim3((N + BROWS - 1) / BROWS), block, 0, stream, net, x, t, t_stride, N,
                       d_xyz, d_rotation, d_scaling, sv.actsT, sv.gates, (const int*)row_order, sv.peT);
  }
  TRASE_POST_LAUNCH("mlp_fwd_train", stream, 0);
  return TRASE_OK;
}

int trase_mlp_backward(const TraseMlpWeights* w, int32_t N, const float* dL_dd_xyz, const float* dL_dd_rotation,
                       const float* dL_dd_scaling, const void* saved, size_t saved_bytes, const TraseMlpGrads* grads,
                       void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream_) {
  return trase_mlp_backward_rows(w, N, nullptr, dL_dd_xyz, dL_dd_rotation, dL_dd_scaling, saved, saved_bytes, grads, ws, ws_bytes,
                                 device, stream_);
}

int trase_mlp_live_tiles(const void* bwd_ws, size_t ws_bytes, int32_t N, int32_t* n_live_device, trase_stream_t stream_) {
  if (!bwd_ws || !n_live_device || N <= 0) { set_error("trase_mlp_live_tiles: bad arguments"); return TRASE_ERR_INVALID; }
  const MlpBwdPlan bp = mlp_bwd_plan(const_cast<void*>(bwd_ws), N);
  if (ws_bytes < bp.bytes) { set_error("trase_mlp_live_tiles: workspace too small"); return TRASE_ERR_WORKSPACE; }
  TRASE_CHECK(hipMemcpyAsync(n_live_device, bp.n_live, sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream_));
  return TRASE_OK;
}

int trase_mlp_backward_rows(const TraseMlpWeights* w, int32_t N, const int32_t* row_order, const float* dL_dd_xyz,
                            const float* dL_dd_rotation, const float* dL_dd_scaling, const void* saved, size_t saved_bytes,
                            const TraseMlpGrads* grads, void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream_) {
  if (N < 0 || !grads) { set_error("trase_mlp_backward: bad arguments"); return TRASE_ERR_INVALID; }
  if (int rc = mlp_check_weights(w, "trase_mlp_backward")) return rc;
  if (N == 0) { set_error("trase_mlp_backward: N == 0 (the caller zero-fills the gradients)"); return TRASE_ERR_INVALID; }
  const MlpSaved sv = mlp_saved_plan(const_cast<void*>(saved), N);
  if (!saved || saved_bytes < sv.bytes) { set_error("trase_mlp_backward: saved-state buffer too small"); return TRASE_ERR_WORKSPACE; }
  const MlpBwdPlan bp = mlp_bwd_plan(ws, N);
  if (!ws || ws_bytes < bp.bytes) { set_error("trase_mlp_backward: workspace too small"); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(device));
  const int tiles = (N + 31) / 32;
  MlpPackTArgs pa;
  MlpNetT net;
  pa.out_wt[0] = nullptr; net.wt[0] = nullptr; pa.w[0] = nullptr;
  for (int l = 1; l < MD; ++l) { pa.w[l] = w->weight[l]; pa.out_wt[l] = bp.wt[l]; net.wt[l] = bp.wt[l]; }
  pa.out_wth = bp.wt_head; net.wt_head = bp.wt_head;
  pa.w_warp = w->w_warp; pa.w_rot = w->w_rotation; pa.w_scale = w->w_scaling;
  const int EMB = w->is_blender ? EMB_B : EMB_T;           // input columns of the encoding block
  pa.emb = EMB;
  {
    ProfScope ps("mlp_pack_t", stream);
    hipLaunchKernelGGL(mlp_pack_t_kernel, dim3(MW * MW / 256, MD), dim3(256), 0, stream, pa);
  }
  TRASE_POST_LAUNCH("mlp_pack_t", stream, 0);
  {   // "dead rows": the tiles that hold a row with a non-zero cotangent, as an ascending list
    ProfScope ps("mlp_live_tiles", stream);
    hipLaunchKernelGGL(mlp_tile_flags_kernel, dim3((tiles + 3) / 4), dim3(256), 0, stream, dL_dd_xyz, dL_dd_rotation, dL_dd_scaling,
                       N, (const int*)row_order, bp.tile_flags);
    hipLaunchKernelGGL(mlp_tile_list_kernel, dim3(1), dim3(1024), 0, stream, (const int*)bp.tile_flags, tiles, bp.live_list, bp.n_live);
  }
  TRASE_POST_LAUNCH("mlp_live_tiles", stream, 0);
  {
    ProfScope ps("mlp_bwd_data", stream);
    const dim3 block(MWAVES * WAVE);
    hipLaunchKernelGGL(mlp_bwd_data_kernel_blk, dim3((N + BROWS - 1) / BROWS), block, 0, stream, net, dL_dd_xyz, dL_dd_rotation,
                       dL_dd_scaling, N, (const uint32_t*)sv.gates, bp.dzT, bp.gT, (const int*)row_order, (const int*)bp.live_list,
                       (const int*)bp.n_live);
  }
  TRASE_POST_LAUNCH("mlp_bwd_data", stream, 0);
  const size_t img = (size_t)tiles * MW * 32;              // one layer's transposed image, elements
  WreduceJobs rj;
  int nr = 0;
  auto reduce_job = [&](const float* partial, const float* bias_partial, float* out, float* bias_out, int M, int NK,
                        int stride, int col_off, int k_valid, int m_valid, int G) -> WreduceJob& {
    WreduceJob& r = rj.j[nr++];
    r.partial = partial; r.bias_partial = bias_partial; r.out = out; r.bias_out = bias_out;
    r.M = M; r.NK = NK; r.stride = stride; r.col_off = col_off; r.k_valid = k_valid; r.m_valid = m_valid; r.G = G; r.head = 0;
    for (int k = 0; k < 3; ++k) { r.head_w[k] = nullptr; r.head_b[k] = nullptr; }
    return r;
  };
  {   // hidden inputs: layers 1..7, input = activations of layer l-1 (layer 5: columns 84.. of its 340-wide weight)
    WgradJobs jobs;
    for (int l = 1; l < MD; ++l) {
      WgradJob& j = jobs.j[l - 1];
      j.A = bp.dzT + (size_t)l * img; j.B = sv.actsT + (size_t)(l - 1) * img;
      j.partial = bp.part_hidden + (size_t)(l - 1) * bp.Gh * wg_plane(MW, MW);
      j.bias_partial = bp.bias_hidden + (size_t)(l - 1) * bp.Gh * MW;
      const int kin = l == SKIP ? EMB + MW : MW;
      reduce_job(j.partial, j.bias_partial, grads->weight[l], grads->bias[l], MW, MW, kin, l == SKIP ? EMB : 0, MW, MW, bp.Gh);
    }
    ProfScope ps("mlp_wgrad_hidden", stream);
    hipLaunchKernelGGL((mlp_wgrad_lds_kernel<8, 8, 2, 2, 9, true, true>), dim3(bp.Gh, MD - 1), dim3(256), 0, stream, jobs, (const int*)bp.live_list,
                       (const int*)bp.n_live, bp.Gh);
  }
  TRASE_POST_LAUNCH("mlp_wgrad_hidden", stream, 0);
  {   // encoding inputs: layer 0 and the first 84 columns of the skip layer
    WgradJobs jobs;
    const int ls[2] = {0, SKIP};
    for (int k = 0; k < 2; ++k) {
      WgradJob& j = jobs.j[k];
      j.A = bp.dzT + (size_t)ls[k] * img; j.B = sv.peT;
      j.partial = bp.part_pe + (size_t)k * bp.Gp * wg_plane(MW, EMBP);
      j.bias_partial = k == 0 ? bp.bias_pe : nullptr;
      reduce_job(j.partial, j.bias_partial, grads->weight[ls[k]], k == 0 ? grads->bias[0] : nullptr, MW, EMBP,
                 k == 0 ? EMB : EMB + MW, 0, EMB, MW, bp.Gp);
    }
    ProfScope ps("mlp_wgrad_pe", stream);
    hipLaunchKernelGGL((mlp_wgrad_lds_kernel<8, 3, 4, 1, 12, true, false>), dim3(bp.Gp, 2), dim3(256), 0, stream, jobs, (const int*)bp.live_list,
                       (const int*)bp.n_live, bp.Gp);
  }
  TRASE_POST_LAUNCH("mlp_wgrad_pe", stream, 0);
  {   // heads: cotangent image (32 padded columns) x activations of the last layer
    WgradJobs jobs;
    WgradJob& j = jobs.j[0];
    j.A = bp.gT; j.B = sv.actsT + (size_t)(MD - 1) * img; j.partial = bp.part_head; j.bias_partial = bp.bias_head;
    WreduceJob& r = reduce_job(j.partial, j.bias_partial, nullptr, nullptr, HEADP, MW, MW, 0, MW, 10, bp.Gd);
    r.head = 1;
    r.head_w[0] = grads->w_warp; r.head_w[1] = grads->w_rotation; r.head_w[2] = grads->w_scaling;
    r.head_b[0] = grads->b_warp; r.head_b[1] = grads->b_rotation; r.head_b[2] = grads->b_scaling;
    ProfScope ps("mlp_wgrad_head", stream);
    hipLaunchKernelGGL((mlp_wgrad_lds_kernel<1, 8, 1, 4, 12, false, false>), dim3(bp.Gd, 1), dim3(256), 0, stream, jobs, (const int*)bp.live_list,
                       (const int*)bp.n_live, bp.Gd);
  }
  TRASE_POST_LAUNCH("mlp_wgrad_head", stream, 0);
  {
    ProfScope ps("mlp_wreduce", stream);
    hipLaunchKernelGGL(mlp_wreduce_kernel, dim3(MW * MW / 256, nr), dim3(256), 0, stream, rj);
  }
  TRASE_POST_LAUNCH("mlp_wreduce", stream, 0);
  return TRASE_OK;
}

}  // extern "C"
