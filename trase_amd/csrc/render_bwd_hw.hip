// render_bwd_hw.hip -- backward of the compositing stage, half-wave formulation (default for F = 32).
//
// The mathematics of render_bwd_gs.hip (lane = Gaussian, DPP scans along the list, one gradient row per (sub-tile, Gaussian)
// pair, no atomics) with both channel contractions as bf16-split MFMA GEMMs, in this shape (the 64-entry-chunk MFMA kernel of
// round 1, render_bwd_mf.hip, was removed in round 4):
//
//   * a chunk is 32 list entries: lane l = (g = l & 31, h = l >> 5).  Both lane halves hold the SAME 32
//     Gaussians; half h visits the pixels of columns 4h..4h+3 of the 8x8 sub-tile.  One instruction still handles 64
//     (pixel, Gaussian) pairs, but
//       - a scan along the list is 5 DPP steps (row_shr 1,2,4,8 + row_bcast:15) instead of 6,
//       - the tail of a list wastes on average 16 lanes instead of 32,
//       - GEMM 1 (s[p][g] = <cot[p], chan[g]>) leaves lane (g,h) with exactly its own pixels
//         (row 8q+4h+r of a 32x32 block = pixel row q, column 4h+r): no permlane swaps,
//       - GEMM 2 (dchan[c][g] = sum_p cot[p][c] w[p][g]) takes the lane's own eight weights of two pixel rows as its
//         B fragment directly (the K index is ordered to match; the cot^T fragment is gathered in the same order),
//       - the accumulators of GEMM 2 are 2 x 16 registers instead of 4 x 16, those of GEMM 1 16 instead of 32:
//         the kernel fits 128 VGPRs (four waves per SIMD instead of three).
//   * the pixel-major cotangent image in LDS has pitch 40 bf16 (36 channels + 4 zeros): 11.25 KB per wave, two waves
//     per workgroup -> 14 waves per CU.  K-step 2 of GEMM 1 reads eight columns past the row for h = 1; the B
//     fragment is zero there (channels 40..47 do not exist), so the product is zero whatever finite values LDS returns.
//   * the B fragments of GEMM 1 come pre-split from GeomBuf::ftab (features, bf16 [hi 32 | lo 32], one cache line) and the
//     geometry record (colour + depth): two lines per list entry, no per-view channel-table pass.
//
// Row format and everything downstream (reduce_rows, preprocess_bwd) are unchanged.
#include "common.h"

namespace trase {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int HW_WPB = 2;     // waves (sub-tiles) per workgroup
constexpr int HW_LD = 40;     // LDS row pitch in bf16 (80 B)
constexpr int HW_G = 32;      // list entries per chunk

template <int LD>
struct HwWaveLds {
  // Pixel-major cotangent image, hi and lo bf16 halves, pitch 40: 36 channels + 4 columns that only pad the row to 16 bytes.
  // Those padding columns carry the per-pixel scan state, so that a wave's LDS is exactly 10 KB and SIXTEEN waves fit a CU
  // (a separate 1 KB state array made it 11.25 KB = fourteen; the kernel is bound by latency at that residency):
  //   hi row p, columns 36..39 (8 bytes): T_end (transmittance behind the entries walked so far), U_end (fp32 bits)
  //   lo row p, columns 36..37 (4 bytes): n_contrib;  38..39: zero
  // GEMM 1 would multiply them with the B fragment's zeros -- raw fp32 bits are not finite bf16 -- so its last K-step
  // masks those two dwords of its A fragments; GEMM 2 only produces unused output rows from them.
  // IMG_ONLY instantiation (no feature cotangent: GAUSSIAN-state iterations, train.py:235-243): pitch 8 -- columns 0..3 =
  // channels 32..35 (r g b depth), columns 4..7 = the scan state: 2 KB per wave, the residency is then bound by registers.
  __bf16 hi[WAVE * LD];
  __bf16 lo[WAVE * LD];
};

struct BwdHwArgs {
  const uint2* ranges; const uint32_t* point_list;
  const float2* xy; const float4* conic_o; const float* bg;
  const float* d_img; const float* d_feat; const float* d_depth;
  const float* final_T; const uint32_t* n_contrib;
  const uint32_t* pair_slot;
  const uint32_t* hdr;   // HDR_PACK: pair_slot holds (id << jb) | pair index
  const float4* geo;     // 64-byte geometry records (GeomBuf::geo): {x, y, first row slot, -}, conic_o, ...
  const uint32_t* ftab;  // feature rows as bf16 [hi 32 | lo 32] (GeomBuf::ftab)
  float* rows;         // (capacity, 44)
  uint8_t* row_flags;
  uint32_t* prof;      // 8 counters (TIMING builds) or null
  int W, H, gx8, ntiles;
  int tile0;             // first sub-tile of the strip being rendered (ntiles counts the strip's sub-tiles)
  int order_mode;        // 0: image order; 8 / 16: blocks of 8x8 / 16x16 sub-tiles (common.h blocked_tile)
  int lineage;           // variant bits TRASE_VARIANT_FEATS_BG / TRASE_VARIANT_DEPTH_NORM (0 = public lineage)
  float feat_bg;
  const float* out_depth;  // the forward's depth map (read only for the normalised-depth switch with a depth cotangent)
};

// LDS that only ONE wave produces and consumes: the LDS queue of a wave is in order, so a later ds_read sees an earlier
// ds_write of another lane without any wait.  A wavefront-scope fence keeps the compiler from reordering the accesses
// and -- unlike a workgroup-scope release -- does not drain vmcnt (the gradient-row stores of the chunk, the loads in
// flight for the next one).
__device__ __forceinline__ void wave_lds_sync_hw() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Two independent inclusive scans over each 32-lane half, interleaved (the partner's instruction is one of the two
// wait states a DPP read needs after the VALU write of its source).
#define TRASE_HSCAN2(OP, CTRL) OP " %0, %0, %0 " CTRL "\n\t" OP " %1, %1, %1 " CTRL "\n\ts_nop 0\n\t"
__device__ __forceinline__ void half_scan_mul2(float& a, float& b) {
  asm volatile("s_nop 1\n\t"
               TRASE_HSCAN2("v_mul_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")
               TRASE_HSCAN2("v_mul_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
               TRASE_HSCAN2("v_mul_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf")
               TRASE_HSCAN2("v_mul_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf")
               "v_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "v_mul_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf"
               : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void half_scan_add2_out(float a, float b, float& oa, float& ob) {
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_add_f32_dpp %1, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 0\n\t"
               TRASE_HSCAN2("v_add_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
               TRASE_HSCAN2("v_add_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf")
               TRASE_HSCAN2("v_add_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf")
               "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf"
               : "=&v"(oa), "=&v"(ob) : "v"(a), "v"(b));
}
#undef TRASE_HSCAN2

// cot^T fragment of one channel column for a K-step (two pixel rows): the lane half h supplies the pixels of columns
// 4h..4h+3 -- rows base+0..3 (first pixel row) and base+8..11 (second pixel row) of the pixel-major image.
template <int HW_LD>
__device__ __forceinline__ bf16x8 gather_column_hw(const __bf16* __restrict__ col) {
  const unsigned short* c = reinterpret_cast<const unsigned short*>(col);
  u32x4 r;
  r[0] = (unsigned)c[0 * HW_LD] | ((unsigned)c[1 * HW_LD] << 16);
  r[1] = (unsigned)c[2 * HW_LD] | ((unsigned)c[3 * HW_LD] << 16);
  r[2] = (unsigned)c[8 * HW_LD] | ((unsigned)c[9 * HW_LD] << 16);
  r[3] = (unsigned)c[10 * HW_LD] | ((unsigned)c[11 * HW_LD] << 16);
  return __builtin_bit_cast(bf16x8, r);
}

// COUNT (TRASE_VARIANT_AB_COUNT, `make AB=1` builds only): lane utilisation.  Header words 40..47 receive, summed over the launch:
//   40 list entries walked (sum of chunk lengths)      41 chunks
//   42 pixel-pair steps executed (of 16 per chunk)     43 steps skipped by the 4-pixel last-contributor test
//   44 (pixel, Gaussian) lane slots of executed steps that hold a real list entry
//   45 ... of those, slots that pass the two exponent gates (power <= 0, alpha >= 1/255)
//   46 ... of those, slots that are blended (also in front of the pixel's last contributor)
//   47 walked (sub-tile, Gaussian) pairs that no pixel blended: their gradient row would be all zeros and is not written
// The same fragment through the transposing LDS read of gfx950 (ds_read_b64_tr_b16): within a 16-lane group, lane
// i = 4 jj + cc hands in the address of four consecutive channels (chunk cc) of pixel row jj, and lane c receives element
// c % 4 of the chunks cc = c / 4 of the rows jj = 0..3 -- i.e. channel c of four pixels: one instruction instead of four
// 16-bit reads and two packs.  `p` is this lane's chunk address for the K-step's first four pixel rows; the second four
// are eight rows further on.
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int HW_LD>
__device__ __forceinline__ bf16x8 gather_column_tr(const __bf16* __restrict__ p) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 8 * HW_LD));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}

// FEAT_ONLY: only dL/dsh_objs (TRASE_VARIANT_FEATURES_ONLY_BWD).  IMG_ONLY: no feature cotangent at all -- the image (and
// depth) cotangent only, what a GAUSSIAN-state iteration back-propagates (train.py:235-243, :299): GEMM 1 shrinks to its
// last K-step (channels 32..35), GEMM 2 to channel block 1, a gradient row to its 12 geometry / colour columns.
template <bool FEAT_ONLY, bool TIMING, bool COUNT = false, bool IMG_ONLY = false>
#ifndef HW_OCC
#define HW_OCC 4
#endif
#ifndef HW_OCC_IMG
#define HW_OCC_IMG 4
#endif
__global__ __launch_bounds__(HW_WPB* WAVE) __attribute__((amdgpu_waves_per_eu(IMG_ONLY ? HW_OCC_IMG : HW_OCC, IMG_ONLY ? HW_OCC_IMG : HW_OCC)))
void render_bwd_hw_kernel(BwdHwArgs a) {
  static_assert(!(FEAT_ONLY && IMG_ONLY), "scopes exclude each other");
  constexpr int F = IMG_ONLY ? 0 : 32;                   // feature columns of a gradient row
  constexpr int HW_LD = IMG_ONLY ? 8 : trase::HW_LD;     // LDS row pitch (shadows the namespace constant inside the kernel)
  constexpr int C0 = IMG_ONLY ? 0 : 32;                  // LDS column of channel 32 (r); the scan state sits at C0 + 4
  constexpr int ST = C0 + 4;
  __shared__ __attribute__((aligned(16))) HwWaveLds<HW_LD> s_w[HW_WPB];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int g = lane & 31, h = lane >> 5;
  const int local = xcd_block(blockIdx.x, gridDim.x) * HW_WPB + wave;
  if (local >= a.ntiles) return;
  int tx, ty;
  if (a.order_mode == 0) { const int t0 = a.tile0 + local; tx = t0 % a.gx8; ty = t0 / a.gx8; }
  else {
    if (a.order_mode == 8) blocked_tile<8>(local, a.gx8, a.ntiles / a.gx8, tx, ty); else blocked_tile<16>(local, a.gx8, a.ntiles / a.gx8, tx, ty);
    ty += a.tile0 / a.gx8;
  }
  const int tile = ty * a.gx8 + tx;
  const uint2 range = a.ranges[tile];
  HwWaveLds<HW_LD>& L = s_w[wave];
  uint64_t t_mark = 0, t_acc[5] = {0, 0, 0, 0, 0};
  auto tick = [&](int k) { if constexpr (TIMING) { const uint64_t t = __builtin_readcyclecounter(); t_acc[k] += t - t_mark; t_mark = t; } };
  if constexpr (TIMING) t_mark = __builtin_readcyclecounter();
  uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // ---- stage this sub-tile's per-pixel data (lane = pixel here) ---------------------------------
  uint32_t last;
  {
    const int px = tx * SUB + (lane & 7), py = ty * SUB + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const size_t hw = (size_t)a.H * a.W;
    const size_t pix = (size_t)py * a.W + px;
    float v[HW_LD];
#pragma unroll
    for (int c = 0; c < HW_LD; ++c) v[c] = 0.f;
    float Tf = 0.f;
    last = 0;
    if (inside) {
      Tf = a.final_T[pix];
      last = a.n_contrib[pix];
      if constexpr (!IMG_ONLY) {
        if (a.d_feat) {
#pragma unroll
          for (int c = 0; c < 32; ++c) v[c] = a.d_feat[(size_t)c * hw + pix];
        }
      }
      if (a.d_img) { v[C0] = a.d_img[pix]; v[C0 + 1] = a.d_img[hw + pix]; v[C0 + 2] = a.d_img[2 * hw + pix]; }
      if (a.d_depth) v[C0 + 3] = a.d_depth[pix];
    }
    // Lineage switches: an output of the form X + T_final * b contributes b * cotangent to the pixel's "background" sum.
    float bextra = 0.f;
    if constexpr (!IMG_ONLY) {
      if (a.lineage & TRASE_VARIANT_FEATS_BG) {
        float sf = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) sf += v[c];
        bextra = a.feat_bg * sf;
      }
    }
    if ((a.lineage & TRASE_VARIANT_DEPTH_NORM) && a.d_depth && inside) {
      // depth_out = D / A, A = 1 - T_final:  dL/dD = g / A,  d depth_out / d T_final = D / A^2 = depth_out / A
      const float A = 1.0f - Tf, gd = v[C0 + 3];
      const float ga = A > 1e-10f ? gd / A : 0.0f;
      v[C0 + 3] = ga;
      bextra = fmaf(ga, a.out_depth[pix], bextra);
    }
    __bf16* rh = L.hi + lane * HW_LD;
    __bf16* rl = L.lo + lane * HW_LD;
    const float bdot = a.bg[0] * v[C0] + a.bg[1] * v[C0 + 1] + a.bg[2] * v[C0 + 2] + bextra;
#pragma unroll
    for (int c8 = 0; c8 < HW_LD / 8; ++c8) {
      bf16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) { hi[e] = (__bf16)v[8 * c8 + e]; lo[e] = (__bf16)(v[8 * c8 + e] - (float)hi[e]); }
      u32x4 uh = __builtin_bit_cast(u32x4, hi), ul = __builtin_bit_cast(u32x4, lo);
      if (c8 == HW_LD / 8 - 1) {                          // columns 36..39: the pixel's scan state instead of zeros
        uh[2] = __float_as_uint(Tf); uh[3] = __float_as_uint(Tf * bdot);
        ul[2] = last; ul[3] = 0u;
      }
      *reinterpret_cast<u32x4*>(rh + 8 * c8) = uh;
      *reinterpret_cast<u32x4*>(rl + 8 * c8) = ul;
    }
  }
  uint32_t wave_last = last;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, o));
  wave_last = __builtin_amdgcn_readfirstlane(wave_last);
  // a step of the pixel loop visits pixels p, p+1 (half 0) and p+4, p+5 (half 1): lane p of `last4` holds the largest
  // last-contributor index of that group, so the per-step skip test is one v_readlane + one scalar compare
  uint32_t last4 = max(last, (uint32_t)__shfl_xor((int)last, 1));
  last4 = max(last4, (uint32_t)__shfl_xor((int)last4, 4));
  wave_lds_sync_hw();
  const float ddx = 0.5f * (float)a.W, ddy = 0.5f * (float)a.H;
  const float bx = (float)(tx * SUB), by = (float)(ty * SUB);
  const __bf16* const ahi = L.hi;
  const __bf16* const alo = L.lo;
  // pixel columns of this lane half: j = 4h + r
  float jv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) jv[r] = (float)(4 * h + r);
  // channel column of this lane for the cot^T gathers: block nb covers channels nb*32 + g; columns >= 40 do not exist
  // (they are zero): read column 39, which is zero, instead
  const int col0 = g, col1 = min(32 + g, HW_LD - 1);
  const unsigned hmask = h ? 0xffffffffu : 0u;           // GEMM 1, last K-step: half 0 holds the state dwords of columns 36..39
  // transposing-read addressing of the same fragments: pixel row (lane & 15) >> 2 of the K-step's four, channel chunk
  // 16 * ((lane >> 4) & 1) + 4 * (lane & 3) of the block; block 1 only has the chunks 32..35 and 36..39 (zeros)
  const int trrow = ((lane & 15) >> 2) * HW_LD;
  const int tr0 = trrow + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int tr1 = trrow + min(C0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3), HW_LD - 4);
  tick(0);
  // The list entries of a chunk are requested one chunk ahead (two registers): one level less in the dependent chain
  // entry -> geometry / channel rows at a chunk start.
  // HDR_PACK (jb > 0): pair_slot holds (id << jb) | pair index; the row slot = the Gaussian's first slot (a word of
  // its geometry record, fetched anyway) + that index.  point_list is not read (nor written by the forward) then.
  const uint32_t jb = a.hdr[HDR_PACK];
  uint32_t id_n = 0, slot_n = 0xffffffffu;
  {
    const uint32_t n0 = min(wave_last, (uint32_t)HW_G);
    const bool v0 = (uint32_t)g < n0;
    const uint32_t p0 = v0 ? (wave_last - 1 - g) : 0;
    if (wave_last > 0) { slot_n = a.pair_slot[range.x + p0]; if (!jb) id_n = a.point_list[range.x + p0]; }
  }
  // A chunk's gradient rows are stored at the START of the next chunk, right behind that chunk's per-entry loads: loads and
  // stores retire through one in-order counter, and loads issued behind eight row stores wait for the stores'
  // acknowledgements before their data counts as arrived.
  struct PendingRow { uint32_t slot; bool write; float4 d[4]; float4 m[3]; };
  PendingRow pend;
  pend.slot = 0xffffffffu; pend.write = false;
  auto store_pending = [&]() {
    if (pend.write) {
      float* row = a.rows + (size_t)pend.slot * bwd_row_stride(F);
      if constexpr (!IMG_ONLY) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(row + 8 * q + 4 * h) = pend.d[q];
      }
      if (h == 0) {
        *reinterpret_cast<float4*>(row + F) = pend.m[0];
        *reinterpret_cast<float4*>(row + F + 4) = pend.m[1];
        *reinterpret_cast<float4*>(row + F + 8) = pend.m[2];
        a.row_flags[pend.slot] = 1;
      }
    }
  };
  // ---- chunks of 32 list entries, back to front ----------------------------------------------------
  for (uint32_t c1 = wave_last; c1 > 0; c1 = (c1 > HW_G) ? c1 - HW_G : 0) {
    const uint32_t c0 = (c1 > HW_G) ? c1 - HW_G : 0;
    const uint32_t n = c1 - c0;
    const bool lane_valid = (uint32_t)g < n;
    if constexpr (COUNT) { cnt[0] += n; cnt[1] += 1; }
    const uint32_t pos = lane_valid ? (c1 - 1 - g) : 0;        // g = 0: farthest entry of the chunk
    const uint32_t id = jb ? (slot_n >> jb) : id_n;
    const float4 gq = a.geo[4 * (size_t)id];
    const float4 co = a.geo[4 * (size_t)id + 1];
    const float2 gxy = make_float2(gq.x, gq.y);
    const uint32_t slot = !lane_valid ? 0xffffffffu : (jb ? __float_as_uint(gq.z) + (slot_n & ((1u << jb) - 1u)) : slot_n);
    if (c0 > 0) {                                              // next (nearer) chunk's entries
      const uint32_t c0n = (c0 > HW_G) ? c0 - HW_G : 0;
      const bool vn = (uint32_t)g < c0 - c0n;
      const uint32_t pn = vn ? (c0 - 1 - g) : 0;
      slot_n = a.pair_slot[range.x + pn];
      if (!jb) id_n = a.point_list[range.x + pn];
    }
    const PairPoly k = pair_poly(gxy, co, bx, by);
    const uint32_t pos_cmp = lane_valid ? pos : 0xffffffffu;
    const uint32_t* const frow = a.ftab + (size_t)id * 32 + 4 * h;   // this lane's B fragments of GEMM 1: bf16 [hi 32 | lo 32]
    const float4 cs = a.geo[4 * (size_t)id + 3];                     // ... and colour + depth, split the same way
    bf16x8 fbh[2], fbl[2];                                           // (requested here, ahead of the pending row stores)
    if constexpr (!FEAT_ONLY && !IMG_ONLY) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        fbh[ks] = *reinterpret_cast<const bf16x8*>(frow + ks * 8);
        fbl[ks] = *reinterpret_cast<const bf16x8*>(frow + ks * 8 + 16);
      }
    }
    store_pending();                                                 // the previous chunk's rows
    tick(1);
    // Accumulators are never zero-filled: the first product of each takes a literal zero C operand (an inline constant
    // of the MFMA encoding), which saves 64 v_mov per chunk.
    const f32x16 ZERO16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool blended = false;                                // some pixel of this lane's columns blended this Gaussian
    f32x16 D[2];                                         // D[channel block]: rows = channels, columns = Gaussians
    if constexpr (FEAT_ONLY) D[1] = ZERO16;              // never accumulated there; its four sums are written as zeros
    // moment sums with the LOCAL column index jl = 0..3 of this lane half (global column j = 4h + jl): the row sums are
    // then plain constants-weighted sums; the shift to the global index happens once per chunk
    float S0 = 0.f, Sj = 0.f, Si = 0.f, Sjj = 0.f, Sij = 0.f, Sii = 0.f;
    unsigned wh[4], wl[4];                               // 8 weights of the current K-step, packed bf16 pairs
    // ---- GEMM 1: S[mb][p][g] for both 32-pixel halves, the channel fragments loaded once -----------------------
    // a = cot fragment (row = pixel mb*32 + (lane & 31), 8 channels), b = channel fragment (column = Gaussian g).
    // Lane (g,h) then holds Gaussian g, pixels mb*32 + 8q + 4h + r in register 4q + r: pixel row 4mb + q, column 4h + r.
    f32x16 Sm[2];
    if constexpr (!FEAT_ONLY) {
#pragma unroll
      for (int ks = IMG_ONLY ? 2 : 0; ks < 3; ++ks) {
        bf16x8 bh, bl;
        if (ks < 2) {
          bh = fbh[ks]; bl = fbl[ks];
        } else {                                         // channels 32..35 = r g b depth in the h = 0 half, zeros elsewhere
          const u32x4 ch4 = {h ? 0u : __float_as_uint(cs.x), h ? 0u : __float_as_uint(cs.y), 0u, 0u};
          const u32x4 cl4 = {h ? 0u : __float_as_uint(cs.z), h ? 0u : __float_as_uint(cs.w), 0u, 0u};
          bh = __builtin_bit_cast(bf16x8, ch4); bl = __builtin_bit_cast(bf16x8, cl4);
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          bf16x8 ph, pl;
          if (ks < 2) {
            ph = *reinterpret_cast<const bf16x8*>(ahi + (mb * 32 + g) * HW_LD + ks * 16 + 8 * h);
            pl = *reinterpret_cast<const bf16x8*>(alo + (mb * 32 + g) * HW_LD + ks * 16 + 8 * h);
          } else {
            // channels 32..39 (h = 0: the state dwords are masked) / 40..47 (h = 1: the next pixel row's first eight columns,
            // multiplied by the B fragment's zeros; the last row of the image reads its own first columns instead)
            // (IMG_ONLY, pitch 8: every lane reads its pixel's own 16 bytes; the half h = 1 -- channels 40..47, which do not
            // exist -- and the state dwords are zeroed: the next row's first columns would be ITS state words there)
            const int off = IMG_ONLY ? (mb * 32 + g) * HW_LD : ((mb == 1 && lane == 63) ? 63 * HW_LD : (mb * 32 + g) * HW_LD + C0 + 8 * h);
            u32x4 rh4 = *reinterpret_cast<const u32x4*>(ahi + off), rl4 = *reinterpret_cast<const u32x4*>(alo + off);
            if constexpr (IMG_ONLY) {
              rh4[0] &= ~hmask; rh4[1] &= ~hmask; rl4[0] &= ~hmask; rl4[1] &= ~hmask;
              rh4[2] = 0u; rh4[3] = 0u; rl4[2] = 0u; rl4[3] = 0u;
            } else {
              rh4[2] &= hmask; rh4[3] &= hmask; rl4[2] &= hmask; rl4[3] &= hmask;
            }
            ph = __builtin_bit_cast(bf16x8, rh4); pl = __builtin_bit_cast(bf16x8, rl4);
          }
          Sm[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, bh, ks == (IMG_ONLY ? 2 : 0) ? ZERO16 : Sm[mb], 0, 0, 0);
          Sm[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, bl, Sm[mb], 0, 0, 0);
          Sm[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pl, bh, Sm[mb], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const f32x16& S = Sm[mb];
      tick(2);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = mb * 4 + q;
        const float fi = (float)i, fii = (float)(i * i);
        const float base = poly_row_base(k, fi, fii);
        const float slope = poly_row_slope(k, fi);
        float R0 = 0.f, R1 = 0.f, R2 = 0.f;              // row sums of q, q*jl, q*jl^2 over this lane's four columns (jl = 0..3)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {                   // two pixels per half and step: their scans are interleaved
          // last-contributor indices of the four pixels of this step (lane = pixel register, wave-uniform reads)
          const int p0 = i * SUB + r;                      // half 0: p0, p0+1; half 1: p0+4, p0+5
          unsigned hb = 0, lb = 0;                         // the step's two weights as packed bf16 pairs: w = hi + lo
          if constexpr (COUNT) { if ((uint32_t)__builtin_amdgcn_readlane((int)last4, p0) > c0) cnt[2] += 1; else cnt[3] += 1; }
          if ((uint32_t)__builtin_amdgcn_readlane((int)last4, p0) > c0) {
            const int pl = p0 + 4 * h;                     // this lane's first pixel of the step
            const float2 pa = *reinterpret_cast<const float2*>(L.hi + pl * HW_LD + ST);          // T_end, U_end
            const float2 pb = *reinterpret_cast<const float2*>(L.hi + (pl + 1) * HW_LD + ST);
            const uint32_t lasta = *reinterpret_cast<const uint32_t*>(L.lo + pl * HW_LD + ST);
            const uint32_t lastb = *reinterpret_cast<const uint32_t*>(L.lo + (pl + 1) * HW_LD + ST);
            const float ea = poly_eval(k, base, slope, jv[r]);
            const float eb = poly_eval(k, base, slope, jv[r + 1]);
            const bool oka = (ea <= k.thr) && (ea >= LOG2_ALPHA_MIN) && (pos_cmp < lasta);
            const bool okb = (eb <= k.thr) && (eb >= LOG2_ALPHA_MIN) && (pos_cmp < lastb);
            blended = blended || oka || okb;
            if constexpr (COUNT) {
              cnt[4] += 2 * (uint32_t)__builtin_popcountll(__ballot(lane_valid));
              cnt[5] += (uint32_t)__builtin_popcountll(__ballot(lane_valid && (ea <= k.thr) && (ea >= LOG2_ALPHA_MIN))) +
                        (uint32_t)__builtin_popcountll(__ballot(lane_valid && (eb <= k.thr) && (eb >= LOG2_ALPHA_MIN)));
              cnt[6] += (uint32_t)__builtin_popcountll(__ballot(oka)) + (uint32_t)__builtin_popcountll(__ballot(okb));
            }
            const float ra = __builtin_amdgcn_exp2f(oka ? ea : -INFINITY);   // opacity * exp(power); closed gate = 0
            const float rb = __builtin_amdgcn_exp2f(okb ? eb : -INFINITY);
            const float ala = fminf(ALPHA_MAX, ra), alb = fminf(ALPHA_MAX, rb);
            const float roma = __builtin_amdgcn_rcpf(1.0f - ala), romb = __builtin_amdgcn_rcpf(1.0f - alb);
            float Pa = roma, Pb = romb;
            half_scan_mul2(Pa, Pb);
            const float Ta = pa.x * Pa, Tb = pb.x * Pb;   // transmittance in front of this Gaussian
            const float wa = ala * Ta, wb = alb * Tb;
            if constexpr (FEAT_ONLY) {
              if (g == HW_G - 1) {
                *reinterpret_cast<float*>(L.hi + pl * HW_LD + ST) = Ta;
                *reinterpret_cast<float*>(L.hi + (pl + 1) * HW_LD + ST) = Tb;
              }
            } else {
              const float sa = S[4 * q + r], sb = S[4 * q + r + 1];
              const float wsa = wa * sa, wsb = wb * sb;
              float ia, ib;
              half_scan_add2_out(wsa, wsb, ia, ib);
              const float Ua = pa.y + (ia - wsa), Ub = pb.y + (ib - wsb);
              const float dLa = Ta * sa - Ua * roma, dLb = Tb * sb - Ub * romb;
              if (g == HW_G - 1) {                         // carries for the next (nearer) chunk
                *reinterpret_cast<float2*>(L.hi + pl * HW_LD + ST) = make_float2(Ta, pa.y + ia);
                *reinterpret_cast<float2*>(L.hi + (pl + 1) * HW_LD + ST) = make_float2(Tb, pb.y + ib);
              }
              const float qa = ra * dLa, qb = rb * dLb;    // == opacity * G * dL/dalpha (straight-through clamp)
              if (r == 0) {                                // local columns jl = 0, 1: weights (1, 0, 0) and (1, 1, 1)
                R0 = qa + qb; R1 = qb; R2 = qb;
              } else {                                     // jl = 2, 3: weights (1, 2, 4) and (1, 3, 9)
                R0 += qa + qb;
                R2 = fmaf(9.0f, qb, fmaf(4.0f, qa, R2));
                R1 = fmaf(3.0f, qb, fmaf(2.0f, qa, R1));
              }
            }
            {                                              // split the two weights, packed
#pragma clang fp contract(off)                             // the residual of the ROUNDED weight in every instantiation (not fma(alpha, T, -hi))
              asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hb) : "v"(wa), "v"(wb));
              const float ra2 = wa - __uint_as_float(hb << 16), rb2 = wb - __uint_as_float(hb & 0xffff0000u);
              asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lb) : "v"(ra2), "v"(rb2));
            }
          }
          wh[2 * (q & 1) + (r >> 1)] = hb;                 // K index = 4 * (pixel row parity) + r
          wl[2 * (q & 1) + (r >> 1)] = lb;
        }
        if constexpr (!FEAT_ONLY) {
          S0 += R0; Sj += R1; Sjj += R2;
          Si = fmaf(fi, R0, Si); Sii = fmaf(fii, R0, Sii); Sij = fmaf(fi, R1, Sij);
        }
        if (q & 1) {
          tick(3);
          // ---- GEMM 2, K-step t = i/2: pixel rows 2t, 2t+1 ---------------------------------------------
          const bf16x8 Bh = __builtin_bit_cast(bf16x8, (u32x4){wh[0], wh[1], wh[2], wh[3]});
          const bf16x8 Bl = __builtin_bit_cast(bf16x8, (u32x4){wl[0], wl[1], wl[2], wl[3]});
          const int rowoff = ((i >> 1) * 16 + 4 * h) * HW_LD;
          constexpr int NBLK = FEAT_ONLY ? 1 : 2;          // channel block 1 = r g b depth
#pragma unroll
          for (int nb = IMG_ONLY ? 1 : 0; nb < NBLK; ++nb) {
#ifdef TRASE_BWD_NO_TR
            const int col = nb == 0 ? col0 : col1;
            const bf16x8 Ah = gather_column_hw<HW_LD>(ahi + rowoff + col), Al = gather_column_hw<HW_LD>(alo + rowoff + col);
#else
            const int tro = nb == 0 ? tr0 : tr1;
            const bf16x8 Ah = gather_column_tr<HW_LD>(ahi + rowoff + tro), Al = gather_column_tr<HW_LD>(alo + rowoff + tro);
#endif
            D[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, i == 1 ? ZERO16 : D[nb], 0, 0, 0);
            D[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, D[nb], 0, 0, 0);
            D[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, D[nb], 0, 0, 0);
          }
          tick(4);
        }
      }
    }
    // ---- one row per pair: [32 feature sums | nx ny ca cb | cc op r g | b d 0 0] -------------------------
    // moment sums of the two lane halves (columns 0..3 and 4..7) -> totals in both halves
    auto both = [](float x) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
      return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    };
    // a pair that no pixel of the sub-tile blended (every gate closed: the 8x8 culling is conservative, and entries can lie
    // behind all last contributors of their own pixels) has an all-zero row: neither row nor flag is written, reduce_rows
    // never reads it.  The two lane halves hold the same Gaussians and different pixel columns.
    bool write_row;
    {
      const unsigned long long bm = __ballot(blended);
      write_row = slot != 0xffffffffu && (((bm | (bm >> 32)) >> g) & 1ull);
      if constexpr (COUNT) cnt[7] += (uint32_t)__builtin_popcountll(__ballot(slot != 0xffffffffu && h == 0 && !write_row));
    }
    pend.slot = slot; pend.write = write_row;
    // D[0]: lane (g,h), register 4q + r = channel 8q + 4h + r
    if constexpr (!IMG_ONLY) {
#pragma unroll
      for (int q = 0; q < 4; ++q) pend.d[q] = make_float4(D[0][4 * q], D[0][4 * q + 1], D[0][4 * q + 2], D[0][4 * q + 3]);
    }
    if constexpr (!FEAT_ONLY) {
      // local -> global column index (j = 4h + jl):  sum q j = Sj + 4h S0,  sum q j^2 = Sjj + 8h Sj + 16 h^2 S0,
      // sum q i j = Sij + 4h Si   (h = 0: unchanged)
      const float h4 = (float)(4 * h);
      Sjj = fmaf(h4, fmaf(h4, S0, 2.0f * Sj), Sjj);
      Sj = fmaf(h4, S0, Sj);
      Sij = fmaf(h4, Si, Sij);
      S0 = both(S0); Sj = both(Sj); Si = both(Si); Sjj = both(Sjj); Sij = both(Sij); Sii = both(Sii);
    }
    {
      // moments about the sub-tile origin -> sums over dx = rx - j, dy = ry - i
      const float rx = gxy.x - bx, ry = gxy.y - by;
      const float Qx = rx * S0 - Sj, Qy = ry * S0 - Si;
      const float Qxx = rx * (rx * S0 - 2.0f * Sj) + Sjj;
      const float Qyy = ry * (ry * S0 - 2.0f * Si) + Sii;
      const float Qxy = rx * (ry * S0 - Si) - ry * Sj + Sij;
      const float a_nx = -(co.x * Qx + co.y * Qy);
      const float a_ny = -(co.z * Qy + co.y * Qx);
      const float a_ca = -0.5f * Qxx, a_cb = -Qxy, a_cc = -0.5f * Qyy;
      const float a_op = (co.w > 0.0f) ? S0 / co.w : 0.0f;
      pend.m[0] = make_float4(a_nx * ddx, a_ny * ddy, a_ca, a_cb);
      // channels 32..35 (r g b depth sums): registers 0..3 of channel block 1 in the h = 0 half
      pend.m[1] = make_float4(a_cc, a_op, D[1][0], D[1][1]);
      pend.m[2] = make_float4(D[1][2], D[1][3], 0.f, 0.f);
    }
    wave_lds_sync_hw();                                   // carries written by lanes 31 / 63 are read by the next chunk
    tick(0);
  }
  store_pending();                                        // the last chunk's rows
  if constexpr (COUNT) {
    if (lane == 0 && a.prof) {
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) atomicAdd(a.prof + 8 + k2, cnt[k2]);
    }
  }
  if constexpr (TIMING) {
    if (lane == 0 && a.prof) {
#pragma unroll
      for (int k2 = 0; k2 < 5; ++k2) atomicAdd(a.prof + k2, (uint32_t)(t_acc[k2] >> 6));
      atomicAdd(a.prof + 5, 1u);
    }
  }
}

int launch_render_bwd_hw(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, const GeomBuf& g,
                         const BinBuf& bb, const ImgBuf& im, const TraseRastGrads& gr, float* rows, uint8_t* row_flags,
                         size_t flag_bytes, const float* out_depth) {
  BwdHwArgs a;
  a.lineage = c.variant & (TRASE_VARIANT_FEATS_BG | TRASE_VARIANT_DEPTH_NORM); a.feat_bg = s.feat_bg; a.out_depth = out_depth;
  a.order_mode = 16;
#ifdef TRASE_AB
  a.order_mode = (c.variant & TRASE_VARIANT_AB_ORDER_IMAGE) ? 0 : ((c.variant & TRASE_VARIANT_AB_ORDER_8) ? 8 : 16);
#endif
  if ((a.lineage & TRASE_VARIANT_DEPTH_NORM) && gr.dL_ddepth && !out_depth) {
    set_error("render_bwd: the normalised-depth switch with a depth cotangent needs the forward's depth map (outputs.depth)");
    return TRASE_ERR_INVALID;
  }
  a.ranges = bb.ranges; a.point_list = bb.point_list; a.xy = g.xy; a.conic_o = g.conic_o; a.bg = s.bg;
  a.d_img = gr.dL_dimage; a.d_feat = gr.dL_dfeats; a.d_depth = gr.dL_ddepth;
  a.final_T = im.final_T; a.n_contrib = im.n_contrib; a.pair_slot = bb.pair_slot; a.hdr = g.hdr; a.geo = g.geo;
  a.ftab = g.ftab; a.rows = rows; a.row_flags = row_flags;
  a.prof = g.hdr + 32;                                   // header words 32..39: phase cycle counters of the TIMING build
  a.W = s.image_width; a.H = s.image_height;
  a.gx8 = (a.W + SUB - 1) / SUB;
  { int lo, hi; strip_subtile_rows(s, lo, hi); a.tile0 = lo * a.gx8; a.ntiles = (hi - lo) * a.gx8; }
  launch_zero_bytes(row_flags, flag_bytes, c.stream);   // (a kernel, not a memset node: common.h)
  if (a.ntiles <= 0) return TRASE_OK;                    // an empty strip: no rows (the flags are cleared)
  {
    ProfScope ps("render_bwd", c.stream);
    const dim3 grid((a.ntiles + HW_WPB - 1) / HW_WPB), block(HW_WPB * WAVE);
    if (in.F == 0) hipLaunchKernelGGL((render_bwd_hw_kernel<false, false, false, true>), grid, block, 0, c.stream, a);   // image-only scope
    else if (c.variant & TRASE_VARIANT_FEATURES_ONLY_BWD) hipLaunchKernelGGL((render_bwd_hw_kernel<true, false>), grid, block, 0, c.stream, a);
#ifdef TRASE_AB                                          // diagnostic instantiations: `make AB=1`
    else if (c.variant & TRASE_VARIANT_AB_TIMING) hipLaunchKernelGGL((render_bwd_hw_kernel<false, true>), grid, block, 0, c.stream, a);        // phase timing
    else if (c.variant & TRASE_VARIANT_AB_COUNT) hipLaunchKernelGGL((render_bwd_hw_kernel<false, false, true>), grid, block, 0, c.stream, a);  // lane-utilisation counters
#endif
    else hipLaunchKernelGGL((render_bwd_hw_kernel<false, false>), grid, block, 0, c.stream, a);
  }
  TRASE_POST_LAUNCH("render_bwd", c.stream, c.debug);
  return TRASE_OK;
}

}  // namespace trase
