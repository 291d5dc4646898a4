// render_bwd_mf.hip -- backward of the compositing stage with the channel contractions on the matrix cores
// (default for F = 32; render_bwd_gs.hip is the pure-VALU formulation it grew out of and the fallback).
//
// Same decomposition as render_bwd_gs.hip: one wave per 8x8 sub-tile, chunks of 64 list entries walked back to
// front, lane = Gaussian, the 64 pixels visited one after the other with DPP scans across the lanes.  In that
// kernel half of the VALU instructions per (pixel, chunk) are two channel contractions:
//     s[g][p]     = sum_c  chan[g][c] * cot[p][c]        (36 channels: 32 features, r, g, b, depth)
//     dchan[g][c] = sum_p  w[g][p]   * cot[p][c]         (w = alpha * T)
// Both are small GEMMs (64 x 64 x 36 per chunk).  Here they run as v_mfma_f32_32x32x16_bf16 with every fp32
// operand split into two bf16 values (x = hi + lo, |x - hi - lo| <= 2^-17 |x|) and three products per term
// (hi*hi + hi*lo + lo*hi, fp32 accumulation): relative error ~1e-5, deterministic, and the VALU keeps only the
// part that really is sequential (exponent, gates, the two scans, the moment sums).
//   * cot[p][c]: split once per sub-tile into wave-private LDS, pixel-major, row pitch 112 B.
//   * chan[g][c]: split once per view into a [P][hi 48 | lo 48] bf16 table (split_channels_kernel); a B fragment
//     (8 consecutive channels of one Gaussian) is one 16-byte global load.
//   * s: GEMM 1 leaves D[p][g] with lane = g; one v_permlane32_swap per register pair brings both pixel halves of a
//     Gaussian to the lane that owns it (lane l owns Gaussian l of the chunk).
//   * w: after every 16 pixels the lane's 16 weights are split, packed and swapped into the two A fragments of
//     GEMM 2; the matching cot^T fragments (8 consecutive pixels of one channel) are gathered from the pixel-major
//     LDS image with 16-bit reads.  The product is taken as cot^T x w^T, so the result has lane = Gaussian and a lane
//     stores 16-byte pieces of its own gradient row.
#include "common.h"

namespace trase {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int MF_WPB = 4;     // waves (sub-tiles) per workgroup
constexpr int MF_CH = 48;     // 32 features | r g b depth | 12 zero columns
constexpr int MF_LD = 48;     // LDS row pitch in bf16 (96 B): 4 waves x (2 x 6 KB + 1 KB) = 52 KB, three workgroups per CU

__device__ __forceinline__ void wave_lds_sync3() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ unsigned bf16_hi_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }

// x -> (hi, lo) bf16 bit patterns
__device__ __forceinline__ void split2(float x, unsigned& hi, unsigned& lo) {
  const __bf16 h = (__bf16)x;
  hi = (unsigned)__builtin_bit_cast(unsigned short, h);
  lo = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)(x - (float)h));
}

// ---- per view: channel table [P][hi 48 | lo 48] ---------------------------------------------------------------
__global__ __launch_bounds__(256) void split_channels_kernel(const float* __restrict__ feats, const float4* __restrict__ rgbd,
                                                             const uint32_t* __restrict__ tiles,
                                                             int P, __bf16* __restrict__ out, uint4* __restrict__ flags16,
                                                             size_t nflags16) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // one thread per (Gaussian, group of 8 channels)
  // the row flags of the backward are cleared here (saves a separate fill launch)
  for (size_t i = (size_t)idx; i < nflags16; i += (size_t)gridDim.x * blockDim.x) flags16[i] = make_uint4(0u, 0u, 0u, 0u);
  if (idx >= P * 6) return;
  const int g = idx / 6, grp = idx % 6;
  if (tiles[g] == 0) return;                                 // no list holds this Gaussian (culled, or outside the strip being rendered)
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (grp < 4) {
    const float4 a = *reinterpret_cast<const float4*>(feats + (size_t)g * 32 + grp * 8);
    const float4 b = *reinterpret_cast<const float4*>(feats + (size_t)g * 32 + grp * 8 + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else if (grp == 4) {
    const float4 c = rgbd[g];
    v[0] = c.x; v[1] = c.y; v[2] = c.z; v[3] = c.w;
  }
  bf16x8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) { hi[e] = (__bf16)v[e]; lo[e] = (__bf16)(v[e] - (float)hi[e]); }
  *reinterpret_cast<bf16x8*>(out + (size_t)g * (2 * MF_CH) + grp * 8) = hi;
  *reinterpret_cast<bf16x8*>(out + (size_t)g * (2 * MF_CH) + MF_CH + grp * 8) = lo;
}

struct BwdMfArgs {
  const uint2* ranges; const uint32_t* point_list;
  const float2* xy; const float4* conic_o; const float* bg;
  const float* d_img; const float* d_feat; const float* d_depth;
  const float* final_T; const uint32_t* n_contrib;
  const uint32_t* pair_slot;
  const uint32_t* hdr; const float4* geo;   // HDR_PACK: pair_slot holds (id << jb) | pair index; first row slot in geo
  const __bf16* chan;  // [P][96]
  float* rows;         // (capacity, 44)
  uint8_t* row_flags;
  int W, H, gx8, ntiles;
  int tile0;             // first sub-tile of the strip being rendered (ntiles counts the strip's sub-tiles)
};

// 8 consecutive pixels (rows p0 .. p0+7 of the pixel-major image) of one channel column -> one MFMA fragment.
// (Plain 16-bit reads + one v_lshl_or per pair: with SRAM-ECC the d16 / d16_hi loads do not preserve the other
// half of the destination register, so they cannot assemble pairs in place.)
__device__ __forceinline__ bf16x8 gather_column(const __bf16* __restrict__ col) {
  const unsigned short* c = reinterpret_cast<const unsigned short*>(col);
  u32x4 r;
#pragma unroll
  for (int q = 0; q < 4; ++q) r[q] = (unsigned)c[(2 * q) * MF_LD] | ((unsigned)c[(2 * q + 1) * MF_LD] << 16);
  return __builtin_bit_cast(bf16x8, r);
}

// FEAT_ONLY: only dL/d(feature channels of the Gaussians) is produced (FEATURE state once densification has ended: nothing
// else requires a gradient then, train.py:244-299 with scene/gaussian_model.py:303-315).  GEMM 1, the dL/dalpha scan and the
// moments disappear; the weights alpha * T and GEMM 2 remain.  The rows keep their format with a zero tail.
template <bool FEAT_ONLY>
__global__ __launch_bounds__(MF_WPB* WAVE) __attribute__((amdgpu_waves_per_eu(3, 3)))
void render_bwd_mf_kernel(BwdMfArgs a) {
  constexpr int F = 32, ROW = F + 12;
  __shared__ __attribute__((aligned(16))) __bf16 s_hi[MF_WPB][WAVE * MF_LD];   // cotangents, pixel-major, high parts
  __shared__ __attribute__((aligned(16))) __bf16 s_lo[MF_WPB][WAVE * MF_LD];   // low parts
  __shared__ __attribute__((aligned(16))) float4 s_pix[MF_WPB][WAVE];          // T_end, U_end, last (bits), -
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  const int local = xcd_block(blockIdx.x, gridDim.x) * MF_WPB + wave;
  if (local >= a.ntiles) return;
  const int tile = a.tile0 + local;
  const int tx = tile % a.gx8, ty = tile / a.gx8;
  const uint2 range = a.ranges[tile];
  // ---- stage this sub-tile's per-pixel data (lane = pixel here) ---------------------------------
  uint32_t last;
  {
    const int px = tx * SUB + (lane & 7), py = ty * SUB + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const size_t hw = (size_t)a.H * a.W;
    const size_t pix = (size_t)py * a.W + px;
    float v[MF_CH];
#pragma unroll
    for (int c = 0; c < MF_CH; ++c) v[c] = 0.f;
    float Tf = 0.f;
    last = 0;
    if (inside) {
      Tf = a.final_T[pix];
      last = a.n_contrib[pix];
      if (a.d_feat) {
#pragma unroll
        for (int c = 0; c < F; ++c) v[c] = a.d_feat[(size_t)c * hw + pix];
      }
      if (a.d_img) { v[F] = a.d_img[pix]; v[F + 1] = a.d_img[hw + pix]; v[F + 2] = a.d_img[2 * hw + pix]; }
      if (a.d_depth) v[F + 3] = a.d_depth[pix];
    }
    __bf16* rh = s_hi[wave] + lane * MF_LD;
    __bf16* rl = s_lo[wave] + lane * MF_LD;
#pragma unroll
    for (int c8 = 0; c8 < MF_CH / 8; ++c8) {
      bf16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) { hi[e] = (__bf16)v[8 * c8 + e]; lo[e] = (__bf16)(v[8 * c8 + e] - (float)hi[e]); }
      *reinterpret_cast<bf16x8*>(rh + 8 * c8) = hi;
      *reinterpret_cast<bf16x8*>(rl + 8 * c8) = lo;
    }
    const float bdot = a.bg[0] * v[F] + a.bg[1] * v[F + 1] + a.bg[2] * v[F + 2];
    s_pix[wave][lane] = make_float4(Tf, Tf * bdot, __uint_as_float(last), 0.f);
  }
  uint32_t wave_last = last;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, o));
  wave_last = __builtin_amdgcn_readfirstlane(wave_last);    // tell the compiler it is wave-uniform (scalar loop control)
  wave_lds_sync3();
  const float ddx = 0.5f * (float)a.W, ddy = 0.5f * (float)a.H;
  const float bx = (float)(tx * SUB), by = (float)(ty * SUB);
  const __bf16* const ahi = s_hi[wave];
  const __bf16* const alo = s_lo[wave];
  // channel column of this lane for the cot^T gathers: block nb covers channels nb*32 + m; columns >= 48 do not
  // exist -- they are zero, read column 47 instead
  const int col0 = m, col1 = min(32 + m, MF_CH - 1);
  // ---- chunks of 64 list entries, back to front ----------------------------------------------------
  const uint32_t jb = a.hdr[HDR_PACK];
  for (uint32_t c1 = wave_last; c1 > 0; c1 = (c1 > WAVE) ? c1 - WAVE : 0) {
    const uint32_t c0 = (c1 > WAVE) ? c1 - WAVE : 0;
    const uint32_t n = c1 - c0;
    const bool lane_valid = (uint32_t)lane < n;
    const uint32_t pos = lane_valid ? (c1 - 1 - lane) : 0;     // lane 0 = farthest entry of the chunk
    const uint32_t lv = a.pair_slot[range.x + pos];
    const uint32_t id = jb ? (lv >> jb) : a.point_list[range.x + pos];   // HDR_PACK: id and pair index in the list value
    const uint32_t slot = !lane_valid ? 0xffffffffu
                                      : (jb ? __float_as_uint(a.geo[4 * (size_t)id].z) + (lv & ((1u << jb) - 1u)) : lv);
    const float2 gxy = a.xy[id];
    const float4 co = a.conic_o[id];
    const PairPoly k = pair_poly(gxy, co, bx, by);
    const uint32_t pos_cmp = lane_valid ? pos : 0xffffffffu;
    // ---- GEMM 1 (per 32-pixel half, just before the half is visited): s[p][g] ------------------------------
    // a = cot fragment (row = pixel mb*32 + m, 8 channels), b = channel fragment (column = Gaussian nb*32 + m).
    // Lane (m, h) of block nb then holds Gaussian nb*32+m, pixels mb*32 + 8q + 4h + r.  Swapping the upper half of
    // the nb=0 register with the lower half of the nb=1 register leaves lane l with ITS Gaussian l in both:
    // Sx[4q+r] = pixel mb*32 + 8q + r, Sy[4q+r] = pixel mb*32 + 8q + 4 + r.
    const uint32_t idb[2] = {(uint32_t)__shfl((int)id, m), (uint32_t)__shfl((int)id, 32 + m)};
    f32x16 Sx, Sy;
    auto gemm1 = [&](int mb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { Sx[r] = 0.f; Sy[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < MF_CH / 16; ++ks) {
        const bf16x8 ph = *reinterpret_cast<const bf16x8*>(ahi + (mb * 32 + m) * MF_LD + ks * 16 + 8 * h);
        const bf16x8 pl = *reinterpret_cast<const bf16x8*>(alo + (mb * 32 + m) * MF_LD + ks * 16 + 8 * h);
        const __bf16* s0 = a.chan + (size_t)idb[0] * (2 * MF_CH) + ks * 16 + 8 * h;
        const __bf16* s1 = a.chan + (size_t)idb[1] * (2 * MF_CH) + ks * 16 + 8 * h;
        const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(s0), bl0 = *reinterpret_cast<const bf16x8*>(s0 + MF_CH);
        const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(s1), bl1 = *reinterpret_cast<const bf16x8*>(s1 + MF_CH);
        Sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, bh0, Sx, 0, 0, 0);
        Sy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, bh1, Sy, 0, 0, 0);
        Sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, bl0, Sx, 0, 0, 0);
        Sy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph, bl1, Sy, 0, 0, 0);
        Sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pl, bh0, Sx, 0, 0, 0);
        Sy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pl, bh1, Sy, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(Sx[r]), __float_as_uint(Sy[r]), false, false);
        Sx[r] = __uint_as_float(sw[0]);
        Sy[r] = __uint_as_float(sw[1]);
      }
    };
    // ---- the sequential part, pixel by pixel; GEMM 2 after every 16 pixels -----------------------------
    f32x16 D[2][2];                                      // D[g block][channel block], accumulated over the 64 pixels
#pragma unroll
    for (int gb = 0; gb < 2; ++gb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) D[gb][nb][r] = 0.f;
    float S0 = 0.f, Sj = 0.f, Si = 0.f, Sjj = 0.f, Sij = 0.f, Sii = 0.f;
    unsigned wh[8], wl[8];                               // 16 weights of the current K-step, packed bf16 pairs
    // two passes over 32 pixels (rolled: halves the code -- 12 resident waves share the instruction cache), four
    // pixel rows unrolled inside so that every register array index below is a compile-time constant
#pragma unroll 1
    for (int mb = 0; mb < 2; ++mb) {
      if constexpr (!FEAT_ONLY) gemm1(mb);
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
      const int i = mb * 4 + ii;
      const float fi = (float)i, fii = (float)(i * i);
      const float base = poly_row_base(k, fi, fii);
      const float slope = poly_row_slope(k, fi);
      float R0 = 0.f, R1 = 0.f, R2 = 0.f;            // row sums of q, q*j, q*j^2
#pragma unroll
      for (int j = 0; j < SUB; j += 2) {                 // two pixels per step: their scans are interleaved
        const int p = i * SUB + j;
        // the pixels' last-contributor indices never change: they stay in a register (lane = pixel) and are read
        // with v_readlane, so the skip test below does not wait for LDS
        const uint32_t lasta = (uint32_t)__builtin_amdgcn_readlane((int)last, p);
        const uint32_t lastb = (uint32_t)__builtin_amdgcn_readlane((int)last, p + 1);
        const float4 pa = s_pix[wave][p], pb = s_pix[wave][p + 1];   // uniform reads: T_end, U_end
        float wa = 0.f, wb = 0.f;
        // a pixel whose last blended entry lies behind this chunk passes no gate (pos >= c0 >= last): processing it
        // next to a live neighbour changes nothing, so the skip is per pair
        if (max(lasta, lastb) > c0) {
          const float ea = poly_eval(k, base, slope, (float)j);
          const float eb = poly_eval(k, base, slope, (float)(j + 1));
          const bool oka = (ea <= k.thr) && (ea >= LOG2_ALPHA_MIN) && (pos_cmp < lasta);
          const bool okb = (eb <= k.thr) && (eb >= LOG2_ALPHA_MIN) && (pos_cmp < lastb);
          // a closed gate is exp2(-inf) = 0: alpha, the weight and q = araw * dL/dalpha all vanish without selects
          const float ra = __builtin_amdgcn_exp2f(oka ? ea : -INFINITY);   // opacity * exp(power)
          const float rb = __builtin_amdgcn_exp2f(okb ? eb : -INFINITY);
          const float ala = fminf(ALPHA_MAX, ra), alb = fminf(ALPHA_MAX, rb);
          // T = T_end / prod(1 - alpha) as T_end * prod(1 / (1 - alpha)): the reciprocal is needed anyway (dL/dalpha)
          const float roma = __builtin_amdgcn_rcpf(1.0f - ala), romb = __builtin_amdgcn_rcpf(1.0f - alb);
          float Pa = roma, Pb = romb;
          wave_scan_mul2_asm(Pa, Pb);
          const float Ta = pa.x * Pa, Tb = pb.x * Pb;     // transmittance in front of this Gaussian
          wa = ala * Ta; wb = alb * Tb;
          if constexpr (FEAT_ONLY) {
            if (lane == WAVE - 1) { s_pix[wave][p].x = Ta; s_pix[wave][p + 1].x = Tb; }   // carries for the next (nearer) chunk
          } else {
          const float sa = (j < 4 ? Sx : Sy)[4 * ii + (j & 3)];
          const float sb = (j < 4 ? Sx : Sy)[4 * ii + ((j + 1) & 3)];
          const float wsa = wa * sa, wsb = wb * sb;
          float ia, ib;
          wave_scan_add2_out_asm(wsa, wsb, ia, ib);
          const float Ua = pa.y + (ia - wsa), Ub = pb.y + (ib - wsb);
          const float dLa = Ta * sa - Ua * roma, dLb = Tb * sb - Ub * romb;
          if (lane == WAVE - 1) {                        // carries for the next (nearer) chunk
            s_pix[wave][p].x = Ta;     s_pix[wave][p].y = pa.y + ia;
            s_pix[wave][p + 1].x = Tb; s_pix[wave][p + 1].y = pb.y + ib;
          }
          const float qa = ra * dLa, qb = rb * dLb;      // == opacity * G * dL/dalpha (straight-through clamp)
          R0 += qa + qb;
          R1 = fmaf(qb, (float)(j + 1), fmaf(qa, (float)j, R1));
          R2 = fmaf(qb, (float)((j + 1) * (j + 1)), fmaf(qa, (float)(j * j), R2));
          }
        }
        {                                                // split the two weights, packed: w = hi + lo
          unsigned hb, lb;
          asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hb) : "v"(wa), "v"(wb));
          const float ra2 = wa - __uint_as_float(hb << 16), rb2 = wb - __uint_as_float(hb & 0xffff0000u);
          asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lb) : "v"(ra2), "v"(rb2));
          wh[(8 * (ii & 1) + j) >> 1] = hb;             // position inside the 16-pixel K-step
          wl[(8 * (ii & 1) + j) >> 1] = lb;
        }
      }
      S0 += R0; Sj += R1; Sjj += R2;
      Si = fmaf(fi, R0, Si); Sii = fmaf(fii, R0, Sii); Sij = fmaf(fi, R1, Sij);
      if (ii & 1) {
        // ---- GEMM 2, K-step t = i/2: pixels 16t .. 16t+15 -----------------------------------------------
        // A fragments: block gb covers Gaussians gb*32 + m; lanes of half h supply pixels 8h .. 8h+7 of the step.
        // Own registers: wh[0..3] = pixels 0..7, wh[4..7] = pixels 8..15 of the lane's own Gaussian.
        bf16x8 Ah[2], Al[2];
        {
          u32x4 h0, h1, l0, l1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const auto sh = __builtin_amdgcn_permlane32_swap(wh[r], wh[4 + r], false, false);
            const auto sl = __builtin_amdgcn_permlane32_swap(wl[r], wl[4 + r], false, false);
            h0[r] = sh[0]; h1[r] = sh[1]; l0[r] = sl[0]; l1[r] = sl[1];
          }
          Ah[0] = __builtin_bit_cast(bf16x8, h0); Ah[1] = __builtin_bit_cast(bf16x8, h1);
          Al[0] = __builtin_bit_cast(bf16x8, l0); Al[1] = __builtin_bit_cast(bf16x8, l1);
        }
        const int rowoff = ((i >> 1) * 16 + 8 * h) * MF_LD;    // first pixel row of this lane half in the K-step
        constexpr int NBLK = FEAT_ONLY ? 1 : 2;            // channel block 1 = r g b depth
        bf16x8 Bh[2], Bl[2];
        Bh[0] = gather_column(ahi + rowoff + col0); Bl[0] = gather_column(alo + rowoff + col0);
        if constexpr (!FEAT_ONLY) { Bh[1] = gather_column(ahi + rowoff + col1); Bl[1] = gather_column(alo + rowoff + col1); }
#pragma unroll
        for (int gb = 0; gb < 2; ++gb)
#pragma unroll
          for (int nb = 0; nb < NBLK; ++nb) {
            // rows = channels (cot^T fragment), columns = Gaussians (weight fragment): the result has lane = Gaussian
            D[gb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bh[nb], Ah[gb], D[gb][nb], 0, 0, 0);
            D[gb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bl[nb], Ah[gb], D[gb][nb], 0, 0, 0);
            D[gb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Bh[nb], Al[gb], D[gb][nb], 0, 0, 0);
          }
      }
      }
    }
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
    // ---- one row per pair: [32 feature sums | nx ny ca cb | cc op r g | b d 0 0] -----------------------
    if (lane_valid) {
      float* row = a.rows + (size_t)slot * bwd_row_stride(F) + F;
      *reinterpret_cast<float4*>(row) = make_float4(a_nx * ddx, a_ny * ddy, a_ca, a_cb);
      *reinterpret_cast<float2*>(row + 4) = make_float2(a_cc, a_op);
      *reinterpret_cast<float2*>(row + 10) = make_float2(0.f, 0.f);
      a.row_flags[slot] = 1;
    }
    // D[gb][nb]: lane (m, h) = Gaussian gb*32 + m, register 4q + r = channel nb*32 + 8q + 4h + r: four consecutive
    // channels per register quad.  The slot of Gaussian gb*32 + m comes from its owner lane with one swap.
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(slot, slot, false, false);
      const uint32_t sl[2] = {sw[0], sw[1]};           // slots of Gaussians m and 32 + m
#pragma unroll
      for (int gb = 0; gb < 2; ++gb) {
        if (sl[gb] != 0xffffffffu) {
          float* row = a.rows + (size_t)sl[gb] * bwd_row_stride(F);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(row + 8 * q + 4 * h) =
                make_float4(D[gb][0][4 * q], D[gb][0][4 * q + 1], D[gb][0][4 * q + 2], D[gb][0][4 * q + 3]);
          if (h == 0) {                                  // channels 32..35: r g b depth sums
            *reinterpret_cast<float2*>(row + F + 6) = make_float2(D[gb][1][0], D[gb][1][1]);
            *reinterpret_cast<float2*>(row + F + 8) = make_float2(D[gb][1][2], D[gb][1][3]);
          }
        }
      }
    }
    wave_lds_sync3();                                     // carries written by lane 63 are read by the next chunk
  }
}

// per view: channel table for the MFMA backward kernels + clears the row flags (saves a fill launch)
int launch_split_channels(const LaunchCtx& c, const TraseRastInputs& in, const GeomBuf& g, void* chan, uint8_t* row_flags,
                          size_t flag_bytes) {
  {
    ProfScope ps("split_channels", c.stream);
    hipLaunchKernelGGL(split_channels_kernel, dim3((in.P * 6 + 255) / 256), dim3(256), 0, c.stream, in.sh_objs, g.rgbd, g.tiles, in.P,
                       (__bf16*)chan, reinterpret_cast<uint4*>(row_flags), flag_bytes / 16);
  }
  TRASE_POST_LAUNCH("split_channels", c.stream, c.debug);
  return TRASE_OK;
}

int launch_render_bwd_mf(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, const GeomBuf& g,
                         const BinBuf& bb, const ImgBuf& im, const TraseRastGrads& gr, float* rows, uint8_t* row_flags,
                         void* chan, size_t flag_bytes) {
  BwdMfArgs a;
  a.ranges = bb.ranges; a.point_list = bb.point_list; a.xy = g.xy; a.conic_o = g.conic_o; a.bg = s.bg;
  a.d_img = gr.dL_dimage; a.d_feat = gr.dL_dfeats; a.d_depth = gr.dL_ddepth;
  a.final_T = im.final_T; a.n_contrib = im.n_contrib; a.pair_slot = bb.pair_slot; a.hdr = g.hdr; a.geo = g.geo;
  a.chan = (const __bf16*)chan; a.rows = rows; a.row_flags = row_flags;
  a.W = s.image_width; a.H = s.image_height;
  a.gx8 = (a.W + SUB - 1) / SUB;
  { int lo, hi; strip_subtile_rows(s, lo, hi); a.tile0 = lo * a.gx8; a.ntiles = (hi - lo) * a.gx8; }
  int rc = launch_split_channels(c, in, g, chan, row_flags, flag_bytes);
  if (rc) return rc;
  if (a.ntiles <= 0) return TRASE_OK;                    // an empty strip: no rows (the flags are cleared)
  {
    ProfScope ps("render_bwd", c.stream);
    const dim3 grid((a.ntiles + MF_WPB - 1) / MF_WPB), block(MF_WPB * WAVE);
    if (c.variant & 0x400) hipLaunchKernelGGL(render_bwd_mf_kernel<true>, grid, block, 0, c.stream, a);   // feature gradients only
    else hipLaunchKernelGGL(render_bwd_mf_kernel<false>, grid, block, 0, c.stream, a);
  }
  TRASE_POST_LAUNCH("render_bwd", c.stream, c.debug);
  return TRASE_OK;
}

}  // namespace trase
