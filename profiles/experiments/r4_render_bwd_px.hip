// render_bwd_px.hip -- backward of the compositing stage, lane = PIXEL formulation (experiment, TRASE_VARIANT_AB_PX_BACKWARD).
//
// render_bwd_hw.hip keeps a Gaussian per lane and pays for what is sequential along the list with two DPP scans per pixel
// (20 DPP instructions + their wait states per pixel pair: a third of its VALU work).  Here the organisation of the
// FORWARD kernel is used instead: lane l = (m = l & 31, h = l >> 5) owns pixel m of a 32-pixel block and half h of every
// 16-entry K-step; the transmittance recurrence runs inside the lane over its eight entries (back to front), the two
// halves exchange their totals with v_permlane32_swap, and nothing is scanned across lanes.  What has to be summed over
// PIXELS -- the channel gradients dchan[c][g] = sum_p cot[p][c] w[p][g] and the six geometry moments
// sum_p phi_k(p) q[p][g] -- runs on the matrix cores with the pixel index as K: w and q = alpha_raw * dL/dalpha leave the
// lanes through a 32 x 32 bf16 hi/lo tile in LDS and come back transposed (ds_read_b64_tr_b16) as B fragments.
//   * a chunk is 32 list entries; the MFMA row rho of "entry" operands / results is the list entry
//     j(rho) = 16 (rho >> 4) + 8 ((rho >> 2) & 1) + 4 ((rho >> 3) & 1) + (rho & 3), so that registers 8tt .. 8tt+7 of a
//     32 x 32 result are the lane's eight entries 16tt + 8h + u of K-step tt;
//   * GEMM 1: S^T[rho][pixel] = <chan[rho], cot[pixel]> (A = channel fragments straight from ftab / the geometry record,
//     B = the cotangent image) -- lane (pixel, h) receives s for exactly its 16 entries;
//   * GEMM 2 / GEMM 3 per 32-pixel block; rows, flags and everything downstream as in render_bwd_hw.hip.
#include "common.h"

namespace trase {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int PX_WPB = 2;      // waves (sub-tiles) per workgroup
constexpr int PX_LD = 40;      // cotangent image pitch (bf16): 36 channels + 4 zeros
constexpr int PX_PT = 40;      // w / q tile pitch (bf16): 32 entries + 8 (the transposing reads then spread over the banks)
constexpr int PX_G = 32;       // list entries per chunk

struct PxWaveLds {
  __bf16 chi[WAVE * PX_LD];    // cot image, pixel-major
  __bf16 clo[WAVE * PX_LD];
  __bf16 thi[32 * PX_PT];      // [pixel of the block][entry row rho]: w, then q
  __bf16 tlo[32 * PX_PT];
  float kp[PX_G / 2][16];      // exponent polynomials, one 64-byte record per PAIR of list entries (render_fwd_mf.hip layout)
};

struct BwdPxArgs {
  const uint2* ranges; const uint32_t* point_list;
  const float* bg;
  const float* d_img; const float* d_feat; const float* d_depth;
  const float* final_T; const uint32_t* n_contrib;
  const uint32_t* pair_slot;
  const uint32_t* hdr;
  const float4* geo;
  const uint32_t* ftab;
  float* rows;
  uint8_t* row_flags;
  int W, H, gx8, ntiles, tile0;
  int lineage;
  float feat_bg;
  const float* out_depth;
};

__device__ __forceinline__ void wave_lds_sync_px() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ f2v px_pk_fma_b0(f2v a, f2v x, f2v y) {
  f2v d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(x), "v"(y));
  return d;
}
__device__ __forceinline__ f2v px_pk_fma_b1(f2v a, f2v x, f2v y) {
  f2v d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(x), "v"(y));
  return d;
}

// eight consecutive rows of one column from a row-major bf16 tile: two transposing reads (rows +0..3, +4..7); `p` is this
// lane's chunk address for the first four rows
template <int PITCH>
__device__ __forceinline__ bf16x8 px_tr8(const __bf16* __restrict__ p) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * PITCH));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ int px_jmap(int rho) { return 16 * (rho >> 4) + 8 * ((rho >> 2) & 1) + 4 * ((rho >> 3) & 1) + (rho & 3); }

__global__ __launch_bounds__(PX_WPB* WAVE) __attribute__((amdgpu_waves_per_eu(2, 3)))
void render_bwd_px_kernel(BwdPxArgs a) {
  constexpr int F = 32;
  __shared__ __attribute__((aligned(16))) PxWaveLds s_w[PX_WPB];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  const int local = xcd_block(blockIdx.x, gridDim.x) * PX_WPB + wave;
  if (local >= a.ntiles) return;
  int tx, ty;
  blocked_tile<16>(local, a.gx8, a.ntiles / a.gx8, tx, ty);
  ty += a.tile0 / a.gx8;
  const int tile = ty * a.gx8 + tx;
  const uint2 range = a.ranges[tile];
  PxWaveLds& L = s_w[wave];
  // ---- stage the sub-tile's cotangents (lane = pixel) ---------------------------------------------------------------
  float Tst[2], Ust[2];          // per pixel of this lane (block mb): transmittance / U behind the entries walked so far
  int lastp[2];
  uint32_t wave_last;
  int blk_last[2];
  {
    const int px = tx * SUB + (lane & 7), py = ty * SUB + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const size_t hw = (size_t)a.H * a.W;
    const size_t pix = (size_t)py * a.W + px;
    float v[PX_LD];
#pragma unroll
    for (int c = 0; c < PX_LD; ++c) v[c] = 0.f;
    float Tf = 0.f;
    uint32_t last = 0;
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
    float bextra = 0.f;
    if (a.lineage & TRASE_VARIANT_FEATS_BG) {
      float sf = 0.f;
#pragma unroll
      for (int c = 0; c < F; ++c) sf += v[c];
      bextra = a.feat_bg * sf;
    }
    if ((a.lineage & TRASE_VARIANT_DEPTH_NORM) && a.d_depth && inside) {
      const float A = 1.0f - Tf, gd = v[F + 3];
      const float ga = A > 1e-10f ? gd / A : 0.0f;
      v[F + 3] = ga;
      bextra = fmaf(ga, a.out_depth[pix], bextra);
    }
    const float bdot = a.bg[0] * v[F] + a.bg[1] * v[F + 1] + a.bg[2] * v[F + 2] + bextra;
    __bf16* rh = L.chi + lane * PX_LD;
    __bf16* rl = L.clo + lane * PX_LD;
#pragma unroll
    for (int c8 = 0; c8 < PX_LD / 8; ++c8) {
      bf16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) { hi[e] = (__bf16)v[8 * c8 + e]; lo[e] = (__bf16)(v[8 * c8 + e] - (float)hi[e]); }
      *reinterpret_cast<u32x4*>(rh + 8 * c8) = __builtin_bit_cast(u32x4, hi);
      *reinterpret_cast<u32x4*>(rl + 8 * c8) = __builtin_bit_cast(u32x4, lo);
    }
    const float U0 = Tf * bdot;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      Tst[mb] = __shfl(Tf, mb * 32 + m);
      Ust[mb] = __shfl(U0, mb * 32 + m);
      lastp[mb] = __shfl((int)last, mb * 32 + m);
    }
    uint32_t wl = last;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) wl = max(wl, (uint32_t)__shfl_xor((int)wl, o));     // max over each 32-lane half = pixel block
    blk_last[0] = __builtin_amdgcn_readlane((int)wl, 0);
    blk_last[1] = __builtin_amdgcn_readlane((int)wl, 32);
    wave_last = (uint32_t)max(blk_last[0], blk_last[1]);
  }
  wave_lds_sync_px();
  const float ddx = 0.5f * (float)a.W, ddy = 0.5f * (float)a.H;
  const float bx = (float)(tx * SUB), by = (float)(ty * SUB);
  const uint32_t jb = a.hdr[HDR_PACK];
  const int jmine = px_jmap(m);                       // list entry (within the chunk) whose parameters this lane holds
  // pixel constants of the packed polynomial evaluation: column j = m & 7, row i = 4 mb + (m >> 3)
  const float fj = (float)(m & 7);
  f2v FI2[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) { const float fi = (float)(4 * mb + (m >> 3)); FI2[mb] = (f2v){fi, fi * fi}; }
  const f2v FJ2 = {fj, 0.f};
  // transposing-read addressing (16-lane group G = lane >> 4: G & 1 = column half, G >> 1 = h)
  const int li = lane & 15, trj = li >> 2, trc = li & 3, ghalf = (lane >> 4) & 1;
  const int tile_tr = (8 * h + trj) * PX_PT + 16 * ghalf + 4 * trc;                    // + 16 ks2 * PX_PT
  const int cot_tr0 = (8 * h + trj) * PX_LD + 16 * ghalf + 4 * trc;                    // channel block 0; + (mb*32 + 16 ks2) * PX_LD
  const int cot_tr1 = (8 * h + trj) * PX_LD + min(32 + 16 * ghalf + 4 * trc, PX_LD - 4);
  // GEMM 3 A fragments: row r = m (monomial: 1, j, i, jj, ij, ii; rows >= 6 zero), k = pixel 16 ks2 + 8h + e of the block:
  // that pixel's column is e, its row 4 mb + 2 ks2 + h
  auto phi_frag = [&](int mb, int ks2) {
    const float fi = (float)(4 * mb + 2 * ks2 + h);
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float fe = (float)e;
      const float val = m == 0 ? 1.0f : m == 1 ? fe : m == 2 ? fi : m == 3 ? fe * fe : m == 4 ? fi * fe : m == 5 ? fi * fi : 0.0f;
      f[e] = (__bf16)val;                                                              // small integers: exact in bf16
    }
    return f;
  };
  bf16x8 PHI[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) PHI[mb][ks2] = phi_frag(mb, ks2);
  // Software pipeline over the chunks (the kernel runs two waves per SIMD: nobody else hides a chunk's dependent loads):
  // list values two chunks ahead, the per-entry records (geometry, colour, channel fragments) one chunk ahead.
  struct EntryRegs { float4 gq, co, cs; bf16x8 fbh[2], fbl[2]; uint32_t lv; bool valid; };
  auto chunk_lo = [](uint32_t hi_) { return hi_ > (uint32_t)PX_G ? hi_ - (uint32_t)PX_G : 0u; };
  auto load_lv = [&](uint32_t c1_) -> uint32_t {             // list value of this lane's entry of the chunk ending at c1_
    const uint32_t c0_ = chunk_lo(c1_);
    return (c1_ > 0 && (uint32_t)jmine < c1_ - c0_) ? a.pair_slot[range.x + c0_ + jmine] : 0u;
  };
  auto load_entry = [&](uint32_t c1_, uint32_t lv_, EntryRegs& e) {
    const uint32_t c0_ = chunk_lo(c1_);
    e.valid = c1_ > 0 && (uint32_t)jmine < c1_ - c0_;
    e.lv = lv_;
    const uint32_t id = e.valid ? (jb ? (lv_ >> jb) : a.point_list[range.x + c0_ + jmine]) : 0u;
    e.gq = a.geo[4 * (size_t)id];
    e.co = a.geo[4 * (size_t)id + 1];
    e.cs = a.geo[4 * (size_t)id + 3];
    const uint32_t* const frow = a.ftab + (size_t)id * 32 + 4 * h;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      e.fbh[ks] = *reinterpret_cast<const bf16x8*>(frow + ks * 8);
      e.fbl[ks] = *reinterpret_cast<const bf16x8*>(frow + ks * 8 + 16);
    }
  };
  EntryRegs cur, nxt;
  uint32_t lv_b;                                              // list value of the NEXT chunk's entry
  {
    const uint32_t lv_a = load_lv(wave_last);
    lv_b = load_lv(chunk_lo(wave_last));
    load_entry(wave_last, lv_a, cur);
  }
  struct PendingRow { uint32_t slot; bool write; float4 d[4]; float4 mo[3]; };
  PendingRow pend;
  pend.slot = 0xffffffffu; pend.write = false;
  auto store_pending = [&]() {
    if (pend.write) {
      float* row = a.rows + (size_t)pend.slot * bwd_row_stride(F);
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(row + 8 * q + 4 * h) = pend.d[q];
      if (h == 0) {
        *reinterpret_cast<float4*>(row + F) = pend.mo[0];
        *reinterpret_cast<float4*>(row + F + 4) = pend.mo[1];
        *reinterpret_cast<float4*>(row + F + 8) = pend.mo[2];
        a.row_flags[pend.slot] = 1;
      }
    }
  };
  const f32x16 ZERO16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // ---- chunks of 32 list entries, back to front -------------------------------------------------------------------------
  for (uint32_t c1 = wave_last; c1 > 0; c1 = (c1 > PX_G) ? c1 - PX_G : 0) {
    const uint32_t c0 = (c1 > PX_G) ? c1 - PX_G : 0;
    const uint32_t n = c1 - c0;
    const bool lane_valid = cur.valid;
    const uint32_t lv = cur.lv;
    const float4 gq = cur.gq, co = cur.co, cs = cur.cs;
    bf16x8 fbh[2] = {cur.fbh[0], cur.fbh[1]}, fbl[2] = {cur.fbl[0], cur.fbl[1]};
    // requests for the chunks ahead: the next one's records (its list value arrived a chunk ago), the list value after that
    load_entry(c0, lv_b, nxt);
    lv_b = load_lv(chunk_lo(c0));
    store_pending();
    const float2 gxy = make_float2(gq.x, gq.y);
    const uint32_t slot = !lane_valid ? 0xffffffffu : (jb ? __float_as_uint(gq.z) + (lv & ((1u << jb) - 1u)) : lv);
    if (!lane_valid) {                                       // entries past the chunk: zero channels (0 * garbage could be NaN)
      const bf16x8 z = __builtin_bit_cast(bf16x8, (u32x4){0u, 0u, 0u, 0u});
      fbh[0] = fbh[1] = fbl[0] = fbl[1] = z;
    }
    if (h == 0) {                                            // exponent polynomial of entry jmine -> its pair record
      float* const rec = &L.kp[jmine >> 1][jmine & 1];
      if (lane_valid) {
        const PairPoly k = pair_poly(gxy, co, bx, by);
        rec[0] = k.k0; rec[2] = k.kj; rec[4] = k.ki; rec[6] = k.kjj; rec[8] = k.kii; rec[10] = k.kij; rec[12] = k.thr;
      } else {
        rec[0] = 0.f; rec[2] = 0.f; rec[4] = 0.f; rec[6] = 0.f; rec[8] = 0.f; rec[10] = 0.f; rec[12] = -INFINITY;
      }
    }
    wave_lds_sync_px();
    f32x16 D[2] = {ZERO16, ZERO16};
    f32x16 Mo = ZERO16;
    uint32_t bm = 0;                                         // bit j: some pixel blended list entry j of this chunk (wave-uniform)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      if (blk_last[mb] <= (int)c0) continue;                 // no pixel of this block blends anything in this chunk (wave-uniform)
      // ---- GEMM 1: S^T[rho][pixel] -------------------------------------------------------------------------------
      f32x16 S;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        bf16x8 ah, al;
        if (ks < 2) { ah = fbh[ks]; al = fbl[ks]; }
        else {
          const bool on = lane_valid && h == 0;
          const u32x4 ch4 = {on ? __float_as_uint(cs.x) : 0u, on ? __float_as_uint(cs.y) : 0u, 0u, 0u};
          const u32x4 cl4 = {on ? __float_as_uint(cs.z) : 0u, on ? __float_as_uint(cs.w) : 0u, 0u, 0u};
          ah = __builtin_bit_cast(bf16x8, ch4); al = __builtin_bit_cast(bf16x8, cl4);
        }
        // channels ks*16 + 8h ..: for ks == 2, h == 1 they do not exist (the A fragment is zero there); the last pixel row
        // reads its own first columns instead of running past the image
        const int off = (ks == 2 && mb == 1 && lane == 63) ? 63 * PX_LD : (mb * 32 + m) * PX_LD + ks * 16 + 8 * h;
        const bf16x8 ph = *reinterpret_cast<const bf16x8*>(L.chi + off), pl = *reinterpret_cast<const bf16x8*>(L.clo + off);
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ph, ks == 0 ? ZERO16 : S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, pl, S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ph, S, 0, 0, 0);
      }
      // ---- the lane's 16 entries of this pixel, back to front: K-step 1 (entries 16..31), then K-step 0 -----------
      float Tb = Tst[mb], Ub = Ust[mb];
      float q16[16];
#pragma unroll
      for (int tt = 1; tt >= 0; --tt) {
        if ((uint32_t)(16 * tt) >= n) {                      // (wave-uniform) no entry in this K-step
#pragma unroll
          for (int u = 0; u < 8; ++u) q16[8 * tt + u] = 0.f;
          const u32x4 z = {0u, 0u, 0u, 0u};
          *reinterpret_cast<uint2*>(L.thi + m * PX_PT + 16 * tt + 4 * h) = make_uint2(z[0], z[1]);
          *reinterpret_cast<uint2*>(L.thi + m * PX_PT + 16 * tt + 8 + 4 * h) = make_uint2(z[0], z[1]);
          *reinterpret_cast<uint2*>(L.tlo + m * PX_PT + 16 * tt + 4 * h) = make_uint2(z[0], z[1]);
          *reinterpret_cast<uint2*>(L.tlo + m * PX_PT + 16 * tt + 8 + 4 * h) = make_uint2(z[0], z[1]);
          continue;
        }
        const int rel = lastp[mb] - (int)c0 - 16 * tt - 8 * h;     // entry u of this lane is in front of the pixel's last contributor iff u < rel
        float raw[8], rr[8], rho[8], sig[8], al8[8];
        // exponents: two entries per packed FMA, the same trees as poly_row_base / poly_row_slope / poly_eval
        float ex8[8], thr8[8];
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          const f4v* const rec = reinterpret_cast<const f4v*>(&L.kp[(16 * tt + 8 * h + u) >> 1][0]);
          const f4v r0 = rec[0], r1 = rec[1], r2 = rec[2];
          const f2v thr2 = *reinterpret_cast<const f2v*>(&rec[3]);
          f2v bs = px_pk_fma_b0(FI2[mb], __builtin_shufflevector(r1, r1, 0, 1), __builtin_shufflevector(r0, r0, 0, 1));
          bs = px_pk_fma_b1(FI2[mb], __builtin_shufflevector(r2, r2, 0, 1), bs);
          const f2v sl = px_pk_fma_b0(FI2[mb], __builtin_shufflevector(r2, r2, 2, 3), __builtin_shufflevector(r0, r0, 2, 3));
          const f2v in = px_pk_fma_b0(FJ2, __builtin_shufflevector(r1, r1, 2, 3), sl);
          const f2v e2 = px_pk_fma_b0(FJ2, in, bs);
          ex8[u] = e2[0]; ex8[u + 1] = e2[1]; thr8[u] = thr2[0]; thr8[u + 1] = thr2[1];
        }
        float run_rho = 1.0f, run_sig = 0.0f;
#pragma unroll
        for (int u = 7; u >= 0; --u) {
          const bool gate = (ex8[u] <= thr8[u]) && (ex8[u] >= LOG2_ALPHA_MIN) && (u < rel);
          const unsigned long long bal = __ballot(gate);
          bm |= ((uint32_t)bal != 0u ? 1u : 0u) << (16 * tt + u);
          bm |= ((uint32_t)(bal >> 32) != 0u ? 1u : 0u) << (16 * tt + 8 + u);
          raw[u] = __builtin_amdgcn_exp2f(gate ? ex8[u] : -INFINITY);
          al8[u] = fminf(ALPHA_MAX, raw[u]);
          rr[u] = __builtin_amdgcn_rcpf(1.0f - al8[u]);
          run_rho *= rr[u];
          rho[u] = run_rho;                                  // prod of 1/(1-alpha) from the far end of this half down to u
          sig[u] = run_sig;                                  // sum over the entries behind u (in this half) of alpha rho s
          run_sig = fmaf(al8[u] * run_rho, S[8 * tt + u], run_sig);
        }
        // totals of the two halves: far = h == 1 (entries 8..15 of the K-step), near = h == 0
        const auto swr = __builtin_amdgcn_permlane32_swap(__float_as_uint(run_rho), __float_as_uint(run_rho), false, false);
        const auto sws = __builtin_amdgcn_permlane32_swap(__float_as_uint(run_sig), __float_as_uint(run_sig), false, false);
        const float rho_near = __uint_as_float(swr[0]), rho_far = __uint_as_float(swr[1]);
        const float sig_near = __uint_as_float(sws[0]), sig_far = __uint_as_float(sws[1]);
        const float Tstart = h ? Tb : Tb * rho_far;
        const float Ustart = h ? Ub : fmaf(Tb, sig_far, Ub);
        float w8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float Tu = Tstart * rho[u];                  // transmittance in front of this entry
          w8[u] = al8[u] * Tu;
          const float Uu = fmaf(Tstart, sig[u], Ustart);
          const float dL = fmaf(Tu, S[8 * tt + u], -(Uu * rr[u]));
          q16[8 * tt + u] = raw[u] * dL;                     // == opacity * G * dL/dalpha (straight-through clamp)
        }
        Ub = fmaf(Tb * rho_far, sig_near, fmaf(Tb, sig_far, Ub));
        Tb = (Tb * rho_far) * rho_near;
        // w of this K-step -> tile[pixel m][rho]: the lane's entries are rows 16tt + 4h + {0..3} and 16tt + 8 + 4h + {0..3}
        {
          unsigned hh[4], ll[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) split_pk(w8[2 * v], w8[2 * v + 1], hh[v], ll[v]);
          *reinterpret_cast<uint2*>(L.thi + m * PX_PT + 16 * tt + 4 * h) = make_uint2(hh[0], hh[1]);
          *reinterpret_cast<uint2*>(L.thi + m * PX_PT + 16 * tt + 8 + 4 * h) = make_uint2(hh[2], hh[3]);
          *reinterpret_cast<uint2*>(L.tlo + m * PX_PT + 16 * tt + 4 * h) = make_uint2(ll[0], ll[1]);
          *reinterpret_cast<uint2*>(L.tlo + m * PX_PT + 16 * tt + 8 + 4 * h) = make_uint2(ll[2], ll[3]);
        }
      }
      Tst[mb] = Tb; Ust[mb] = Ub;
      wave_lds_sync_px();
      // ---- GEMM 2: D[nb][c][rho] += sum over the block's 32 pixels of cot[p][c] w[p][rho] --------------------------
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        const bf16x8 Bh = px_tr8<PX_PT>(L.thi + tile_tr + 16 * ks2 * PX_PT), Bl = px_tr8<PX_PT>(L.tlo + tile_tr + 16 * ks2 * PX_PT);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int o = (mb * 32 + 16 * ks2) * PX_LD + (nb == 0 ? cot_tr0 : cot_tr1);
          const bf16x8 Ah = px_tr8<PX_LD>(L.chi + o), Al = px_tr8<PX_LD>(L.clo + o);
          D[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, D[nb], 0, 0, 0);
          D[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, D[nb], 0, 0, 0);
          D[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, D[nb], 0, 0, 0);
        }
      }
      wave_lds_sync_px();
      // ---- q -> the same tile; GEMM 3: Mo[k][rho] += sum_p phi_k(p) q[p][rho] -------------------------------------------
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        unsigned hh[4], ll[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) split_pk(q16[8 * tt + 2 * v], q16[8 * tt + 2 * v + 1], hh[v], ll[v]);
        *reinterpret_cast<uint2*>(L.thi + m * PX_PT + 16 * tt + 4 * h) = make_uint2(hh[0], hh[1]);
        *reinterpret_cast<uint2*>(L.thi + m * PX_PT + 16 * tt + 8 + 4 * h) = make_uint2(hh[2], hh[3]);
        *reinterpret_cast<uint2*>(L.tlo + m * PX_PT + 16 * tt + 4 * h) = make_uint2(ll[0], ll[1]);
        *reinterpret_cast<uint2*>(L.tlo + m * PX_PT + 16 * tt + 8 + 4 * h) = make_uint2(ll[2], ll[3]);
      }
      wave_lds_sync_px();
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        const bf16x8 Bh = px_tr8<PX_PT>(L.thi + tile_tr + 16 * ks2 * PX_PT), Bl = px_tr8<PX_PT>(L.tlo + tile_tr + 16 * ks2 * PX_PT);
        const bf16x8 Ap = PHI[mb][ks2];
        Mo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ap, Bh, Mo, 0, 0, 0);
        Mo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ap, Bl, Mo, 0, 0, 0);
      }
      wave_lds_sync_px();                                    // the tile is rewritten by the next block / chunk
    }
    // ---- one row per pair: [32 feature sums | nx ny ca cb | cc op r g | b d 0 0] ------------------------------------
    // Mo: lane (rho, h), register 4q + r = monomial 8q + 4h + r: h = 0 holds S0 Sj Si Sjj, h = 1 holds Sij Sii
    auto half0 = [](float x) { const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false); return __uint_as_float(sw[0]); };
    auto half1 = [](float x) { const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false); return __uint_as_float(sw[1]); };
    const float S0 = half0(Mo[0]), Sj = half0(Mo[1]), Si = half0(Mo[2]), Sjj = half0(Mo[3]);
    const float Sij = half1(Mo[0]), Sii = half1(Mo[1]);
    pend.slot = slot;
    pend.write = slot != 0xffffffffu && ((bm >> jmine) & 1u);
#pragma unroll
    for (int q = 0; q < 4; ++q) pend.d[q] = make_float4(D[0][4 * q], D[0][4 * q + 1], D[0][4 * q + 2], D[0][4 * q + 3]);
    {
      const float rx = gxy.x - bx, ry = gxy.y - by;
      const float Qx = rx * S0 - Sj, Qy = ry * S0 - Si;
      const float Qxx = rx * (rx * S0 - 2.0f * Sj) + Sjj;
      const float Qyy = ry * (ry * S0 - 2.0f * Si) + Sii;
      const float Qxy = rx * (ry * S0 - Si) - ry * Sj + Sij;
      const float a_nx = -(co.x * Qx + co.y * Qy);
      const float a_ny = -(co.z * Qy + co.y * Qx);
      const float a_ca = -0.5f * Qxx, a_cb = -Qxy, a_cc = -0.5f * Qyy;
      const float a_op = (co.w > 0.0f) ? S0 / co.w : 0.0f;
      pend.mo[0] = make_float4(a_nx * ddx, a_ny * ddy, a_ca, a_cb);
      pend.mo[1] = make_float4(a_cc, a_op, D[1][0], D[1][1]);
      pend.mo[2] = make_float4(D[1][2], D[1][3], 0.f, 0.f);
    }
    cur = nxt;
  }
  store_pending();
}

int launch_render_bwd_px(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, const GeomBuf& g,
                         const BinBuf& bb, const ImgBuf& im, const TraseRastGrads& gr, float* rows, uint8_t* row_flags,
                         size_t flag_bytes, const float* out_depth) {
  BwdPxArgs a;
  a.lineage = c.variant & (TRASE_VARIANT_FEATS_BG | TRASE_VARIANT_DEPTH_NORM); a.feat_bg = s.feat_bg; a.out_depth = out_depth;
  if ((a.lineage & TRASE_VARIANT_DEPTH_NORM) && gr.dL_ddepth && !out_depth) {
    set_error("render_bwd: the normalised-depth switch with a depth cotangent needs the forward's depth map (outputs.depth)");
    return TRASE_ERR_INVALID;
  }
  a.ranges = bb.ranges; a.point_list = bb.point_list; a.bg = s.bg;
  a.d_img = gr.dL_dimage; a.d_feat = gr.dL_dfeats; a.d_depth = gr.dL_ddepth;
  a.final_T = im.final_T; a.n_contrib = im.n_contrib; a.pair_slot = bb.pair_slot; a.hdr = g.hdr; a.geo = g.geo;
  a.ftab = g.ftab; a.rows = rows; a.row_flags = row_flags;
  a.W = s.image_width; a.H = s.image_height;
  a.gx8 = (a.W + SUB - 1) / SUB;
  { int lo, hi; strip_subtile_rows(s, lo, hi); a.tile0 = lo * a.gx8; a.ntiles = (hi - lo) * a.gx8; }
  (void)in;
  TRASE_CHECK(hipMemsetAsync(row_flags, 0, flag_bytes, c.stream));
  if (a.ntiles <= 0) return TRASE_OK;
  {
    ProfScope ps("render_bwd", c.stream);
    const dim3 grid((a.ntiles + PX_WPB - 1) / PX_WPB), block(PX_WPB * WAVE);
    hipLaunchKernelGGL(render_bwd_px_kernel, grid, block, 0, c.stream, a);
  }
  TRASE_POST_LAUNCH("render_bwd", c.stream, c.debug);
  return TRASE_OK;
}

}  // namespace trase
