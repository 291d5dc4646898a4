// render_fwd_mf.hip -- forward compositing with the 36-channel accumulation on the matrix cores (default for F = 32;
// render.hip keeps the packed-FP32 formulation for F = 0 / 16 and as the cross-check, TRASE_VARIANT_VALU_FORWARD).
//
// One WORKGROUP of two waves per 8x8 sub-tile: wave w composites the 32 pixels of rows 4w .. 4w+3, both share one
// staged copy of the list chunk.  The list is walked front to back in chunks of 64 entries.  The blend
//     out[p][c] = sum_g w[p][g] * chan[g][c],     w = alpha * T,   36 channels (32 features, r, g, b, depth)
// is a GEMM with K = list entries.  Lane l = (m = l & 31, h = l >> 5): the lanes m and m + 32 share the pixel m of the
// wave's 32-pixel block and split every K-step of 16 entries between them (h = 0: entries 0..7, h = 1: entries 8..15)
// -- exactly the layout of an MFMA A fragment (row m, k = 8h..8h+7), so the weights go from the registers that produced
// them into v_mfma_f32_32x32x16_bf16 without any cross-lane movement (fp32 split in two bf16 values, three products
// per term: ~1e-5 relative, deterministic).  What is sequential along the list stays in the lane: each lane keeps the
// running product P_u = prod_{v<=u} (1 - alpha_v) of its eight entries, the two halves exchange their totals with one
// v_permlane32_swap, and w_u = (P_{u-1} - P_u) * T_in.  The stop rule (T (1 - alpha) < 1e-4: the pixel is finished and
// this entry is NOT blended) is a monotone threshold on those products; only a step in which some pixel of the wave
// crosses it takes the masked path.
//   * why two waves per sub-tile: the accumulators of 64 pixels x 64 channel columns are 64 registers; with 32 pixels
//     per wave the kernel needs ~100 VGPRs instead of ~150 and twice as many waves are resident to cover the dependent
//     load chain slot -> id -> geometry / feature rows at every chunk start (PMC: the one-wave version spent 52 % of its
//     wave cycles parked at s_waitcnt with 2.4 waves per SIMD).  The staging work is shared, not duplicated.
//   * chan[g][c]: per chunk the 64 feature rows are fetched with wave-wide 16-byte loads (eight lanes per row: 8 cache
//     lines per instruction instead of one per lane), split to bf16 hi / lo and stored AS ROWS ([entry][channel], one
//     8-byte store per lane and half); a B fragment (eight consecutive entries of one channel) is gathered by two
//     transposing LDS reads (ds_read_b64_tr_b16: within a 16-lane group lane 4 jj + cc hands in the address of channel
//     chunk cc of entry jj and lane c receives channel c of four entries).  Round 2 wrote the tile transposed with 32
//     two-byte stores per lane and chunk: as many LDS bank-conflict cycles as LDS instruction cycles (PMC).
//     r, g, b, depth are channels 32..35 of the same tile.
//   * the DEPTH map is the one output whose magnitude is not O(1) (view depth up to zfar = 100, scene/cameras.py:70): the
//     bf16-split product carries ~1e-5 RELATIVE, i.e. 5e-4 abs at z = 50.  Depth is therefore accumulated by each lane in
//     fp32 from the weights it already holds (8 FMAs per K-step; the two lanes of a pixel add their halves in the epilogue)
//     and holds the 1e-4 abs bar at any depth; channel 35 of the GEMM is left in place, unused.
//   * the background term T_final * bg is added in fp32 in the epilogue (final_T reaches the accumulator layout
//     through LDS): an empty pixel shows the background exactly.
// Semantics: SURVEY.md Appendix A "Render fwd"; gates evaluated on the same exponent polynomial as every backward
// kernel (common.h pair_poly / poly_eval), so forward and backward agree bit for bit on which entries a pixel blends.
#include "common.h"

namespace trase {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int FM_WAVES = 2;       // waves per sub-tile (= workgroup)
#ifndef FM_G_
#define FM_G_ 64
#endif
constexpr int FM_G = FM_G_;       // list entries per chunk
constexpr int FM_P = 48;          // tile row pitch in bf16: 32 features, r g b depth, 12 zeros (96 B: the transposing reads
                                  // of a 16-lane group -- 4 rows x 4 eight-byte chunks -- then fall on 16 different bank pairs)

struct FmLds {
  __bf16 hi[FM_G * FM_P];         // [entry][channel]: staged with 8-byte row-contiguous stores, read back TRANSPOSED
  __bf16 lo[FM_G * FM_P];
  // exponent polynomials, one 64-byte record per PAIR of entries (a = 2p, b = 2p + 1):
  //   k0a k0b kja kjb | kia kib kjja kjjb | kiia kiib kija kijb | thra thrb - -
  // so that a K-step evaluates two entries per v_pk_fma_f32 without any register shuffling
  float kp[FM_G / 2][16];
  float zs[FM_G];                 // view depth of the chunk's entries: the depth map is accumulated in fp32 (see below)
  float tfin[2][32];              // final_T of the two pixel blocks (epilogue)
  int live[2];                    // "some pixel of wave w is still live"
};

struct FwdMfArgs {
  const uint2* ranges; const uint32_t* point_list;
  const float2* xy; const float4* conic_o; const float4* rgbd; const float* feats; const float* bg;
  const uint32_t* pair_slot; const uint32_t* pair_gauss; uint32_t* point_list_w; uint32_t cap;
  const uint32_t* hdr;                 // HDR_PACK: the list values carry the Gaussian id in their upper bits
  const float4* geo;                   // 64-byte geometry records (GeomBuf::geo)
  const uint32_t* ftab;                // feature rows as bf16 [hi 32 | lo 32] (GeomBuf::ftab)
  float* out_img; float* out_feat; float* out_depth; float* final_T; uint32_t* n_contrib;
  int W, H, gx8, ntiles;
  int tile0;             // first sub-tile of the strip being rendered (ntiles counts the strip's sub-tiles)
  int order_mode;        // 0: image order; 8 / 16: blocks of 8x8 / 16x16 sub-tiles (common.h blocked_tile)
  int lineage;           // variant bits TRASE_VARIANT_FEATS_BG / TRASE_VARIANT_DEPTH_NORM (0 = public lineage)
  float feat_bg;
};

typedef float f2v __attribute__((ext_vector_type(2)));
// d = a.lo * x + y  /  d = a.hi * x + y  on both halves of (x, y): v_pk_fma_f32 with a broadcast first operand
__device__ __forceinline__ f2v pk_fma_b0(f2v a, f2v x, f2v y) {
  f2v d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(x), "v"(y));
  return d;
}
__device__ __forceinline__ f2v pk_fma_b1(f2v a, f2v x, f2v y) {
  f2v d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(x), "v"(y));
  return d;
}

// LDS handed between the two waves of the workgroup: LDS-only barrier (no vmcnt drain: stores / loads stay in flight)
__device__ __forceinline__ void wg_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


// B fragment (eight consecutive entries of this lane's channel) from the [entry][channel] tile: two transposing reads
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 frag_tr(const __bf16* __restrict__ p) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * FM_P));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}

__global__ __launch_bounds__(FM_WAVES* WAVE) __attribute__((amdgpu_waves_per_eu(5, 5)))
void render_fwd_mf_kernel(FwdMfArgs a) {
  constexpr int F = 32;
  __shared__ __attribute__((aligned(16))) FmLds L;
  const int wv = threadIdx.x >> 6;                       // pixel block of this wave: rows 4 wv .. 4 wv + 3
  const int lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  const int local = xcd_block(blockIdx.x, gridDim.x);
  if (local >= a.ntiles) return;
  int tx, ty;
  if (a.order_mode == 0) { const int t0 = a.tile0 + local; tx = t0 % a.gx8; ty = t0 / a.gx8; }
  else {
    if (a.order_mode == 8) blocked_tile<8>(local, a.gx8, a.ntiles / a.gx8, tx, ty); else blocked_tile<16>(local, a.gx8, a.ntiles / a.gx8, tx, ty);
    ty += a.tile0 / a.gx8;
  }
  const int tile = ty * a.gx8 + tx;
  const uint2 range = a.ranges[tile];
  const float bx = (float)(tx * SUB), by = (float)(ty * SUB);
  // this lane's pixel: row 4 wv + (m >> 3), column m & 7
  const int pi = 4 * wv + (m >> 3), pj = m & 7;
  const float fj = (float)pj, fi = (float)pi, fii = (float)(pi * pi);
  const f2v FI2 = {fi, fii}, FJ2 = {fj, 0.f};            // broadcast operands of the packed polynomial evaluation
  const bool inside = (tx * SUB + pj) < a.W && (ty * SUB + pi) < a.H;
  // zero the tile once: channels 36..47 are never written again
  {
    uint4* z = reinterpret_cast<uint4*>(L.hi);
    for (int o = threadIdx.x; o < (int)(2 * FM_G * FM_P * sizeof(__bf16) / 16); o += FM_WAVES * WAVE) z[o] = make_uint4(0u, 0u, 0u, 0u);
  }
  f32x16 D[2];                                           // D[channel block]: rows = pixels, columns = channels
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) D[nb][r] = 0.f;
  // per pixel (identical in the two lanes that share it): Tin = transmittance while the pixel is live, 0 once it is
  // finished; Tc = the value final_T reports; lastc = n_contrib
  float Tin = inside ? 1.0f : 0.0f, Tc = 1.0f;
  uint32_t lastc = 0;
  float dacc = 0.f;                                      // fp32 depth of this lane's half of every K-step
  // this lane's chunk address in the transposing B-fragment reads: entry (lane & 15) >> 2 of the four, channel chunk
  // 16 * ((lane >> 4) & 1) + 4 * (lane & 3) of the block (block 1 = channels 32..47: r g b depth and zeros)
  const int btr = (8 * h + ((lane & 15) >> 2)) * FM_P + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  if (lane == 0) L.live[wv] = 1;
  wg_lds_barrier();

  const uint32_t jb = a.point_list_w ? a.hdr[HDR_PACK] : 0u;
  // the list values of a chunk are requested one chunk ahead (one register): the staging step sits on the critical path of
  // the workgroup -- both waves wait at its barrier -- and this takes one of its two dependent load levels off it
  const uint32_t* const lv_src = a.point_list_w ? a.pair_slot : a.point_list;
  uint32_t lv_next = (range.x + (uint32_t)lane < range.y) ? lv_src[range.x + lane] : 0u;
  for (uint32_t base = range.x; base < range.y; base += FM_G) {
    if (!(L.live[0] | L.live[1])) break;                 // workgroup-uniform: both waves read the same flags
    const uint32_t n = min((uint32_t)FM_G, range.y - base);
    // ---- stage the chunk: lane e = list entry (both waves fetch the ids; each does half of the rest) --------------
    uint32_t my_id = 0;
    const uint32_t lv = lv_next;
    lv_next = (base + FM_G + (uint32_t)lane < range.y) ? lv_src[base + FM_G + lane] : 0u;
    if ((uint32_t)lane < n) {
      if (a.point_list_w) {                              // the list still holds emit-order slots: translate, record
        const uint32_t slot = lv;
        if (jb) {                                        // packed value: the id is a shift away, nothing to record
          my_id = slot >> jb;
        } else {
          my_id = a.pair_gauss[slot < a.cap ? slot : 0];
          if (wv == 0) a.point_list_w[base + lane] = my_id;
        }
      } else {
        my_id = lv;
      }
    }
    // feature rows: 16 rows per instruction and workgroup (eight lanes, 16 bytes each, per row); wave wv takes rows
    // 8wv .. 8wv+7 of each 16
    // (rows of the bf16 table GeomBuf::ftab, [hi 32 | lo 32]: lanes 0..3 of a row fetch the hi parts of channels
    // 8 (lane & 3) .. + 7, lanes 4..7 the lo parts)
    uint4 fr[FM_G / 16];
#pragma unroll
    for (int r = 0; r < FM_G / 16; ++r) {
      const uint32_t rid = (uint32_t)__shfl((int)my_id, 16 * r + 8 * wv + (lane >> 3));
      // past the end of the list: zeros (weight 0 times a stale or never-written row could be 0 * NaN)
      fr[r] = make_uint4(0u, 0u, 0u, 0u);
      if ((uint32_t)(16 * r + 8 * wv + (lane >> 3)) < n) fr[r] = reinterpret_cast<const uint4*>(a.ftab)[(size_t)rid * 8 + (lane & 7)];
    }
    if (wv == 0) {                                       // exponent polynomials
      if ((uint32_t)lane < n) {
        const float4 gq = a.geo[4 * (size_t)my_id];
        const PairPoly k = pair_poly(make_float2(gq.x, gq.y), a.geo[4 * (size_t)my_id + 1], bx, by);
        float* const rec = &L.kp[lane >> 1][lane & 1];
        rec[0] = k.k0; rec[2] = k.kj; rec[4] = k.ki; rec[6] = k.kjj; rec[8] = k.kii; rec[10] = k.kij; rec[12] = k.thr;
      } else {
        // past the end of the list: a closed gate (e <= -inf never holds); the tile keeps stale finite values, weight 0
        float* const rec = &L.kp[lane >> 1][lane & 1];
        rec[0] = 0.f; rec[2] = 0.f; rec[4] = 0.f; rec[6] = 0.f; rec[8] = 0.f; rec[10] = 0.f; rec[12] = -INFINITY;
      }
    } else {                                             // r g b depth -> tile rows 32..35; depth also as fp32
      float z = 0.f;                                     // past the end of the list: 0 (its weight is 0 as well)
      if ((uint32_t)lane < n) {
        z = reinterpret_cast<const float*>(a.geo)[16 * (size_t)my_id + 11];
        const float4 cs = a.geo[4 * (size_t)my_id + 3];  // the colour + depth already split (preprocess)
        *reinterpret_cast<uint2*>(L.hi + lane * FM_P + 32) = make_uint2(__float_as_uint(cs.x), __float_as_uint(cs.y));
        *reinterpret_cast<uint2*>(L.lo + lane * FM_P + 32) = make_uint2(__float_as_uint(cs.z), __float_as_uint(cs.w));
      }
      L.zs[lane] = z;
    }
#pragma unroll
    for (int r = 0; r < FM_G / 16; ++r) {                // tile[entry 16r + 8wv + (lane >> 3)][channels 8 (lane & 3) .. + 7]
      const int o = (16 * r + 8 * wv + (lane >> 3)) * FM_P + 8 * (lane & 3);
      *reinterpret_cast<uint4*>(((lane & 4) ? L.lo : L.hi) + o) = fr[r];
    }
    wg_lds_barrier();
    // ---- K-steps of 16 entries ---------------------------------------------------------------------------------
    if (__any(Tin > 0.0f)) {
#pragma unroll 1
      for (int t = 0; t < FM_G / 16; ++t) {
        if ((uint32_t)(16 * t) >= n) break;
        const int e0 = 16 * t + 8 * h;                   // this lane's eight entries
        float Pu[8];
        float P = 1.0f;
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          // two entries per packed FMA; per component the same fmaf tree as poly_row_base / poly_row_slope / poly_eval
          // (common.h), so the gates agree bit for bit with the backward's
          const f4v* const rec = reinterpret_cast<const f4v*>(&L.kp[(e0 + u) >> 1][0]);
          const f4v r0 = rec[0], r1 = rec[1], r2 = rec[2];
          const f2v thr2 = *reinterpret_cast<const f2v*>(&rec[3]);
          f2v bs = pk_fma_b0(FI2, __builtin_shufflevector(r1, r1, 0, 1), __builtin_shufflevector(r0, r0, 0, 1));   // i ki + k0
          bs = pk_fma_b1(FI2, __builtin_shufflevector(r2, r2, 0, 1), bs);                                          // + ii kii
          const f2v sl = pk_fma_b0(FI2, __builtin_shufflevector(r2, r2, 2, 3), __builtin_shufflevector(r0, r0, 2, 3));   // i kij + kj
          const f2v in = pk_fma_b0(FJ2, __builtin_shufflevector(r1, r1, 2, 3), sl);                                 // j kjj + slope
          const f2v ex2 = pk_fma_b0(FJ2, in, bs);                                                                  // j (.) + base
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            const float ex = ex2[v];
            const bool gate = (ex <= thr2[v]) && (ex >= LOG2_ALPHA_MIN);
            const float alpha = fminf(ALPHA_MAX, __builtin_amdgcn_exp2f(gate ? ex : -INFINITY));   // closed gate: 0
            P = fmaf(-alpha, P, P);
            Pu[u + v] = P;
          }
        }
        // totals of the two halves: Q0 (entries 0..7), Q1 (entries 8..15)
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(P), __float_as_uint(P), false, false);
        const float Q0 = __uint_as_float(sw[0]), Q1 = __uint_as_float(sw[1]);
        const float tin = Tin;
        const float Ts = h ? tin * Q0 : tin;             // transmittance in front of this lane's first entry
        const float Tout = (tin * Q0) * Q1;
        float w[8];
        w[0] = (1.0f - Pu[0]) * Ts;
#pragma unroll
        for (int u = 1; u < 8; ++u) w[u] = (Pu[u - 1] - Pu[u]) * Ts;
        const bool crossing = tin > 0.0f && Tout < T_STOP;
        if (__any(crossing)) {
          // some pixel finishes inside this step: entries from the first one with T (1 - alpha) < T_STOP on are not
          // blended.  The products are monotone, so "still live at entry u" is a threshold on Ts * Pu[u].
          uint32_t lastb = 0;                            // 1 + index of this lane's last blended entry
          float cand = INFINITY;                         // transmittance after the last live entry of this lane
          // (selects, not branches: in a dense scene most pixels finish, each in its own step, so this block runs in a
          // large share of the K-steps; the per-entry exec-masked regions the compiler made of the nested ifs were ~250
          // instructions with 96 register-pair moves)
          float prev = 1.0f;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float tu = Ts * Pu[u];
            const bool live = crossing && tu >= T_STOP;
            const bool dead = crossing && !(tu >= T_STOP);
            w[u] = dead ? 0.f : w[u];
            cand = live ? tu : cand;
            lastb = (live && Pu[u] < prev) ? (uint32_t)(u + 1) : lastb;
            prev = Pu[u];
          }
          const auto sl = __builtin_amdgcn_permlane32_swap(lastb, lastb, false, false);
          const auto sd = __builtin_amdgcn_permlane32_swap(__float_as_uint(cand), __float_as_uint(cand), false, false);
          if (crossing) {
            // the half h = 1 only has live entries when the half h = 0 is live throughout (monotone)
            const uint32_t l0 = sl[0], l1 = sl[1];
            const uint32_t in_step = l1 ? 8u + l1 : l0;
            if (in_step) lastc = (base - range.x) + 16u * t + in_step;
            Tc = fminf(tin, fminf(__uint_as_float(sd[0]), __uint_as_float(sd[1])));
            Tin = 0.0f;
          }
        }
        if (!crossing && tin > 0.0f) {
          Tin = Tout; Tc = Tout;
          // the list position of the last entry any gate let through is not tracked entry by entry: a pixel that never
          // finishes reports the whole list (entries behind its last blended one fail the same gates in the backward)
          lastc = (base - range.x) + min(n, (uint32_t)(16 * t + 16));
        }
        const float4 z0 = *reinterpret_cast<const float4*>(&L.zs[e0]), z1 = *reinterpret_cast<const float4*>(&L.zs[e0 + 4]);
        dacc = fmaf(w[0], z0.x, dacc); dacc = fmaf(w[1], z0.y, dacc); dacc = fmaf(w[2], z0.z, dacc); dacc = fmaf(w[3], z0.w, dacc);
        dacc = fmaf(w[4], z1.x, dacc); dacc = fmaf(w[5], z1.y, dacc); dacc = fmaf(w[6], z1.z, dacc); dacc = fmaf(w[7], z1.w, dacc);
        // ---- MFMA: D[nb] += w (32 pixels x 16 entries) * tile (16 entries x 32 channels) ------------------------
        u32x4 ah, al;
#pragma unroll
        for (int v = 0; v < 4; ++v) { unsigned hh, ll; split_pk(w[2 * v], w[2 * v + 1], hh, ll); ah[v] = hh; al[v] = ll; }
        const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah), Al = __builtin_bit_cast(bf16x8, al);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int o = btr + 16 * t * FM_P + 32 * nb;
          const bf16x8 Bh = frag_tr(L.hi + o), Bl = frag_tr(L.lo + o);
          D[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, D[nb], 0, 0, 0);
          D[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, D[nb], 0, 0, 0);
          D[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, D[nb], 0, 0, 0);
        }
      }
    }
    {
      const bool any_live = __any(Tin > 0.0f);
      if (lane == 0) L.live[wv] = any_live ? 1 : 0;
    }
    wg_lds_barrier();                                    // the tile may be overwritten; both waves see both flags
  }

  // ---- outputs -----------------------------------------------------------------------------------------------
  const size_t hw = (size_t)a.H * a.W;
  const int x0 = tx * SUB, y0 = ty * SUB;
  {
    const auto sd = __builtin_amdgcn_permlane32_swap(__float_as_uint(dacc), __float_as_uint(dacc), false, false);
    dacc = __uint_as_float(sd[0]) + __uint_as_float(sd[1]);
  }
  if (h == 0) {
    if (inside) {
      const size_t pix = (size_t)(y0 + pi) * a.W + x0 + pj;
      if (!(a.lineage & TRASE_VARIANT_FORWARD_ONLY)) {   // (what only the backward reads: skipped under no_grad)
        a.final_T[pix] = Tc;
        a.n_contrib[pix] = lastc;
      }
      if (a.lineage & TRASE_VARIANT_DEPTH_NORM) {        // lineage switch: depth / accumulated alpha
        const float A = 1.0f - Tc;
        dacc = A > 1e-10f ? dacc / A : 0.0f;
      }
      a.out_depth[pix] = dacc;
    }
    L.tfin[wv][m] = Tc;                                  // final_T in pixel order: the accumulator layout reads it back
  }
  const float bgc = m < 3 ? a.bg[m] : 0.0f;              // lanes 0..2 of channel block 1 hold r, g, b; depth gets no background
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // tfin is written and read by the same wave (in-order LDS queue)
  // D[nb]: lane (c = m, h), register 4q + r = pixel row 4 wv + q, column 4h + r of channel nb*32 + c
  const bool full = (x0 + SUB <= a.W) && ((a.W & 3) == 0);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int y = y0 + 4 * wv + q;
    if (y >= a.H) continue;
    const size_t rowo = (size_t)y * a.W + x0 + 4 * h;
    float* pf = a.out_feat + (size_t)m * hw + rowo;
    float* pc = nullptr;                                 // channel block 1: r g b (image planes) and depth
    if (m < 3) pc = a.out_img + (size_t)m * hw + rowo;   // (depth: written above from the fp32 accumulation)
    float4 vf = make_float4(D[0][4 * q], D[0][4 * q + 1], D[0][4 * q + 2], D[0][4 * q + 3]);
    const float4 tq = *reinterpret_cast<const float4*>(&L.tfin[wv][8 * q + 4 * h]);   // final_T of this register quad's pixels
    if (a.lineage & TRASE_VARIANT_FEATS_BG)              // lineage switch: features over a background value
      vf = make_float4(fmaf(tq.x, a.feat_bg, vf.x), fmaf(tq.y, a.feat_bg, vf.y), fmaf(tq.z, a.feat_bg, vf.z), fmaf(tq.w, a.feat_bg, vf.w));
    const float4 vc = make_float4(fmaf(tq.x, bgc, D[1][4 * q]), fmaf(tq.y, bgc, D[1][4 * q + 1]),
                                  fmaf(tq.z, bgc, D[1][4 * q + 2]), fmaf(tq.w, bgc, D[1][4 * q + 3]));
    if (full) {
      *reinterpret_cast<float4*>(pf) = vf;
      if (pc) *reinterpret_cast<float4*>(pc) = vc;
    } else {
      const float f4[4] = {vf.x, vf.y, vf.z, vf.w}, c4[4] = {vc.x, vc.y, vc.z, vc.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (x0 + 4 * h + r < a.W) {
          pf[r] = f4[r];
          if (pc) pc[r] = c4[r];
        }
      }
    }
  }
}

int launch_render_fwd_mf(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, const TraseRastOutputs& out,
                         const GeomBuf& g, const BinBuf& b, const ImgBuf& im, const uint32_t* pair_gauss, uint32_t cap) {
  FwdMfArgs a;
  a.ranges = b.ranges; a.point_list = b.point_list; a.xy = g.xy; a.conic_o = g.conic_o; a.rgbd = g.rgbd;
  a.feats = in.sh_objs; a.bg = s.bg;
  a.pair_slot = nullptr; a.pair_gauss = nullptr; a.point_list_w = nullptr; a.cap = 0;
  if (pair_gauss) { a.pair_slot = b.pair_slot; a.pair_gauss = pair_gauss; a.point_list_w = b.point_list; a.cap = cap; }
  a.hdr = g.hdr; a.geo = g.geo; a.ftab = g.ftab;
  a.out_img = out.image; a.out_feat = out.feats; a.out_depth = out.depth; a.final_T = im.final_T; a.n_contrib = im.n_contrib;
  a.W = s.image_width; a.H = s.image_height;
  a.lineage = c.variant & (TRASE_VARIANT_FEATS_BG | TRASE_VARIANT_DEPTH_NORM | TRASE_VARIANT_FORWARD_ONLY); a.feat_bg = s.feat_bg;
  a.order_mode = 16;
#ifdef TRASE_AB
  a.order_mode = (c.variant & TRASE_VARIANT_AB_ORDER_IMAGE) ? 0 : ((c.variant & TRASE_VARIANT_AB_ORDER_8) ? 8 : 16);
#endif
  a.gx8 = (a.W + SUB - 1) / SUB;
  { int lo, hi; strip_subtile_rows(s, lo, hi); a.tile0 = lo * a.gx8; a.ntiles = (hi - lo) * a.gx8; }
  if (a.ntiles <= 0) return TRASE_OK;                    // an empty strip
  {
    ProfScope ps("render_fwd", c.stream);
    hipLaunchKernelGGL(render_fwd_mf_kernel, dim3(a.ntiles), dim3(FM_WAVES * WAVE), 0, c.stream, a);
  }
  TRASE_POST_LAUNCH("render_fwd", c.stream, c.debug);
  return TRASE_OK;
}

}  // namespace trase
