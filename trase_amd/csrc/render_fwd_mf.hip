// render_fwd_mf.hip -- forward compositing with the 36-channel accumulation on the matrix cores (default for F = 32;
// render.hip keeps the packed-FP32 formulation for F = 0 / 16 and as the A/B reference, variant bit 0x2000).
//
// One wave per 8x8 sub-tile, the list walked front to back in chunks of 32 entries.  The blend
//     out[p][c] = sum_g w[p][g] * chan[g][c],     w = alpha * T,   36 channels (32 features, r, g, b, depth)
// is a GEMM with K = list entries.  Lane l = (m = l & 31, h = l >> 5): the lanes m and m + 32 share the pixel m of a
// 32-pixel block and split every K-step of 16 entries between them (h = 0: entries 0..7, h = 1: entries 8..15) --
// exactly the layout of an MFMA A fragment (row m, k = 8h..8h+7), so the weights go from the registers that produced
// them into v_mfma_f32_32x32x16_bf16 without any cross-lane movement (fp32 split in two bf16 values, three products
// per term: ~1e-5 relative, deterministic).  What is sequential along the list stays in the lane: each lane keeps the
// running product P_u = prod_{v<=u} (1 - alpha_v) of its eight entries, the two halves exchange their totals with one
// v_permlane32_swap, and w_u = (P_{u-1} - P_u) * T_in.  The stop rule (T (1 - alpha) < 1e-4: the pixel is finished and
// this entry is NOT blended) is a monotone threshold on those products; only a step in which some pixel of the wave
// crosses it takes the masked path.
//   * chan[g][c]: per chunk the 32 feature rows are fetched with four wave-wide 16-byte loads (eight lanes per row:
//     8 cache lines per instruction instead of one per lane), split to bf16 hi / lo and written TRANSPOSED into a
//     [channel][entry] tile in wave-private LDS, so that a B fragment (eight consecutive entries of one channel) is one
//     ds_read_b128.  r, g, b, depth are channels 32..35 of the same tile.
//   * the background term T_final * bg is added in fp32 in the epilogue (final_T reaches the accumulator layout
//     through LDS): an empty pixel shows the background exactly.
// Semantics: SURVEY.md Appendix A "Render fwd"; gates evaluated on the same exponent polynomial as every backward
// kernel (common.h pair_poly / poly_eval), so forward and backward agree bit for bit on which entries a pixel blends.
#include "common.h"

namespace trase {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int FM_WPB = 2;     // waves (sub-tiles) per workgroup
constexpr int FM_G = 32;      // list entries per chunk
constexpr int FM_LD = 40;     // tile row pitch in bf16 (32 entries + 8 pad: conflict-free fragment reads)
constexpr int FM_CH = 40;     // tile rows: 32 features, r g b depth, 4 zero rows

struct FmWaveLds {
  __bf16 hi[FM_CH * FM_LD];
  __bf16 lo[FM_CH * FM_LD];
  float4 k0[FM_G];            // k0, kj, ki, kjj
  float4 k1[FM_G];            // kii, kij, thr, -
};

struct FwdMfArgs {
  const uint2* ranges; const uint32_t* point_list;
  const float2* xy; const float4* conic_o; const float4* rgbd; const float* feats; const float* bg;
  const uint32_t* pair_slot; const uint32_t* pair_gauss; uint32_t* point_list_w; uint32_t cap;
  float* out_img; float* out_feat; float* out_depth; float* final_T; uint32_t* n_contrib;
  int W, H, gx8, ntiles;
  int tile0;             // first sub-tile of the strip being rendered (ntiles counts the strip's sub-tiles)
};

__device__ __forceinline__ void wave_lds_order() {     // wave-private LDS: the queue is in order, only the compiler must not reorder
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (a, b) -> packed bf16 high parts and packed bf16 residuals
__device__ __forceinline__ void split_pk(float a, float b, unsigned& hi, unsigned& lo) {
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(ra), "v"(rb));
}

#ifndef FM_OCC
#define FM_OCC 3
#endif
__global__ __launch_bounds__(FM_WPB* WAVE) __attribute__((amdgpu_waves_per_eu(FM_OCC, FM_OCC)))
void render_fwd_mf_kernel(FwdMfArgs a) {
  constexpr int F = 32;
  __shared__ __attribute__((aligned(16))) FmWaveLds s_w[FM_WPB];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  const int local = xcd_block(blockIdx.x, gridDim.x) * FM_WPB + wave;
  if (local >= a.ntiles) return;
  const int tile = a.tile0 + local;
  FmWaveLds& L = s_w[wave];
  const int tx = tile % a.gx8, ty = tile / a.gx8;
  const uint2 range = a.ranges[tile];
  const float bx = (float)(tx * SUB), by = (float)(ty * SUB);
  // this lane's two pixels: block mb covers pixel rows 4mb .. 4mb+3; pixel (row 4mb + (m >> 3), column m & 7)
  const float fj = (float)(m & 7);
  float fi[2], fii[2];
  bool inside[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int i = 4 * mb + (m >> 3);
    fi[mb] = (float)i; fii[mb] = (float)(i * i);
    inside[mb] = (tx * SUB + (m & 7)) < a.W && (ty * SUB + i) < a.H;
  }
  // zero the tile once: rows 36..39 and the pad columns are never written again
  {
    uint4* z = reinterpret_cast<uint4*>(L.hi);
    for (int o = lane; o < (int)(2 * FM_CH * FM_LD * sizeof(__bf16) / 16); o += WAVE) z[o] = make_uint4(0u, 0u, 0u, 0u);
  }
  f32x16 D[2][2];                                        // D[pixel block][channel block]: rows = pixels, columns = channels
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) D[mb][nb][r] = 0.f;
  // per pixel (identical in the two lanes that share it): Tin = transmittance while the pixel is live, 0 once it is
  // finished; Tc = the value final_T reports; lastc = n_contrib
  float Tin[2], Tc[2];
  uint32_t lastc[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) { Tin[mb] = inside[mb] ? 1.0f : 0.0f; Tc[mb] = 1.0f; lastc[mb] = 0; }
  // channel column of this lane in the B fragments: block 0 = feature m; block 1 = channel 32 + m, rows >= 40 do not
  // exist (zero): read the zero row 39 instead
  const int brow0 = m * FM_LD, brow1 = min(32 + m, FM_CH - 1) * FM_LD;
  wave_lds_order();

  for (uint32_t base = range.x; base < range.y; base += FM_G) {
    if (!__any(Tin[0] > 0.0f || Tin[1] > 0.0f)) break;
    const uint32_t n = min((uint32_t)FM_G, range.y - base);
    // ---- stage the chunk: lane e < n = list entry ---------------------------------------------------------
    uint32_t my_id = 0;
    if ((uint32_t)lane < n) {
      uint32_t id;
      if (a.point_list_w) {                              // the list still holds emit-order slots: translate, record
        const uint32_t slot = a.pair_slot[base + lane];
        id = a.pair_gauss[slot < a.cap ? slot : 0];
        a.point_list_w[base + lane] = id;
      } else {
        id = a.point_list[base + lane];
      }
      my_id = id;
    }
    // feature rows: instruction r fetches rows 8r .. 8r+7, eight lanes (16 bytes each) per row
    float4 fr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t rid = (uint32_t)__shfl((int)my_id, 8 * r + (lane >> 3));
      fr[r] = *reinterpret_cast<const float4*>(a.feats + (size_t)rid * F + 4 * (lane & 7));
    }
    if ((uint32_t)lane < n) {
      const PairPoly k = pair_poly(a.xy[my_id], a.conic_o[my_id], bx, by);
      L.k0[lane] = make_float4(k.k0, k.kj, k.ki, k.kjj);
      L.k1[lane] = make_float4(k.kii, k.kij, k.thr, 0.f);
      const float4 cd = a.rgbd[my_id];
      unsigned h01, l01, h23, l23;
      split_pk(cd.x, cd.y, h01, l01);
      split_pk(cd.z, cd.w, h23, l23);
      unsigned short* th = reinterpret_cast<unsigned short*>(L.hi) + 32 * FM_LD + lane;
      unsigned short* tl = reinterpret_cast<unsigned short*>(L.lo) + 32 * FM_LD + lane;
      th[0] = (unsigned short)h01; th[FM_LD] = (unsigned short)(h01 >> 16); th[2 * FM_LD] = (unsigned short)h23; th[3 * FM_LD] = (unsigned short)(h23 >> 16);
      tl[0] = (unsigned short)l01; tl[FM_LD] = (unsigned short)(l01 >> 16); tl[2 * FM_LD] = (unsigned short)l23; tl[3 * FM_LD] = (unsigned short)(l23 >> 16);
    } else if (lane < FM_G) {
      // past the end of the list: a closed gate (e <= -inf never holds); the tile keeps stale finite values, weight 0
      L.k0[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
      L.k1[lane] = make_float4(0.f, 0.f, -INFINITY, 0.f);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                        // transposed: tile[channel 4 (lane & 7) + e][entry 8r + (lane >> 3)]
      unsigned h01, l01, h23, l23;
      split_pk(fr[r].x, fr[r].y, h01, l01);
      split_pk(fr[r].z, fr[r].w, h23, l23);
      const int o = (4 * (lane & 7)) * FM_LD + 8 * r + (lane >> 3);
      unsigned short* th = reinterpret_cast<unsigned short*>(L.hi) + o;
      unsigned short* tl = reinterpret_cast<unsigned short*>(L.lo) + o;
      th[0] = (unsigned short)h01; th[FM_LD] = (unsigned short)(h01 >> 16); th[2 * FM_LD] = (unsigned short)h23; th[3 * FM_LD] = (unsigned short)(h23 >> 16);
      tl[0] = (unsigned short)l01; tl[FM_LD] = (unsigned short)(l01 >> 16); tl[2 * FM_LD] = (unsigned short)l23; tl[3 * FM_LD] = (unsigned short)(l23 >> 16);
    }
    wave_lds_order();
    // ---- two K-steps of 16 entries -------------------------------------------------------------------------
#ifndef FM_REP
#define FM_REP 1
#endif
#pragma unroll 1
    for (int t = 0; t < 2 * FM_REP; ++t) {
      if ((uint32_t)(16 * (t & 1)) >= n) continue;
      const int e0 = 16 * (t & 1) + 8 * h;               // this lane's eight entries
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        float Pu[8];
        float P = 1.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float4 q0 = L.k0[e0 + u];
          const float4 q1 = L.k1[e0 + u];
          PairPoly k;
          k.k0 = q0.x; k.kj = q0.y; k.ki = q0.z; k.kjj = q0.w; k.kii = q1.x; k.kij = q1.y; k.thr = q1.z;
          const float ex = poly_eval(k, poly_row_base(k, fi[mb], fii[mb]), poly_row_slope(k, fi[mb]), fj);
          const bool gate = (ex <= k.thr) && (ex >= LOG2_ALPHA_MIN);
          const float alpha = fminf(ALPHA_MAX, __builtin_amdgcn_exp2f(gate ? ex : -INFINITY));   // closed gate: 0
          P = fmaf(-alpha, P, P);
          Pu[u] = P;
        }
        // totals of the two halves: Q0 (entries 0..7), Q1 (entries 8..15)
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(P), __float_as_uint(P), false, false);
        const float Q0 = __uint_as_float(sw[0]), Q1 = __uint_as_float(sw[1]);
        const float tin = Tin[mb];
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
          uint32_t cnt = 0, lastb = 0;                   // live entries of this lane; 1 + index of its last blended one
          float cand = INFINITY;                         // transmittance after the last live entry of this lane
          float prev = 1.0f;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const bool live = Ts * Pu[u] >= T_STOP;
            if (crossing) {
              if (!live) w[u] = 0.f;
              else { cnt += 1; cand = Ts * Pu[u]; if (Pu[u] < prev) lastb = (uint32_t)(u + 1); }
            }
            prev = Pu[u];
          }
          const auto sc = __builtin_amdgcn_permlane32_swap(cnt, cnt, false, false);
          const auto sl = __builtin_amdgcn_permlane32_swap(lastb, lastb, false, false);
          const auto sd = __builtin_amdgcn_permlane32_swap(__float_as_uint(cand), __float_as_uint(cand), false, false);
          if (crossing) {
            // the half h = 1 only has live entries when the half h = 0 is live throughout (monotone)
            const uint32_t l0 = sl[0], l1 = sl[1];
            const uint32_t in_step = l1 ? 8u + l1 : l0;
            if (in_step) lastc[mb] = (base - range.x) + 16u * (t & 1) + in_step;
            Tc[mb] = fminf(tin, fminf(__uint_as_float(sd[0]), __uint_as_float(sd[1])));
            Tin[mb] = 0.0f;
            (void)sc;
          }
        }
        if (!crossing && tin > 0.0f) {
          Tin[mb] = Tout; Tc[mb] = Tout;
          // the list position of the last entry any gate let through is not tracked entry by entry: a pixel that never
          // finishes reports the whole list (entries behind its last blended one fail the same gates in the backward)
          lastc[mb] = (base - range.x) + min(n, (uint32_t)(16 * (t & 1) + 16));
        }
        // ---- MFMA: D[mb][nb] += w (32 pixels x 16 entries) * tile (16 entries x 32 channels) -----------------
        u32x4 ah, al;
#pragma unroll
        for (int v = 0; v < 4; ++v) { unsigned hh, ll; split_pk(w[2 * v], w[2 * v + 1], hh, ll); ah[v] = hh; al[v] = ll; }
        const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah), Al = __builtin_bit_cast(bf16x8, al);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int o = (nb == 0 ? brow0 : brow1) + e0;
          const bf16x8 Bh = *reinterpret_cast<const bf16x8*>(L.hi + o);
          const bf16x8 Bl = *reinterpret_cast<const bf16x8*>(L.lo + o);
          D[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, D[mb][nb], 0, 0, 0);
          D[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, D[mb][nb], 0, 0, 0);
          D[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, D[mb][nb], 0, 0, 0);
        }
      }
    }
    wave_lds_order();
  }

  // ---- background: final_T goes through LDS into the lane = channel layout of the accumulators -------------------
  float* const tbuf = reinterpret_cast<float*>(L.k0);    // 64 floats, pixel-major (the coefficient staging is dead now)
  wave_lds_order();
  if (h == 0) { tbuf[m] = Tc[0]; tbuf[32 + m] = Tc[1]; }
  wave_lds_order();
  const float bgc = m < 3 ? a.bg[m] : 0.0f;              // lanes 0..2 of channel block 1 hold r, g, b; depth gets no background

  // ---- outputs -----------------------------------------------------------------------------------------------
  const size_t hw = (size_t)a.H * a.W;
  const int x0 = tx * SUB, y0 = ty * SUB;
  if (h == 0) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      if (inside[mb]) {
        const size_t pix = (size_t)(y0 + 4 * mb + (m >> 3)) * a.W + x0 + (m & 7);
        a.final_T[pix] = Tc[mb];
        a.n_contrib[pix] = lastc[mb];
      }
    }
  }
  // D[mb][nb]: lane (c = m, h), register 4q + r = pixel row 4mb + q, column 4h + r of channel nb*32 + c
  const bool full = (x0 + SUB <= a.W) && ((a.W & 3) == 0);
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int y = y0 + 4 * mb + q;
      if (y >= a.H) continue;
      const size_t rowo = (size_t)y * a.W + x0 + 4 * h;
      float* pf = a.out_feat + (size_t)m * hw + rowo;
      float* pc = nullptr;                               // channel block 1: r g b (image planes) and depth
      if (m < 3) pc = a.out_img + (size_t)m * hw + rowo;
      else if (m == 3) pc = a.out_depth + rowo;
      const float4 vf = make_float4(D[mb][0][4 * q], D[mb][0][4 * q + 1], D[mb][0][4 * q + 2], D[mb][0][4 * q + 3]);
      const float4 tq = *reinterpret_cast<const float4*>(tbuf + 32 * mb + 8 * q + 4 * h);   // final_T of this register quad's pixels
      const float4 vc = make_float4(fmaf(tq.x, bgc, D[mb][1][4 * q]), fmaf(tq.y, bgc, D[mb][1][4 * q + 1]),
                                    fmaf(tq.z, bgc, D[mb][1][4 * q + 2]), fmaf(tq.w, bgc, D[mb][1][4 * q + 3]));
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
}

int launch_render_fwd_mf(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, const TraseRastOutputs& out,
                         const GeomBuf& g, const BinBuf& b, const ImgBuf& im, const uint32_t* pair_gauss, uint32_t cap) {
  FwdMfArgs a;
  a.ranges = b.ranges; a.point_list = b.point_list; a.xy = g.xy; a.conic_o = g.conic_o; a.rgbd = g.rgbd;
  a.feats = in.sh_objs; a.bg = s.bg;
  a.pair_slot = nullptr; a.pair_gauss = nullptr; a.point_list_w = nullptr; a.cap = 0;
  if (pair_gauss) { a.pair_slot = b.pair_slot; a.pair_gauss = pair_gauss; a.point_list_w = b.point_list; a.cap = cap; }
  a.out_img = out.image; a.out_feat = out.feats; a.out_depth = out.depth; a.final_T = im.final_T; a.n_contrib = im.n_contrib;
  a.W = s.image_width; a.H = s.image_height;
  a.gx8 = (a.W + SUB - 1) / SUB;
  { int lo, hi; strip_subtile_rows(s, lo, hi); a.tile0 = lo * a.gx8; a.ntiles = (hi - lo) * a.gx8; }
  if (a.ntiles <= 0) return TRASE_OK;                    // an empty strip
  {
    ProfScope ps("render_fwd", c.stream);
    hipLaunchKernelGGL(render_fwd_mf_kernel, dim3((a.ntiles + FM_WPB - 1) / FM_WPB), dim3(FM_WPB * WAVE), 0, c.stream, a);
  }
  TRASE_POST_LAUNCH("render_fwd", c.stream, c.debug);
  return TRASE_OK;
}

}  // namespace trase
