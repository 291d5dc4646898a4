// render_bwd_gs.hip -- backward of the compositing stage, "lane = Gaussian" formulation (default).
//
// One wave per 8x8 sub-tile.  The sub-tile's depth-ordered list is walked back to front in chunks
// of 64 entries; lane l owns ONE Gaussian of the chunk (lane 0 = farthest) and keeps its
// parameters, its 32-float feature row and all of its gradient accumulators in registers.  The
// wave then visits the 64 pixels of the sub-tile one after the other (pixel data is wave-uniform:
// cotangents come from wave-private LDS as broadcast reads).  What is sequential along the list
// for a fixed pixel becomes a scan across lanes:
//     T_l  = T_end / prod_{k<=l} (1 - a_k)                      (DPP multiplicative scan)
//     U_l  = U_end + sum_{k<l} w_k s_k ,  w = a T               (DPP additive scan, exclusive)
//     dL/da_l = T_l s_l - U_l / (1 - a_l),   s_l = <channels of Gaussian l, pixel cotangent>
// (U_end starts as T_final * <bg, d_rgb>), so no per-value cross-lane reduction is ever needed:
// every per-Gaussian sum over pixels is a plain in-register accumulation.  Each chunk then writes
// ONE row per (sub-tile, Gaussian) pair into the pair's emit-order slot; rows of one Gaussian are
// contiguous there and are summed by reduce_rows_kernel (binning.hip).  No atomics anywhere.
// Semantics: SURVEY.md Appendix A "Render bwd" (lineage: straight-through 0.99 clamp, true
// derivative wrt the conic's off-diagonal entry).
#include "common.h"

namespace trase {

constexpr int GWPB = 4;   // waves (sub-tiles) per workgroup
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct BwdGsArgs {
  const uint2* ranges; const uint32_t* point_list;
  const float2* xy; const float4* conic_o; const float4* rgbd; const float* feats; const float* bg;
  const float* d_img; const float* d_feat; const float* d_depth;
  const float* final_T; const uint32_t* n_contrib;
  const uint32_t* pair_slot;   // emit-order slot of every list entry (or the packed value, HDR_PACK)
  const uint32_t* hdr; const float4* geo;
  float* rows;         // (capacity, F+12) one gradient row per pair, indexed by slot
  uint8_t* row_flags;  // (capacity) 1 where a row was written (zeroed by the caller beforehand)
  int W, H, gx8, ntiles;
  int tile0;             // first sub-tile of the strip being rendered (ntiles counts the strip's sub-tiles)
  int lineage;         // variant bits TRASE_VARIANT_FEATS_BG / TRASE_VARIANT_DEPTH_NORM (0 = public lineage)
  float feat_bg;
  const float* out_depth;
};

__device__ __forceinline__ void wave_lds_sync2() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int F, bool ASM>
__global__ __launch_bounds__(GWPB* WAVE) void render_bwd_gs_kernel(BwdGsArgs a) {
  constexpr int CH = 4 + F;                      // r g b depth | features
  __shared__ __attribute__((aligned(16))) float s_cot[GWPB][WAVE][CH];   // pixel-major cotangents
  __shared__ __attribute__((aligned(16))) float4 s_pix[GWPB][WAVE];      // T_end, U_end, last (bits), -
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int local = xcd_block(blockIdx.x, gridDim.x) * GWPB + wave;
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
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, gd = 0.f, Tf = 0.f;
    last = 0;
    if (inside) {
      Tf = a.final_T[pix];
      last = a.n_contrib[pix];
      if (a.d_img) { g0 = a.d_img[pix]; g1 = a.d_img[hw + pix]; g2 = a.d_img[2 * hw + pix]; }
      if (a.d_depth) gd = a.d_depth[pix];
    }
    float bextra = 0.f;                             // lineage switches: see render_bwd_hw.hip
    if ((a.lineage & TRASE_VARIANT_DEPTH_NORM) && a.d_depth && inside) {
      const float A = 1.0f - Tf;
      gd = A > 1e-10f ? gd / A : 0.0f;
      bextra = gd * a.out_depth[pix];
    }
    float fsum = 0.f;
    float* row = s_cot[wave][lane];
    *reinterpret_cast<float4*>(row) = make_float4(g0, g1, g2, gd);
    if (F > 0) {
#pragma unroll
      for (int c4 = 0; c4 < F / 4; ++c4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (inside && a.d_feat) {
          v.x = a.d_feat[(size_t)(4 * c4 + 0) * hw + pix];
          v.y = a.d_feat[(size_t)(4 * c4 + 1) * hw + pix];
          v.z = a.d_feat[(size_t)(4 * c4 + 2) * hw + pix];
          v.w = a.d_feat[(size_t)(4 * c4 + 3) * hw + pix];
        }
        *reinterpret_cast<float4*>(row + 4 + 4 * c4) = v;
        fsum += (v.x + v.y) + (v.z + v.w);
      }
    }
    if (a.lineage & TRASE_VARIANT_FEATS_BG) bextra = fmaf(a.feat_bg, fsum, bextra);
    const float bdot = a.bg[0] * g0 + a.bg[1] * g1 + a.bg[2] * g2 + bextra;
    s_pix[wave][lane] = make_float4(Tf, Tf * bdot, __uint_as_float(last), 0.f);
  }
  uint32_t wave_last = last;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, o));
  wave_lds_sync2();
  const float ddx = 0.5f * (float)a.W, ddy = 0.5f * (float)a.H;
  const float bx = (float)(tx * SUB), by = (float)(ty * SUB);
  // ---- chunks of 64 list entries, back to front ----------------------------------------------------
  // Entries behind the last one any pixel of this sub-tile blended are never touched: they get no
  // row (their flag stays 0).  Chunks are aligned to that last blended entry.
  const uint32_t jb = a.hdr[HDR_PACK];
  for (uint32_t c1 = wave_last; c1 > 0; c1 = (c1 > WAVE) ? c1 - WAVE : 0) {
    const uint32_t c0 = (c1 > WAVE) ? c1 - WAVE : 0;
    const uint32_t n = c1 - c0;
    const bool lane_valid = (uint32_t)lane < n;
    const uint32_t pos = lane_valid ? (c1 - 1 - lane) : 0;     // lane 0 = farthest entry of the chunk
    const uint32_t lv = a.pair_slot[range.x + pos];
    const uint32_t id = jb ? (lv >> jb) : a.point_list[range.x + pos];   // HDR_PACK: id and pair index in the list value
    const uint32_t slot = jb ? __float_as_uint(a.geo[4 * (size_t)id].z) + (lv & ((1u << jb) - 1u)) : lv;
    constexpr int ROW = F + 12;
    const float2 gxy = a.xy[id];
    const float4 co = a.conic_o[id];
    const float4 col = a.rgbd[id];
    f32x2 f2[F > 0 ? F / 2 : 1];
    if (F > 0) {
      const float4* fr = reinterpret_cast<const float4*>(a.feats + (size_t)id * F);
#pragma unroll
      for (int c4 = 0; c4 < F / 4; ++c4) {
        const float4 v = fr[c4];
        f2[2 * c4] = f32x2{v.x, v.y}; f2[2 * c4 + 1] = f32x2{v.z, v.w};
      }
    }
    const PairPoly k = pair_poly(gxy, co, bx, by);
    const uint32_t pos_cmp = lane_valid ? pos : 0xffffffffu;     // invalid lanes never pass pos < plast
    // pixel-moment sums of q = alpha_raw * dL/dalpha (geometry gradients are linear in them)
    float S0 = 0.f, Sj = 0.f, Si = 0.f, Sjj = 0.f, Sij = 0.f, Sii = 0.f;
    float a_r = 0.f, a_g = 0.f, a_b = 0.f, a_d = 0.f;
    f32x2 af2[F > 0 ? F / 2 : 1];
#pragma unroll
    for (int c = 0; c < F / 2; ++c) af2[c] = f32x2{0.f, 0.f};
    for (int i = 0; i < SUB; ++i) {
      const float fi = (float)i, fii = (float)(i * i);
      const float base = poly_row_base(k, fi, fii);
      const float slope = poly_row_slope(k, fi);
      float R0 = 0.f, R1 = 0.f, R2 = 0.f;            // row sums of q, q*j, q*j^2
#pragma unroll
      for (int j = 0; j < SUB; ++j) {
        const int p = i * SUB + j;
        const float4 pst = s_pix[wave][p];               // uniform read
        const uint32_t plast = __builtin_amdgcn_readfirstlane(__float_as_uint(pst.z));
        if (plast <= c0) continue;                       // nothing of this chunk was blended into pixel p
        const float T_end = pst.x, U_end = pst.y;
        const float e = poly_eval(k, base, slope, (float)j);
        const float araw = __builtin_amdgcn_exp2f(e);    // opacity * exp(power)
        const bool ok = (e <= k.thr) && (e >= LOG2_ALPHA_MIN) && (pos_cmp < plast);
        const float al = ok ? fminf(ALPHA_MAX, araw) : 0.0f;
        const float om = 1.0f - al;
        const float PP = ASM ? wave_scan_mul_asm(om) : wave_scan_mul(om);
        const float T = T_end * __builtin_amdgcn_rcpf(PP);   // transmittance in front of this Gaussian
        const float w = al * T;
        // s = <channels, cotangent of pixel p>
        const float* cot = s_cot[wave][p];
        const float4 cg = *reinterpret_cast<const float4*>(cot);
        float s = col.x * cg.x + col.y * cg.y + col.z * cg.z + col.w * cg.w;
        f32x2 gf2[F > 0 ? F / 2 : 1];
        if (F > 0) {
          f32x2 sacc = {0.f, 0.f};
#pragma unroll
          for (int c4 = 0; c4 < F / 4; ++c4) {
            const float4 v = *reinterpret_cast<const float4*>(cot + 4 + 4 * c4);
            gf2[2 * c4] = f32x2{v.x, v.y}; gf2[2 * c4 + 1] = f32x2{v.z, v.w};
            sacc = f2[2 * c4] * gf2[2 * c4] + sacc;           // v_pk_fma_f32
            sacc = f2[2 * c4 + 1] * gf2[2 * c4 + 1] + sacc;
          }
          s += sacc.x + sacc.y;
        }
        const float ws = w * s;
        const float incl = ASM ? wave_scan_add_asm(ws) : wave_scan_add(ws);
        const float U = U_end + (incl - ws);
        const float dL_dalpha = ok ? (T * s - U * __builtin_amdgcn_rcpf(om)) : 0.0f;
        // carries for the next (nearer) chunk: lane 63 sees the whole chunk
        if (lane == WAVE - 1) {
          s_pix[wave][p].x = T;                          // om == 1 on invalid lanes: T in front of the chunk
          s_pix[wave][p].y = U_end + incl;
        }
        const float q = araw * dL_dalpha;                // == opacity * G * dL/dalpha (straight-through clamp)
        R0 += q;
        R1 = fmaf(q, (float)j, R1);
        R2 = fmaf(q, (float)(j * j), R2);
        a_r = fmaf(w, cg.x, a_r); a_g = fmaf(w, cg.y, a_g); a_b = fmaf(w, cg.z, a_b); a_d = fmaf(w, cg.w, a_d);
        if (F > 0) {
          const f32x2 w2 = {w, w};
#pragma unroll
          for (int c = 0; c < F / 2; ++c) af2[c] = w2 * gf2[c] + af2[c];   // v_pk_fma_f32
        }
      }
      S0 += R0; Sj += R1; Sjj += R2;
      Si = fmaf(fi, R0, Si); Sii = fmaf(fii, R0, Sii); Sij = fmaf(fi, R1, Sij);
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
    // ---- write this chunk's per-Gaussian sums: one row per pair --------------------------------
    if (lane_valid) {
      float4* row = reinterpret_cast<float4*>(a.rows + (size_t)slot * bwd_row_stride(F));
      if (F > 0) {
#pragma unroll
        for (int q = 0; q < F / 4; ++q) row[q] = make_float4(af2[2 * q].x, af2[2 * q].y, af2[2 * q + 1].x, af2[2 * q + 1].y);
      }
      row[F / 4 + 0] = make_float4(a_nx * ddx, a_ny * ddy, a_ca, a_cb);
      row[F / 4 + 1] = make_float4(a_cc, a_op, a_r, a_g);
      row[F / 4 + 2] = make_float4(a_b, a_d, 0.f, 0.f);
      a.row_flags[slot] = 1;
    }
    wave_lds_sync2();   // carries written by lane 63 are read by the next chunk
  }
}

int launch_render_bwd_gs(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, const GeomBuf& g,
                         const BinBuf& bb, const ImgBuf& im, const TraseRastGrads& gr, float* rows, uint8_t* row_flags,
                         const float* out_depth) {
  BwdGsArgs a;
  a.lineage = c.variant & (TRASE_VARIANT_FEATS_BG | TRASE_VARIANT_DEPTH_NORM); a.feat_bg = s.feat_bg; a.out_depth = out_depth;
  if ((a.lineage & TRASE_VARIANT_DEPTH_NORM) && gr.dL_ddepth && !out_depth) {
    set_error("render_bwd: the normalised-depth switch with a depth cotangent needs the forward's depth map (outputs.depth)");
    return TRASE_ERR_INVALID;
  }
  a.ranges = bb.ranges; a.point_list = bb.point_list; a.xy = g.xy; a.conic_o = g.conic_o; a.rgbd = g.rgbd;
  a.feats = in.sh_objs; a.bg = s.bg;
  a.d_img = gr.dL_dimage; a.d_feat = gr.dL_dfeats; a.d_depth = gr.dL_ddepth;
  a.final_T = im.final_T; a.n_contrib = im.n_contrib; a.pair_slot = bb.pair_slot; a.hdr = g.hdr; a.geo = g.geo; a.rows = rows; a.row_flags = row_flags;
  a.W = s.image_width; a.H = s.image_height;
  a.gx8 = (a.W + SUB - 1) / SUB;
  { int lo, hi; strip_subtile_rows(s, lo, hi); a.tile0 = lo * a.gx8; a.ntiles = (hi - lo) * a.gx8; }
  if (a.ntiles <= 0) return TRASE_OK;                    // an empty strip: no rows (the caller has cleared the flags)
  const int blocks = (a.ntiles + GWPB - 1) / GWPB;
  {
    ProfScope ps("render_bwd", c.stream);
    switch (in.F) {
#define TRASE_BWD_GS(FF) hipLaunchKernelGGL((render_bwd_gs_kernel<FF, true>), dim3(blocks), dim3(GWPB * WAVE), 0, c.stream, a)
      case 0: TRASE_BWD_GS(0); break;
      case 16: TRASE_BWD_GS(16); break;
      case 32: TRASE_BWD_GS(32); break;
      default: set_error("render_bwd: feature width %d not compiled in (0,16,32)", in.F); return TRASE_ERR_UNSUPPORTED;
    }
  }
  TRASE_POST_LAUNCH("render_bwd", c.stream, c.debug);
  return TRASE_OK;
}

}  // namespace trase
