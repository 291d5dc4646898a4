// render.hip -- alpha compositing of the per-tile depth-ordered lists, packed-FP32 forward (F = 0 / 16; F = 32 as the
// cross-check of the MFMA forward, TRASE_VARIANT_VALU_FORWARD).
// Variant 0 ("valu"): one wave per 8x8 sub-tile (4 independent waves per workgroup, no workgroup
// barriers); batches of 64 list entries are staged through wave-private LDS; the per-Gaussian
// feature row is fetched with a wave-uniform address and only when some lane of the wave
// actually blends that Gaussian.
// Semantics: SURVEY.md Appendix A "Render fwd" / "Render bwd"; outputs as consumed at
// gaussian_renderer/__init__.py:137-155 (image incl. background, feats without, blended depth).
#include "common.h"

namespace trase {

constexpr int WPB = 4;     // independent waves (sub-tiles) per workgroup
constexpr int RB = WPB * WAVE;

struct RenderArgs {
  const uint2* ranges; const uint32_t* point_list;
  const float2* xy; const float4* conic_o; const float4* rgbd; const float* feats; const float* bg;
  int W, H, gx8, ntiles;
  int tile0;             // first sub-tile of the strip being rendered (ntiles counts the strip's sub-tiles)
  // forward only: when set, the list still holds emit-order slots; the staging step translates them to Gaussian
  // ids (slot -> id is one more dependent load, hidden like the others) and records the ids for the backward
  const uint32_t* pair_slot; const uint32_t* pair_gauss; uint32_t* point_list_w; uint32_t cap;
  const uint32_t* hdr;
  int lineage;           // variant bits TRASE_VARIANT_FEATS_BG / TRASE_VARIANT_DEPTH_NORM (0 = public lineage)
  float feat_bg;
};

// LDS produced and consumed by one wave only: order the accesses without a workgroup barrier
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// one wave == one 8x8 sub-tile; lane -> pixel
__device__ __forceinline__ int subtile_of_wave(const RenderArgs& a, int& px, int& py, int& wave, int& lane) {
  wave = threadIdx.x >> 6;
  lane = threadIdx.x & 63;
  const int local = xcd_block(blockIdx.x, gridDim.x) * WPB + wave;
  if (local >= a.ntiles) return -1;
  const int tile = a.tile0 + local;
  const int tx = tile % a.gx8, ty = tile / a.gx8;
  px = tx * SUB + (lane & 7);
  py = ty * SUB + (lane >> 3);
  return tile;
}

template <int F>
__global__ __launch_bounds__(RB) void render_fwd_kernel(RenderArgs a, float* __restrict__ out_img,
                                                        float* __restrict__ out_feat, float* __restrict__ out_depth,
                                                        float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
  // per staged list entry: the blend-exponent polynomial (common.h pair_poly), colour+depth, id
  __shared__ float4 s_k0[WPB][WAVE];   // k0, kj, ki, kjj
  __shared__ float4 s_k1[WPB][WAVE];   // kii, kij, thr, id (bits)
  __shared__ float4 s_cd[WPB][WAVE];
  int px, py, wave, lane;
  const int tile = subtile_of_wave(a, px, py, wave, lane);
  if (tile < 0) return;
  const bool inside = px < a.W && py < a.H;
  const float bx = (float)((tile % a.gx8) * SUB), by = (float)((tile / a.gx8) * SUB);
  const float fj = (float)(lane & 7), fi = (float)(lane >> 3);
  const float fii = fi * fi;
  const uint2 range = a.ranges[tile];
  float T = 1.0f;
  uint32_t last = 0;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f, cd = 0.f;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 fa2[F > 0 ? F / 2 : 1];                          // channel pairs: one v_pk_fma_f32 per pair
#pragma unroll
  for (int c = 0; c < F / 2; ++c) fa2[c] = f32x2{0.f, 0.f};
  // a finished pixel (outside the image, or transmittance exhausted) is encoded as live == 0
  float live = inside ? 1.0f : 0.0f;
  const uint32_t jb = a.point_list_w ? a.hdr[HDR_PACK] : 0u;
  for (uint32_t base = range.x; base < range.y; base += WAVE) {
    if (!__any(live != 0.0f)) break;
    const uint32_t n = min((uint32_t)WAVE, range.y - base);
    wave_lds_sync();
    uint32_t my_id = 0;                                  // lane = list entry: kept for v_readlane in the blend loop
    if ((uint32_t)lane < n) {
      uint32_t id;
      if (a.point_list_w) {
        const uint32_t slot = a.pair_slot[base + lane];
        id = jb ? (slot >> jb) : a.pair_gauss[slot < a.cap ? slot : 0];   // HDR_PACK: the id rides in the list value
        if (!jb) a.point_list_w[base + lane] = id;
      } else {
        id = a.point_list[base + lane];
      }
      my_id = id;
      // pull this entry's 128-B feature row towards the L2 now: the blend loop reads it through the scalar cache up to
      // 64 entries later.  The loaded word is unused; its register stays reserved until the explicit wait below
      // (the compiler does not know an asm load is pending).
      uint32_t sink = 0;
      if (F > 0) asm volatile("global_load_dword %0, %1, off" : "=v"(sink) : "v"(a.feats + (size_t)id * F));
      const PairPoly k = pair_poly(a.xy[id], a.conic_o[id], bx, by);
      s_k0[wave][lane] = make_float4(k.k0, k.kj, k.ki, k.kjj);
      s_k1[wave][lane] = make_float4(k.kii, k.kij, k.thr, __uint_as_float(id));
      s_cd[wave][lane] = a.rgbd[id];
      if (F > 0) asm volatile("s_waitcnt vmcnt(0)" ::"v"(sink));   // every staging load has been consumed by now
    }
    wave_lds_sync();
    for (uint32_t j = 0; j < n; ++j) {
      const float4 q0 = s_k0[wave][j];
      const float4 q1 = s_k1[wave][j];
      PairPoly k;
      k.k0 = q0.x; k.kj = q0.y; k.ki = q0.z; k.kjj = q0.w; k.kii = q1.x; k.kij = q1.y; k.thr = q1.z;
      const float e = poly_eval(k, poly_row_base(k, fi, fii), poly_row_slope(k, fi), fj);
      const float alpha = fminf(ALPHA_MAX, __builtin_amdgcn_exp2f(e));
      const bool gate = (live != 0.0f) && (e <= k.thr) && (e >= LOG2_ALPHA_MIN);
      const float test_T = T * (1.0f - alpha);
      const bool stop = gate && (test_T < T_STOP);       // this one is NOT blended and the pixel is finished
      const bool ok = gate && !stop;
      live = stop ? 0.0f : live;
      if (__any(ok)) {
        const float w = ok ? alpha * T : 0.0f;
        const float4 col = s_cd[wave][j];
        c0 = fmaf(w, col.x, c0); c1 = fmaf(w, col.y, c1); c2 = fmaf(w, col.z, c2); cd = fmaf(w, col.w, cd);
        if (F > 0) {
          // wave-uniform row: read it through the scalar cache (constant address space => s_load)
          // the row address comes from a register, not from the LDS record: the scalar load does not wait for LDS
          const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)my_id, (int)j);
          typedef __attribute__((address_space(4))) const f32x2 cfloat2;
          cfloat2* f = (cfloat2*)(a.feats + (size_t)id * F);
          const f32x2 w2 = {w, w};
#pragma unroll
          for (int c = 0; c < F / 2; ++c) fa2[c] = __builtin_elementwise_fma(w2, f[c], fa2[c]);
        }
        T = ok ? test_T : T;
        last = ok ? (base - range.x + j + 1) : last;
      }
    }
  }
  if (inside) {
    const size_t hw = (size_t)a.H * a.W;
    const size_t pix = (size_t)py * a.W + px;
    final_T[pix] = T;
    n_contrib[pix] = last;
    out_img[pix] = c0 + T * a.bg[0];
    out_img[hw + pix] = c1 + T * a.bg[1];
    out_img[2 * hw + pix] = c2 + T * a.bg[2];
    if (a.lineage & TRASE_VARIANT_DEPTH_NORM) { const float A = 1.0f - T; cd = A > 1e-10f ? cd / A : 0.0f; }
    out_depth[pix] = cd;
    const float fb = (a.lineage & TRASE_VARIANT_FEATS_BG) ? a.feat_bg : 0.0f;
#pragma unroll
    for (int c = 0; c < F; ++c)
      out_feat[(size_t)c * hw + pix] = (a.lineage & TRASE_VARIANT_FEATS_BG) ? fmaf(T, fb, fa2[c >> 1][c & 1]) : fa2[c >> 1][c & 1];
  }
}

static void fill_render_args(RenderArgs& a, const TraseRastSettings& s, const TraseRastInputs& in, const GeomBuf& g,
                             const BinBuf& b) {
  a.ranges = b.ranges; a.point_list = b.point_list; a.xy = g.xy; a.conic_o = g.conic_o; a.rgbd = g.rgbd;
  a.feats = in.sh_objs; a.bg = s.bg; a.W = s.image_width; a.H = s.image_height;
  a.gx8 = (a.W + SUB - 1) / SUB;
  { int lo, hi; strip_subtile_rows(s, lo, hi); a.tile0 = lo * a.gx8; a.ntiles = (hi - lo) * a.gx8; }
  a.pair_slot = nullptr; a.pair_gauss = nullptr; a.point_list_w = nullptr; a.cap = 0;
  a.lineage = 0; a.feat_bg = s.feat_bg;
}

int launch_render_fwd(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, const TraseRastOutputs& out,
                      const GeomBuf& g, const BinBuf& b, const ImgBuf& im, const uint32_t* pair_gauss, uint32_t cap) {
  if (in.F == 32 && !(c.variant & TRASE_VARIANT_VALU_FORWARD))               // default: channel accumulation on the matrix cores
    return launch_render_fwd_mf(c, s, in, out, g, b, im, pair_gauss, cap);
  RenderArgs a;
  fill_render_args(a, s, in, g, b);
  a.lineage = c.variant & (TRASE_VARIANT_FEATS_BG | TRASE_VARIANT_DEPTH_NORM);
  if (pair_gauss) { a.pair_slot = b.pair_slot; a.pair_gauss = pair_gauss; a.point_list_w = b.point_list; a.cap = cap; }
  a.hdr = g.hdr;
  if (a.ntiles <= 0) return TRASE_OK;                    // an empty strip
  const int T = (a.ntiles + WPB - 1) / WPB;
  {
    ProfScope ps("render_fwd", c.stream);
    switch (in.F) {
      case 0: hipLaunchKernelGGL(render_fwd_kernel<0>, dim3(T), dim3(RB), 0, c.stream, a, out.image, out.feats, out.depth, im.final_T, im.n_contrib); break;
      case 16: hipLaunchKernelGGL(render_fwd_kernel<16>, dim3(T), dim3(RB), 0, c.stream, a, out.image, out.feats, out.depth, im.final_T, im.n_contrib); break;
      case 32: hipLaunchKernelGGL(render_fwd_kernel<32>, dim3(T), dim3(RB), 0, c.stream, a, out.image, out.feats, out.depth, im.final_T, im.n_contrib); break;
      default: set_error("render_fwd: feature width %d not compiled in (0,16,32)", in.F); return TRASE_ERR_UNSUPPORTED;
    }
  }
  TRASE_POST_LAUNCH("render_fwd", c.stream, c.debug);
  return TRASE_OK;
}

}  // namespace trase
