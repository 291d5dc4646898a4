// render.hip -- alpha compositing of the per-tile depth-ordered lists (forward) and its backward.
// Variant 0 ("valu"): one workgroup of 4 waves per 16x16 tile, each wave owns an 8x8 pixel
// quadrant; batches of 256 list entries are staged through LDS; the per-Gaussian feature row
// is fetched with a wave-uniform address (scalar-cache path), and only when some lane of the
// wave actually blends that Gaussian.
// Semantics: SURVEY.md Appendix A "Render fwd" / "Render bwd"; outputs as consumed at
// gaussian_renderer/__init__.py:137-155 (image incl. background, feats without, blended depth).
#include "common.h"

namespace trase {

constexpr int RB = 256;   // list entries staged per batch == threads per workgroup

struct RenderArgs {
  const uint2* ranges; const uint32_t* point_list;
  const float2* xy; const float4* conic_o; const float4* rgbd; const float* feats; const float* bg;
  int W, H, gx, gy;
};

__device__ __forceinline__ void pixel_of_thread(int tile, int gx, int& px, int& py, int& wave, int& lane) {
  wave = threadIdx.x >> 6;
  lane = threadIdx.x & 63;
  const int tx = tile % gx, ty = tile / gx;
  px = tx * TILE + (wave & 1) * 8 + (lane & 7);
  py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
}

template <int F>
__global__ __launch_bounds__(RB) void render_fwd_kernel(RenderArgs a, float* __restrict__ out_img,
                                                        float* __restrict__ out_feat, float* __restrict__ out_depth,
                                                        float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
  __shared__ float2 s_xy[RB];
  __shared__ float4 s_co[RB];
  __shared__ float4 s_cd[RB];
  __shared__ uint32_t s_id[RB];
  const int tile = blockIdx.x;
  int px, py, wave, lane;
  pixel_of_thread(tile, a.gx, px, py, wave, lane);
  const bool inside = px < a.W && py < a.H;
  const float pxf = (float)px, pyf = (float)py;
  const uint2 range = a.ranges[tile];
  float T = 1.0f;
  uint32_t contributor = 0, last = 0;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f, cd = 0.f;
  float fa[F > 0 ? F : 1];
#pragma unroll
  for (int c = 0; c < F; ++c) fa[c] = 0.f;
  bool done = !inside;
  for (uint32_t base = range.x; base < range.y; base += RB) {
    if (__syncthreads_and(done)) break;
    const uint32_t n = min((uint32_t)RB, range.y - base);
    if (threadIdx.x < n) {
      const uint32_t id = a.point_list[base + threadIdx.x];
      s_id[threadIdx.x] = id;
      s_xy[threadIdx.x] = a.xy[id];
      s_co[threadIdx.x] = a.conic_o[id];
      s_cd[threadIdx.x] = a.rgbd[id];
    }
    __syncthreads();
    for (uint32_t j = 0; j < n; ++j) {
      if (__all(done)) break;
      ++contributor;
      const float2 g = s_xy[j];
      const float4 co = s_co[j];
      const float dx = g.x - pxf, dy = g.y - pyf;
      const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
      const float alpha = fminf(ALPHA_MAX, co.w * __expf(power));
      bool ok = !done && power <= 0.0f && alpha >= ALPHA_MIN;
      const float test_T = T * (1.0f - alpha);
      if (ok && test_T < T_STOP) { done = true; ok = false; }
      if (__any(ok)) {
        const float w = ok ? alpha * T : 0.0f;
        const float4 col = s_cd[j];
        c0 += w * col.x; c1 += w * col.y; c2 += w * col.z; cd += w * col.w;
        if (F > 0) {
          const uint32_t id = __builtin_amdgcn_readfirstlane(s_id[j]);
          const float* __restrict__ f = a.feats + (size_t)id * F;
#pragma unroll
          for (int c = 0; c < F; ++c) fa[c] += w * f[c];
        }
        if (ok) { T = test_T; last = contributor; }
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
    out_depth[pix] = cd;
#pragma unroll
    for (int c = 0; c < F; ++c) out_feat[(size_t)c * hw + pix] = fa[c];
  }
}

int launch_render_fwd(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, const TraseRastOutputs& out,
                      const GeomBuf& g, const BinBuf& b, const ImgBuf& im) {
  RenderArgs a;
  a.ranges = b.ranges; a.point_list = b.point_list; a.xy = g.xy; a.conic_o = g.conic_o; a.rgbd = g.rgbd;
  a.feats = in.sh_objs; a.bg = s.bg; a.W = s.image_width; a.H = s.image_height;
  a.gx = (a.W + TILE - 1) / TILE; a.gy = (a.H + TILE - 1) / TILE;
  const int T = a.gx * a.gy;
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

// ---- backward -----------------------------------------------------------------------------------
// Back-to-front over the same lists.  With s_g = <channels of Gaussian g, pixel cotangent> the
// whole multi-channel recurrence collapses to scalars:
//   dL/dalpha_g = T_g * (s_g - A_g),   A_g = alpha_{g+1} s_{g+1} + (1 - alpha_{g+1}) A_{g+1},  A_last = <bg, d_rgb>
// Per-Gaussian sums over the wave's 64 pixels are reduced with DPP before one atomic per value.
struct RenderBwdArgs {
  RenderArgs r;
  const float* d_img; const float* d_feat; const float* d_depth;
  const float* final_T; const uint32_t* n_contrib;
  float* acc;        // (P, BWD_ACC)
  float* d_feats_out;  // (P, F) or null
};

template <int F>
__global__ __launch_bounds__(RB) void render_bwd_kernel(RenderBwdArgs b) {
  const RenderArgs& a = b.r;
  __shared__ float2 s_xy[RB];
  __shared__ float4 s_co[RB];
  __shared__ float4 s_cd[RB];
  __shared__ uint32_t s_id[RB];
  const int tile = blockIdx.x;
  int px, py, wave, lane;
  pixel_of_thread(tile, a.gx, px, py, wave, lane);
  const bool inside = px < a.W && py < a.H;
  const float pxf = (float)px, pyf = (float)py;
  const uint2 range = a.ranges[tile];
  const uint32_t todo = range.y - range.x;
  const size_t hw = (size_t)a.H * a.W;
  const size_t pix = (size_t)py * a.W + px;
  // pixel cotangents
  float g0 = 0.f, g1 = 0.f, g2 = 0.f, gd = 0.f;
  float gf[F > 0 ? F : 1];
#pragma unroll
  for (int c = 0; c < F; ++c) gf[c] = 0.f;
  float T_final = 0.f;
  uint32_t last = 0;
  if (inside) {
    T_final = b.final_T[pix];
    last = b.n_contrib[pix];
    if (b.d_img) { g0 = b.d_img[pix]; g1 = b.d_img[hw + pix]; g2 = b.d_img[2 * hw + pix]; }
    if (b.d_depth) gd = b.d_depth[pix];
    if (F > 0 && b.d_feat) {
#pragma unroll
      for (int c = 0; c < F; ++c) gf[c] = b.d_feat[(size_t)c * hw + pix];
    }
  }
  float T = T_final;
  float A = a.bg[0] * g0 + a.bg[1] * g1 + a.bg[2] * g2;   // "colour behind", projected on the cotangent
  const float ddx = 0.5f * (float)a.W, ddy = 0.5f * (float)a.H;
  uint32_t contributor = todo;
  // highest list position any pixel of this workgroup blended: entries behind it are skipped
  for (uint32_t prog = 0; prog < todo; prog += RB) {
    const uint32_t n = min((uint32_t)RB, todo - prog);
    __syncthreads();
    if (threadIdx.x < n) {
      const uint32_t id = a.point_list[range.y - 1 - prog - threadIdx.x];
      s_id[threadIdx.x] = id;
      s_xy[threadIdx.x] = a.xy[id];
      s_co[threadIdx.x] = a.conic_o[id];
      s_cd[threadIdx.x] = a.rgbd[id];
    }
    __syncthreads();
    for (uint32_t j = 0; j < n; ++j) {
      --contributor;   // 0-based list position of this entry
      const bool live = contributor < last;
      if (!__any(live)) continue;
      const float2 g = s_xy[j];
      const float4 co = s_co[j];
      const float dx = g.x - pxf, dy = g.y - pyf;
      const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
      const float G = __expf(power);
      const float alpha = fminf(ALPHA_MAX, co.w * G);
      const bool ok = live && power <= 0.0f && alpha >= ALPHA_MIN;
      if (!__any(ok)) continue;
      const uint32_t id = __builtin_amdgcn_readfirstlane(s_id[j]);
      const float4 col = s_cd[j];
      float s = col.x * g0 + col.y * g1 + col.z * g2 + col.w * gd;
      const float* __restrict__ f = a.feats + (size_t)id * F;
      float fr[F > 0 ? F : 1];
      if (F > 0) {
#pragma unroll
        for (int c = 0; c < F; ++c) { fr[c] = f[c]; s += fr[c] * gf[c]; }
      }
      float wgt = 0.f, dL_dalpha = 0.f;
      if (ok) {
        T = T / (1.0f - alpha);
        wgt = alpha * T;
        dL_dalpha = T * (s - A);
        A = alpha * s + (1.0f - alpha) * A;
      }
      const float dL_dG = co.w * dL_dalpha;
      const float gdx = G * dx, gdy = G * dy;
      const float dG_ddelx = -gdx * co.x - gdy * co.y;
      const float dG_ddely = -gdy * co.z - gdx * co.y;
      // 10 geometry/colour sums + F feature sums, reduced over the wave then one atomic each
      float mine = 0.f;
      float r;
#define TRASE_RED(slot, expr)                                   \
  r = readlane_f(wave_sum_lane63(expr), 63);                    \
  if (lane == (slot)) mine = r;
      TRASE_RED(F + ACC_NDCX, dL_dG * dG_ddelx * ddx)
      TRASE_RED(F + ACC_NDCY, dL_dG * dG_ddely * ddy)
      TRASE_RED(F + ACC_CA, -0.5f * gdx * dx * dL_dG)
      TRASE_RED(F + ACC_CB, -gdx * dy * dL_dG)
      TRASE_RED(F + ACC_CC, -0.5f * gdy * dy * dL_dG)
      TRASE_RED(F + ACC_OP, G * dL_dalpha)
      TRASE_RED(F + ACC_R, wgt * g0)
      TRASE_RED(F + ACC_G, wgt * g1)
      TRASE_RED(F + ACC_B, wgt * g2)
      TRASE_RED(F + ACC_D, wgt * gd)
      if (F > 0) {
#pragma unroll
        for (int c = 0; c < F; ++c) { TRASE_RED(c, wgt * gf[c]) }
      }
#undef TRASE_RED
      if (lane < F) {
        if (b.d_feats_out) atomic_add_f32(b.d_feats_out + (size_t)id * F + lane, mine);
      } else if (lane < F + 10) {
        atomic_add_f32(b.acc + (size_t)id * BWD_ACC + (lane - F), mine);
      }
    }
  }
}

int launch_render_bwd(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, const GeomBuf& g,
                      const BinBuf& bb, const ImgBuf& im, const TraseRastGrads& gr, float* acc) {
  RenderBwdArgs b;
  RenderArgs& a = b.r;
  a.ranges = bb.ranges; a.point_list = bb.point_list; a.xy = g.xy; a.conic_o = g.conic_o; a.rgbd = g.rgbd;
  a.feats = in.sh_objs; a.bg = s.bg; a.W = s.image_width; a.H = s.image_height;
  a.gx = (a.W + TILE - 1) / TILE; a.gy = (a.H + TILE - 1) / TILE;
  b.d_img = gr.dL_dimage; b.d_feat = gr.dL_dfeats; b.d_depth = gr.dL_ddepth;
  b.final_T = im.final_T; b.n_contrib = im.n_contrib; b.acc = acc; b.d_feats_out = gr.dL_dsh_objs;
  const int T = a.gx * a.gy;
  {
    ProfScope ps("render_bwd", c.stream);
    switch (in.F) {
      case 0: hipLaunchKernelGGL(render_bwd_kernel<0>, dim3(T), dim3(RB), 0, c.stream, b); break;
      case 16: hipLaunchKernelGGL(render_bwd_kernel<16>, dim3(T), dim3(RB), 0, c.stream, b); break;
      case 32: hipLaunchKernelGGL(render_bwd_kernel<32>, dim3(T), dim3(RB), 0, c.stream, b); break;
      default: set_error("render_bwd: feature width %d not compiled in (0,16,32)", in.F); return TRASE_ERR_UNSUPPORTED;
    }
  }
  TRASE_POST_LAUNCH("render_bwd", c.stream, c.debug);
  return TRASE_OK;
}

}  // namespace trase
