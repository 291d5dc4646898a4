// preprocess.hip -- per-Gaussian forward (projection, EWA covariance, conic, SH colour, tile
// count, depth key) and its backward.  One thread per Gaussian; the arithmetic lives in
// gs_math.h.  Replaces the lineage's preprocessCUDA / computeCov2DCUDA+preprocessCUDA backward
// (SURVEY.md 2.2); call-site contract gaussian_renderer/__init__.py:137-146.
#include "common.h"

namespace trase {

__device__ __forceinline__ void load_view(const float* __restrict__ vm, const float* __restrict__ pm,
                                          const float* __restrict__ cam, View& v) {
#pragma unroll
  for (int i = 0; i < 16; ++i) { v.V[i] = vm[i]; v.PM[i] = pm[i]; }
  v.cam[0] = cam[0]; v.cam[1] = cam[1]; v.cam[2] = cam[2];
}

struct PreArgs {
  const float* means3D; const float* shs; const float* colors; const float* opac; const float* scales;
  const float* rots; const float* cov3d; const float* vm; const float* pm; const float* cam;
  int P, M, deg, W, H;
  float tanx, tany, mod;
  int sy_lo, sy_hi;          // sub-tile rows of the strip being rendered
  int key27;                 // 27-bit depth keys (common.h depth_sort_key)
};

__device__ __forceinline__ void fill_view(const PreArgs& a, View& v) {
  load_view(a.vm, a.pm, a.cam, v);
  v.tanx = a.tanx; v.tany = a.tany;
  v.fx = (float)a.W / (2.0f * a.tanx); v.fy = (float)a.H / (2.0f * a.tany);
  v.mod = a.mod; v.W = a.W; v.H = a.H;
  v.gx = (a.W + TILE - 1) / TILE; v.gy = (a.H + TILE - 1) / TILE;
  v.deg = a.deg;
}

template <bool USE_COV, bool USE_SH>
__global__ __launch_bounds__(256) void preprocess_fwd_kernel(PreArgs a, int32_t* __restrict__ radii,
                                                             float2* __restrict__ xy, float4* __restrict__ conic_o,
                                                             float4* __restrict__ rgbd, float4* __restrict__ geo, uint32_t* __restrict__ tiles,
                                                             uint32_t* __restrict__ clamped,
                                                             uint32_t* __restrict__ depth_keys,
                                                             uint32_t* __restrict__ hdr) {
  const int gi = blockIdx.x * blockDim.x + threadIdx.x;
  if (gi == 0) {                                     // the header starts clean (nobody else touches it in this kernel: one
#pragma unroll                                       // memset launch less per view) + a device-resident copy of P for the depth sort
    for (int k = 0; k < HDR_WORDS - 1; ++k) hdr[k] = 0u;
    hdr[HDR_WORDS - 1] = (uint32_t)a.P;
  }
  // no early return: the wave-level reduction below needs lane 63 of every wave alive
  const bool active = gi < a.P;
  const int i = active ? gi : a.P - 1;
  View v;
  fill_view(a, v);
  const float p[3] = {a.means3D[3 * i], a.means3D[3 * i + 1], a.means3D[3 * i + 2]};
  float sc[3] = {0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f}, cv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (USE_COV) {
#pragma unroll
    for (int k = 0; k < 6; ++k) cv[k] = a.cov3d[6 * i + k];
  } else {
    sc[0] = a.scales[3 * i]; sc[1] = a.scales[3 * i + 1]; sc[2] = a.scales[3 * i + 2];
    const float4 qq = reinterpret_cast<const float4*>(a.rots)[i];
    q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
  }
  float shl[48];
  float col[3] = {0.f, 0.f, 0.f};
  if (USE_SH) {
    const int n3 = 3 * ncoef(a.deg);
    const float* src = a.shs + (size_t)i * a.M * 3;
    if (a.M == 16) {            // full rows: 12 aligned 16-byte loads
      const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const float4 t = s4[k];
        shl[4 * k] = t.x; shl[4 * k + 1] = t.y; shl[4 * k + 2] = t.z; shl[4 * k + 3] = t.w;
      }
#pragma unroll
      for (int k = 0; k < 48; ++k) shl[k] = (k < n3) ? shl[k] : 0.f;
    } else {
#pragma unroll
      for (int k = 0; k < 48; ++k) shl[k] = (k < n3) ? src[k] : 0.f;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 48; ++k) shl[k] = 0.f;
    col[0] = a.colors[3 * i]; col[1] = a.colors[3 * i + 1]; col[2] = a.colors[3 * i + 2];
  }
  Splat o;
  const bool vis = splat_forward<USE_COV, USE_SH>(v, p, sc, q, cv, shl, col, o) && active;
  if (active) radii[i] = vis ? o.radius : 0;
  // (the lineage pair count R is totalled by scan_partial_kernel from radii + centres: no same-address atomics here)
  // live 8x8 sub-tiles inside the rect (exact culling, see gs_math.h)
  uint32_t live = 0;
  const float opac = a.opac[i];
  if (vis) {
    const SubtileCull cull = subtile_cull_setup(o.px, o.py, o.ca, o.cb, o.cc, opac);
    // per sub-tile row the live columns are an interval (subtile_row_live): O(rows) per splat
    for (int sy = max(2 * o.y0, a.sy_lo); sy < min(2 * o.y1, a.sy_hi) && sy * SUB < a.H; ++sy) {
      int c0, c1;
      subtile_row_live(cull, sy, a.W, a.H, 2 * o.x0, 2 * o.x1, c0, c1);
      live += (uint32_t)(c1 - c0);
    }
  }
  if (active) {
    tiles[i] = live;
    depth_keys[i] = (vis && live) ? depth_sort_key(o.depth, a.key27 != 0) : (a.key27 ? DEPTH27_DEAD : 0xffffffffu);
  }
  if (vis) {
    xy[i] = make_float2(o.px, o.py);
    conic_o[i] = make_float4(o.ca, o.cb, o.cc, opac);
    rgbd[i] = make_float4(o.rgb[0], o.rgb[1], o.rgb[2], o.depth);
    // the same 40 bytes as ONE 64-byte record: the compositing kernels fetch a list entry's geometry from one cache line
    // (word 2 of the record: the Gaussian's first row slot, written by emit_pairs)
    geo[4 * (size_t)i + 0] = make_float4(o.px, o.py, 0.f, __int_as_float(o.radius));   // .w: the radius, for emit_pairs
    geo[4 * (size_t)i + 1] = make_float4(o.ca, o.cb, o.cc, opac);
    geo[4 * (size_t)i + 2] = make_float4(o.rgb[0], o.rgb[1], o.rgb[2], o.depth);
    geo[4 * (size_t)i + 3] = split_rgbd(o.rgb[0], o.rgb[1], o.rgb[2], o.depth);
    clamped[i] = o.clamped;
  }
}

static void fill_pre_args(PreArgs& a, const TraseRastSettings& s, const TraseRastInputs& in) {
  a.means3D = in.means3D; a.shs = in.shs; a.colors = in.colors_precomp; a.opac = in.opacities;
  a.scales = in.scales; a.rots = in.rotations; a.cov3d = in.cov3D_precomp;
  a.vm = s.viewmatrix; a.pm = s.projmatrix; a.cam = s.campos;
  a.P = in.P; a.M = in.M; a.deg = s.sh_degree; a.W = s.image_width; a.H = s.image_height;
  a.tanx = s.tanfovx; a.tany = s.tanfovy; a.mod = s.scale_modifier;
  strip_subtile_rows(s, a.sy_lo, a.sy_hi);
}

int launch_preprocess_fwd(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, int32_t* radii,
                          const GeomBuf& g, uint32_t* depth_keys, bool key27) {
  PreArgs a;
  fill_pre_args(a, s, in);
  a.key27 = key27 ? 1 : 0;
  const dim3 grid((in.P + 255) / 256), block(256);
  const bool cov = in.cov3D_precomp != nullptr, sh = in.shs != nullptr;
  {
    ProfScope ps("preprocess_fwd", c.stream);
#define TRASE_PRE_FWD(C, S) hipLaunchKernelGGL((preprocess_fwd_kernel<C, S>), grid, block, 0, c.stream, a, radii, g.xy, g.conic_o, g.rgbd, g.geo, g.tiles, g.clamped, depth_keys, g.hdr)
    if (cov) { if (sh) TRASE_PRE_FWD(true, true); else TRASE_PRE_FWD(true, false); }
    else { if (sh) TRASE_PRE_FWD(false, true); else TRASE_PRE_FWD(false, false); }
#undef TRASE_PRE_FWD
  }
  TRASE_POST_LAUNCH("preprocess_fwd", c.stream, c.debug);
  if (in.F == 32) return launch_feature_table(c, in.sh_objs, g.tiles, in.P, g.ftab);
  return TRASE_OK;
}

// F = 32: feature rows -> bf16 [hi 32 | lo 32] (GeomBuf::ftab).  One thread per (Gaussian, four channels); the fused
// (raw) preprocess writes the same table itself.
__global__ __launch_bounds__(256) void feature_table_kernel(const float* __restrict__ feats, const uint32_t* __restrict__ tiles,
                                                            int P, uint32_t* __restrict__ ftab) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * 8) return;
  const int g = idx >> 3, l = idx & 7;
  if (tiles[g] == 0) return;                                 // in no list: its row is never read
  const float4 v = reinterpret_cast<const float4*>(feats + (size_t)g * 32)[l];
  unsigned h01, l01, h23, l23;
  split_pk(v.x, v.y, h01, l01);
  split_pk(v.z, v.w, h23, l23);
  *reinterpret_cast<uint2*>(ftab + (size_t)g * 32 + 2 * l) = make_uint2(h01, h23);
  *reinterpret_cast<uint2*>(ftab + (size_t)g * 32 + 16 + 2 * l) = make_uint2(l01, l23);
}
int launch_feature_table(const LaunchCtx& c, const float* feats, const uint32_t* tiles, int P, uint32_t* ftab) {
  if (P <= 0) return TRASE_OK;
  {
    ProfScope ps("feature_table", c.stream);
    hipLaunchKernelGGL(feature_table_kernel, dim3((P * 8 + 255) / 256), dim3(256), 0, c.stream, feats, tiles, P, ftab);
  }
  TRASE_POST_LAUNCH("feature_table", c.stream, c.debug);
  return TRASE_OK;
}

// ---- backward -----------------------------------------------------------------------------------
struct PreBwdOut {
  float* d_means3D; float* d_means2D; float* d_shs; float* d_colors; float* d_opac; float* d_scales;
  float* d_rots; float* d_cov3d;
};

template <bool USE_COV, bool USE_SH>
__global__ __launch_bounds__(256) void preprocess_bwd_kernel(PreArgs a, const int32_t* __restrict__ radii,
                                                             const uint32_t* __restrict__ clamped,
                                                             const uint32_t* __restrict__ tiles,
                                                             const float* __restrict__ acc, PreBwdOut o) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.P) return;
  // a Gaussian without a (sub-tile, Gaussian) pair -- culled, fainter than 1/255 everywhere, or outside the tile-row strip
  // being rendered -- received no gradient row: its gradients are exact zeros, nothing is evaluated
  const bool vis = radii[i] > 0 && tiles[i] > 0;
  SplatGradOut go;
  float dsh[48];
#pragma unroll
  for (int k = 0; k < 48; ++k) dsh[k] = 0.f;
  SplatGradIn gi;
  gi.d_ndcx = gi.d_ndcy = gi.d_ca = gi.d_cb = gi.d_cc = gi.d_depth = 0.f;
  gi.d_rgb[0] = gi.d_rgb[1] = gi.d_rgb[2] = 0.f;
  float d_op = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) { go.d_p[k] = 0.f; go.d_scale[k] = 0.f; }
#pragma unroll
  for (int k = 0; k < 4; ++k) go.d_quat[k] = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k) go.d_cov[k] = 0.f;
  if (vis) {
    const float4* r4 = reinterpret_cast<const float4*>(acc + (size_t)i * BWD_ACC);
    const float4 r0 = r4[0], r1 = r4[1], r2 = r4[2];
    gi.d_ndcx = r0.x; gi.d_ndcy = r0.y; gi.d_ca = r0.z; gi.d_cb = r0.w;
    gi.d_cc = r1.x; d_op = r1.y; gi.d_rgb[0] = r1.z; gi.d_rgb[1] = r1.w;
    gi.d_rgb[2] = r2.x; gi.d_depth = r2.y;
    View v;
    fill_view(a, v);
    const float p[3] = {a.means3D[3 * i], a.means3D[3 * i + 1], a.means3D[3 * i + 2]};
    float sc[3] = {0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f}, cv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (USE_COV) {
#pragma unroll
      for (int k = 0; k < 6; ++k) cv[k] = a.cov3d[6 * i + k];
    } else {
      sc[0] = a.scales[3 * i]; sc[1] = a.scales[3 * i + 1]; sc[2] = a.scales[3 * i + 2];
      const float4 qq = reinterpret_cast<const float4*>(a.rots)[i];
      q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
    }
    float shl[48];
    if (USE_SH) {
      const int n3 = 3 * ncoef(a.deg);
      const float* src = a.shs + (size_t)i * a.M * 3;
      if (a.M == 16) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          const float4 t = s4[k];
          shl[4 * k] = t.x; shl[4 * k + 1] = t.y; shl[4 * k + 2] = t.z; shl[4 * k + 3] = t.w;
        }
#pragma unroll
        for (int k = 0; k < 48; ++k) shl[k] = (k < n3) ? shl[k] : 0.f;
      } else {
#pragma unroll
        for (int k = 0; k < 48; ++k) shl[k] = (k < n3) ? src[k] : 0.f;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 48; ++k) shl[k] = 0.f;
    }
    splat_backward<USE_COV, USE_SH>(v, p, sc, q, cv, shl, USE_SH ? clamped[i] : 0u, gi, go, dsh);
  }
  if (o.d_means3D) { o.d_means3D[3 * i] = go.d_p[0]; o.d_means3D[3 * i + 1] = go.d_p[1]; o.d_means3D[3 * i + 2] = go.d_p[2]; }
  if (o.d_means2D) { o.d_means2D[3 * i] = gi.d_ndcx; o.d_means2D[3 * i + 1] = gi.d_ndcy; o.d_means2D[3 * i + 2] = 0.f; }
  if (o.d_opac) o.d_opac[i] = d_op;
  if (!USE_COV) {
    if (o.d_scales) { o.d_scales[3 * i] = go.d_scale[0]; o.d_scales[3 * i + 1] = go.d_scale[1]; o.d_scales[3 * i + 2] = go.d_scale[2]; }
    if (o.d_rots) reinterpret_cast<float4*>(o.d_rots)[i] = make_float4(go.d_quat[0], go.d_quat[1], go.d_quat[2], go.d_quat[3]);
  } else if (o.d_cov3d) {
#pragma unroll
    for (int k = 0; k < 6; ++k) o.d_cov3d[6 * i + k] = go.d_cov[k];
  }
  if (!USE_SH) {
    if (o.d_colors) { o.d_colors[3 * i] = gi.d_rgb[0]; o.d_colors[3 * i + 1] = gi.d_rgb[1]; o.d_colors[3 * i + 2] = gi.d_rgb[2]; }
  } else if (o.d_shs) {
    float* dst = o.d_shs + (size_t)i * a.M * 3;
    if (a.M == 16) {
      float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
      for (int k = 0; k < 12; ++k) d4[k] = make_float4(dsh[4 * k], dsh[4 * k + 1], dsh[4 * k + 2], dsh[4 * k + 3]);
    } else {
      const int m3 = a.M * 3;
#pragma unroll
      for (int k = 0; k < 48; ++k)
        if (k < m3) dst[k] = dsh[k];
    }
  }
}

int launch_preprocess_bwd(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, const int32_t* radii,
                          const GeomBuf& g, const float* acc, const TraseRastGrads& gr) {
  PreArgs a;
  fill_pre_args(a, s, in);
  PreBwdOut o;
  o.d_means3D = gr.dL_dmeans3D; o.d_means2D = gr.dL_dmeans2D; o.d_shs = in.shs ? gr.dL_dshs : nullptr;
  o.d_colors = in.colors_precomp ? gr.dL_dcolors : nullptr; o.d_opac = gr.dL_dopacities;
  o.d_scales = in.cov3D_precomp ? nullptr : gr.dL_dscales; o.d_rots = in.cov3D_precomp ? nullptr : gr.dL_drotations;
  o.d_cov3d = in.cov3D_precomp ? gr.dL_dcov3D : nullptr;
  const dim3 grid((in.P + 255) / 256), block(256);
  const bool cov = in.cov3D_precomp != nullptr, sh = in.shs != nullptr;
  {
    ProfScope ps("preprocess_bwd", c.stream);
#define TRASE_PRE_BWD(C, S) hipLaunchKernelGGL((preprocess_bwd_kernel<C, S>), grid, block, 0, c.stream, a, radii, g.clamped, g.tiles, acc, o)
    if (cov) { if (sh) TRASE_PRE_BWD(true, true); else TRASE_PRE_BWD(true, false); }
    else { if (sh) TRASE_PRE_BWD(false, true); else TRASE_PRE_BWD(false, false); }
#undef TRASE_PRE_BWD
  }
  TRASE_POST_LAUNCH("preprocess_bwd", c.stream, c.debug);
  return TRASE_OK;
}

}  // namespace trase
