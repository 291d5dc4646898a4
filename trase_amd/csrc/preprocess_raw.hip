// preprocess_raw.hip -- the per-Gaussian forward/backward with the reference's A1 "prep" fused in.
//
// gaussian_renderer.render() prepares the rasterizer inputs with ~10 small PyTorch kernels
// (gaussian_renderer/__init__.py:82-121; activations scene/gaussian_model.py:43-51,183-205):
//     means3D   = _xyz + d_xyz                         opacity = sigmoid(_opacity)
//     scales    = exp(_scaling) + d_scaling             rotations = normalize(_rotation) + d_rotation
//     shs       = cat(_features_dc, _features_rest)     sh_objs = f / (||f||_2 + 1e-9)
// and autograd runs their backward as ~15 more.  Here they happen in registers inside the per-Gaussian
// kernels: the forward reads the RAW parameters (plus the per-view deformation), the backward chains the
// activation derivatives and writes gradients of the raw parameters (and of the deformation, which the
// deformation MLP consumes).  The only extra buffer is the normalised feature row the compositing
// kernels read.  
#include "common.h"

namespace trase {

struct RawFwdArgs {
  const float* xyz; const float* d_xyz; const float* f_dc; const float* f_rest; const float* opacity;
  const float* scaling; const float* d_scaling; const float* rotation; const float* d_rotation;
  const float* features; float* featn;
  int write_featn;        // 0: nobody reads the fp32 rows (F = 32 default kernels read the bf16 table): skip the 128-byte store
  const float* vm; const float* pm; const float* cam;
  int P, F, deg, W, H, norm_features;
  float tanx, tany, mod;
  int sy_lo, sy_hi;          // sub-tile rows of the strip being rendered
  int strip;                 // forward: a tile-row strip is rendered (most Gaussians have no pair): colour and records only for those that do
  int p_begin, p_end;        // backward only: the Gaussians [p_begin, p_end) (p_begin a multiple of 64)
  int key27;                 // forward: 27-bit depth keys (common.h depth_sort_key)
  uint32_t key_or, key_dead; // forward: OR-ed into a live Gaussian's depth key / the key of one without a pair (the two-view forward keeps
                             // the view index in the sign bit of the float32 key: z > 0.2, so the bit is free)
  // render()'s other call patterns (gaussian_renderer/__init__.py:75-80, :103-113, :123-135), read by the INFER instantiations only:
  const float* colors;       // override_color (P,3): the colour as given, no SH evaluation
  const uint8_t* mask;       // mask (P bytes): a 0 removes the Gaussian from the view (the reference indexes every input by the mask)
  const float* se3;          // is_6dof (P,4,4): means3D = from_homogenous(M [xyz, 1]) instead of xyz + d_xyz
  int sh_dir_raw;            // pipe.convert_SHs_python: the SH direction is taken from the UNDEFORMED position (:105)
  int fwd_only;              // forward under no_grad: the arrays only the backward reads (rgbd, clamp bits) are not stored
};

__device__ __forceinline__ void raw_view(const RawFwdArgs& a, View& v) {
#pragma unroll
  for (int i = 0; i < 16; ++i) { v.V[i] = a.vm[i]; v.PM[i] = a.pm[i]; }
  v.cam[0] = a.cam[0]; v.cam[1] = a.cam[1]; v.cam[2] = a.cam[2];
  v.tanx = a.tanx; v.tany = a.tany;
  v.fx = (float)a.W / (2.0f * a.tanx); v.fy = (float)a.H / (2.0f * a.tany);
  v.mod = a.mod; v.W = a.W; v.H = a.H;
  v.gx = (a.W + TILE - 1) / TILE; v.gy = (a.H + TILE - 1) / TILE;
  v.deg = a.deg;
}

struct Activated { float p[3], sc[3], q[4], opac, qn[4], inv_n, es[3], sig; };

// activations of one Gaussian (also returns what the backward needs: unit quaternion, 1/norm, exp, sigmoid)
template <bool INFER = false>
__device__ __forceinline__ void activate(const RawFwdArgs& a, int i, Activated& o) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    o.p[k] = a.xyz[3 * i + k] + (a.d_xyz ? a.d_xyz[3 * i + k] : 0.f);
    o.es[k] = __expf(a.scaling[3 * i + k]);
    o.sc[k] = o.es[k] + (a.d_scaling ? a.d_scaling[3 * i + k] : 0.f);
  }
  if constexpr (INFER) {
    if (a.se3) {               // is_6dof: h = M [x, y, z, 1], p = h.xyz / h.w  (utils/rigid_utils.py:128-150; a true division, as there)
      const float4* M = reinterpret_cast<const float4*>(a.se3) + 4 * (size_t)i;
      const float x = a.xyz[3 * i], y = a.xyz[3 * i + 1], z = a.xyz[3 * i + 2];
      const float4 r0 = M[0], r1 = M[1], r2 = M[2], r3 = M[3];
      const float h0 = r0.x * x + r0.y * y + r0.z * z + r0.w, h1 = r1.x * x + r1.y * y + r1.z * z + r1.w;
      const float h2 = r2.x * x + r2.y * y + r2.z * z + r2.w, h3 = r3.x * x + r3.y * y + r3.z * z + r3.w;
      o.p[0] = h0 / h3; o.p[1] = h1 / h3; o.p[2] = h2 / h3;
    }
  }
  const float4 r = reinterpret_cast<const float4*>(a.rotation)[i];
  const float n = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
  o.inv_n = 1.0f / fmaxf(n, 1e-12f);                       // torch.nn.functional.normalize eps
  o.qn[0] = r.x * o.inv_n; o.qn[1] = r.y * o.inv_n; o.qn[2] = r.z * o.inv_n; o.qn[3] = r.w * o.inv_n;
  float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.d_rotation) dq = reinterpret_cast<const float4*>(a.d_rotation)[i];
  o.q[0] = o.qn[0] + dq.x; o.q[1] = o.qn[1] + dq.y; o.q[2] = o.qn[2] + dq.z; o.q[3] = o.qn[3] + dq.w;
  o.sig = 1.0f / (1.0f + __expf(-a.opacity[i]));
  o.opac = o.sig;
}

// ---- wave-cooperative access to the [P][45] f_rest rows ------------------------------------------------------------
// A lane that reads ITS row straight from memory issues twelve 16-byte loads at a 180-byte stride: every instruction
// touches 64 cache lines and the texture-address unit serialises them (measured: TA busy 45 % of the kernel, VALU 20 %).
// Instead the wave copies the 64 consecutive rows (11 520 contiguous bytes) with fully coalesced loads into a wave-private
// LDS slab and each lane reads its row from there (row stride 45 words: odd, conflict-free).
constexpr int REST_W = 45;
constexpr int REST_SLAB = 64 * REST_W;                       // floats per wave
constexpr int REST_Q = REST_SLAB / 4 / 64 + 1;               // 16-byte loads per lane (the last one covers 16 lanes)

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct RestRegs { float4 q[REST_Q]; };

// issue the loads (rows [row0, row0 + 64) of `base`, clipped to P); `aligned16` is wave-uniform
__device__ __forceinline__ void rest_rows_load(const float* __restrict__ base, int row0, int P, bool aligned16, RestRegs& r) {
  const int lane = threadIdx.x & 63;
  const int nfl = min(64, P - row0) * REST_W;                 // valid floats of this wave's block
  const float* src = base + (size_t)row0 * REST_W;
#pragma unroll
  for (int k = 0; k < REST_Q; ++k) {
    const int e = 4 * (k * 64 + lane);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (aligned16 && e + 3 < nfl) v = *reinterpret_cast<const float4*>(src + e);
    else {
      if (e < nfl) v.x = src[e];
      if (e + 1 < nfl) v.y = src[e + 1];
      if (e + 2 < nfl) v.z = src[e + 2];
      if (e + 3 < nfl) v.w = src[e + 3];
    }
    r.q[k] = v;
  }
}

__device__ __forceinline__ void rest_rows_to_lds(const RestRegs& r, float* slab) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < REST_Q; ++k) {
    const int e = 4 * (k * 64 + lane);
    if (e < REST_SLAB) *reinterpret_cast<float4*>(slab + e) = r.q[k];
  }
  wave_lds_sync();
}

// the lane's 16 x 3 coefficients: f_dc from memory (12-byte stride, three lines per instruction), f_rest from the slab
__device__ __forceinline__ void load_sh_split(const RawFwdArgs& a, int i, const float* slab, float shl[48]) {
  const int n3 = 3 * ncoef(a.deg);
  const float* dc = a.f_dc + 3 * (size_t)i;
  shl[0] = dc[0]; shl[1] = dc[1]; shl[2] = dc[2];
  const float* row = slab + (threadIdx.x & 63) * REST_W;
#pragma unroll
  for (int k = 3; k < 48; ++k) shl[k] = (k < n3) ? row[k - 3] : 0.f;
}

// cooperative store of 64 rows held in a slab (the backward's dL/df_rest)
__device__ __forceinline__ void rest_rows_store(float* __restrict__ base, int row0, int P, bool aligned16, const float* slab) {
  const int lane = threadIdx.x & 63;
  const int nfl = min(64, P - row0) * REST_W;
  float* dst = base + (size_t)row0 * REST_W;
#pragma unroll
  for (int k = 0; k < REST_Q; ++k) {
    const int e = 4 * (k * 64 + lane);
    if (e >= REST_SLAB) continue;
    const float4 v = *reinterpret_cast<const float4*>(slab + e);
    if (aligned16 && e + 3 < nfl) *reinterpret_cast<float4*>(dst + e) = v;
    else {
      if (e < nfl) dst[e] = v.x;
      if (e + 1 < nfl) dst[e + 1] = v.y;
      if (e + 2 < nfl) dst[e + 2] = v.z;
      if (e + 3 < nfl) dst[e + 3] = v.w;
    }
  }
}

#ifndef TRASE_RAW_BLOCK
#define TRASE_RAW_BLOCK 64
#endif
constexpr int RAW_BLOCK = TRASE_RAW_BLOCK;                   // waves x 64 per workgroup: 11.5 KB of LDS per wave

// BLOCK: 64 for the whole image; 256 for a tile-row strip -- most waves have nothing to move there and the kernel is bound by
// the rate at which workgroups can be dispatched (~250 per microsecond: 39k one-wave workgroups = 0.15 ms at 2.5 M Gaussians)
// INFER: the instantiation that also serves override_color / mask / is_6dof / convert_SHs_python (RawFwdArgs) -- the training
// instantiation (INFER = false) is compiled without a trace of them
template <int F, int BLOCK, bool INFER = false>
__global__ __launch_bounds__(BLOCK) void preprocess_fwd_raw_kernel(RawFwdArgs a, int32_t* __restrict__ radii,
                                                                 float2* __restrict__ xy, float4* __restrict__ conic_o,
                                                                 float4* __restrict__ rgbd, float4* __restrict__ geo, uint32_t* __restrict__ ftab,
                                                                 uint32_t* __restrict__ tiles,
                                                                 uint32_t* __restrict__ clamped,
                                                                 uint32_t* __restrict__ depth_keys,
                                                                 uint32_t* __restrict__ hdr) {
#ifdef TRASE_RAW_SETPRIO
  __builtin_amdgcn_s_setprio(3);                             // experiment build (profiles/r6_two_streams.md)
#endif
  const int gi = blockIdx.x * blockDim.x + threadIdx.x;
  if (gi == 0) {                                             // clean header (one memset launch less per view) + P for the depth sort
#pragma unroll
    for (int k = 0; k < HDR_WORDS - 1; ++k) hdr[k] = 0u;
    hdr[HDR_WORDS - 1] = (uint32_t)a.P;
  }
  const bool active = gi < a.P;
  const int i = active ? gi : a.P - 1;
  constexpr bool strip = BLOCK != RAW_BLOCK;                 // the 256-thread instantiation IS the tile-row-strip one
#ifdef TRASE_RAW_NO_SLAB
  // experiment build (profiles/r6_two_streams.md): no LDS in this kernel at all -- every lane reads its own
  // f_rest row from memory, as the strip path does
  constexpr bool noslab = true;
#else
  constexpr bool noslab = strip;
#endif
  __shared__ __attribute__((aligned(16))) float slabs[noslab ? 1 : BLOCK / 64][noslab ? 4 : REST_SLAB];   // (the strip path moves no slab)
  float* const slab = slabs[noslab ? 0 : (threadIdx.x >> 6)];
  const int row0 = gi & ~63;                                  // first Gaussian of this wave
  if (row0 >= a.P) return;                                    // (a whole wave past the end: BLOCK > 64 only)
  const bool rest16 = ((reinterpret_cast<uintptr_t>(a.f_rest) & 15) == 0);
  const bool precol = INFER && a.colors != nullptr;          // (wave-uniform) override_color: no SH coefficients are read at all
  RestRegs rr;
  if (!noslab && !precol) rest_rows_load(a.f_rest, row0, a.P, rest16, rr);   // (requested first: the geometry arithmetic hides the trip)
  View v;
  raw_view(a, v);
  Activated act;
  activate<INFER>(a, i, act);
  const bool masked_out = INFER && a.mask != nullptr && a.mask[i] == 0;
  float shl[48];
  const float cv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, col[3] = {0.f, 0.f, 0.f};
  Splat o;
  // Geometry and colour are two call sites that BOTH modes go through (whole image / tile-row strip): a strip must reproduce
  // the full render bit for bit, and two inlined copies of the same arithmetic are not guaranteed to contract alike (measured:
  // a second copy of splat_forward differed by one ulp in 15 % of the conics).  What differs between the modes is only WHICH
  // Gaussians get a colour and a record: every visible one, or -- strip -- those with a pair in the strip (~1 / world of them;
  // the 192 bytes of SH coefficients are 1/3 of what a Gaussian moves through this kernel).
  bool vis = splat_forward<false, false>(v, act.p, act.sc, act.q, cv, shl, col, o) && active && !masked_out;
  if (active) radii[i] = vis ? o.radius : 0;
  uint32_t live = 0;
  if (vis) {
    const SubtileCull cull = subtile_cull_setup(o.px, o.py, o.ca, o.cb, o.cc, act.opac);
    // per sub-tile row the live columns are an interval (subtile_row_live): O(rows) per splat
    for (int sy = max(2 * o.y0, a.sy_lo); sy < min(2 * o.y1, a.sy_hi) && sy * SUB < a.H; ++sy) {
      int c0, c1;
      subtile_row_live(cull, sy, a.W, a.H, 2 * o.x0, 2 * o.x1, c0, c1);
      live += (uint32_t)(c1 - c0);
    }
  }
  if (active) {
    tiles[i] = live;
    depth_keys[i] = (vis && live) ? (depth_sort_key(o.depth, a.key27 != 0) | a.key_or) : a.key_dead;
  }
  const bool want = vis && (live != 0 || !strip);             // gets a colour and a record
  if (!noslab && !precol) {
    rest_rows_to_lds(rr, slab);
  }
  if (strip) {
    if (vis && !live) xy[i] = make_float2(o.px, o.py);        // (the lineage pair count R is totalled from radii + centres)
  }
  if (want && precol) {
    o.rgb[0] = a.colors[3 * (size_t)i]; o.rgb[1] = a.colors[3 * (size_t)i + 1]; o.rgb[2] = a.colors[3 * (size_t)i + 2];
    o.clamped = 0u;
  } else if (want) {
    if (!noslab) {
      load_sh_split(a, i, slab, shl);
    } else {
      // the few lanes that have a pair read their own rows (all 45 loads in flight at once; a cooperative copy of the rows one
      // by one is a memory round trip per row, the slab copy moves eight times what is needed)
      const int n3 = 3 * ncoef(a.deg);
      const float* dc = a.f_dc + 3 * (size_t)i;
      const float* rw = a.f_rest + (size_t)i * REST_W;
      shl[0] = dc[0]; shl[1] = dc[1]; shl[2] = dc[2];
#pragma unroll
      for (int k = 3; k < 48; ++k) shl[k] = (k < n3) ? rw[k - 3] : 0.f;
    }
    if (INFER && a.sh_dir_raw) {   // convert_SHs_python evaluates the SH at the UNDEFORMED position (gaussian_renderer/__init__.py:105)
      const float p0[3] = {a.xyz[3 * (size_t)i], a.xyz[3 * (size_t)i + 1], a.xyz[3 * (size_t)i + 2]};
      splat_colour_sh(v, p0, shl, o);
    } else {
      splat_colour_sh(v, act.p, shl, o);
    }
  }
  vis = want;
  if (vis) {
    xy[i] = make_float2(o.px, o.py);
    conic_o[i] = make_float4(o.ca, o.cb, o.cc, act.opac);
    if (!(INFER && a.fwd_only)) rgbd[i] = make_float4(o.rgb[0], o.rgb[1], o.rgb[2], o.depth);   // (read by the backward and the VALU forward only)
    // the same 40 bytes as ONE 64-byte record: the compositing kernels fetch a list entry's geometry from one cache line
    // (word 2 of the record: the Gaussian's first row slot, written by emit_pairs)
    geo[4 * (size_t)i + 0] = make_float4(o.px, o.py, 0.f, __int_as_float(o.radius));   // .w: the radius, for emit_pairs
    geo[4 * (size_t)i + 1] = make_float4(o.ca, o.cb, o.cc, act.opac);
    geo[4 * (size_t)i + 2] = make_float4(o.rgb[0], o.rgb[1], o.rgb[2], o.depth);
    geo[4 * (size_t)i + 3] = split_rgbd(o.rgb[0], o.rgb[1], o.rgb[2], o.depth);
    if (!(INFER && a.fwd_only)) clamped[i] = o.clamped;
  }
  // feature rows the compositing kernels read: f / (||f|| + 1e-9)  (gaussian_renderer/__init__.py:120-121).  F / 4 lanes per
  // Gaussian, one float4 each: the wave's 64 rows are read and written as contiguous 1 KB runs; the squared norm is summed
  // over the lane group with DPP (quad swaps + half-row mirror).
  if constexpr (F > 0) {
    constexpr int LPG = F / 4, GPI = 64 / LPG;               // lanes per Gaussian, Gaussians per wave-wide access
    static_assert(LPG == 4 || LPG == 8, "feature width 16 or 32");
    const unsigned long long need = __ballot(active && vis && live);
    const int lane = threadIdx.x & 63;
    float4 r[LPG];
    bool on[LPG];
#pragma unroll
    for (int k = 0; k < LPG; ++k) {
      const int gl = k * GPI + lane / LPG;
      on[k] = (need >> gl) & 1ull;
      r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (on[k]) r[k] = reinterpret_cast<const float4*>(a.features + (size_t)(row0 + gl) * F)[lane % LPG];
    }
#pragma unroll
    for (int k = 0; k < LPG; ++k) {
      float n2 = r[k].x * r[k].x + r[k].y * r[k].y + r[k].z * r[k].z + r[k].w * r[k].w;
      n2 += dpp_f<0xB1>(n2);                                  // quad_perm [1,0,3,2]
      n2 += dpp_f<0x4E>(n2);                                  // quad_perm [2,3,0,1]
      if (LPG == 8) n2 += dpp_f<0x141>(n2);                   // row_half_mirror: the other quad of the 8-lane group
      const float sc = a.norm_features ? 1.0f / (sqrtf(n2) + 1e-9f) : 1.0f;
      const int gl = k * GPI + lane / LPG;
      if (on[k]) {
        const float4 v = make_float4(r[k].x * sc, r[k].y * sc, r[k].z * sc, r[k].w * sc);
        if (F != 32 || a.write_featn) reinterpret_cast<float4*>(a.featn + (size_t)(row0 + gl) * F)[lane % LPG] = v;
        if constexpr (F == 32) {                              // the same row as bf16 [hi 32 | lo 32] (GeomBuf::ftab)
          unsigned h01, l01, h23, l23;
          split_pk(v.x, v.y, h01, l01);
          split_pk(v.z, v.w, h23, l23);
          uint32_t* const t = ftab + (size_t)(row0 + gl) * 32 + 2 * (lane % LPG);
          *reinterpret_cast<uint2*>(t) = make_uint2(h01, h23);
          *reinterpret_cast<uint2*>(t + 16) = make_uint2(l01, l23);
        }
      }
    }
  }
}

int launch_preprocess_fwd_raw(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastRawInputs& raw, int32_t* radii,
                              const GeomBuf& g, uint32_t* depth_keys, bool key27, uint32_t key_or, uint32_t key_dead) {
  RawFwdArgs a;
  a.key27 = key27 ? 1 : 0; a.key_or = key_or; a.key_dead = key_dead;
  a.xyz = raw.xyz; a.d_xyz = raw.d_xyz; a.f_dc = raw.features_dc; a.f_rest = raw.features_rest; a.opacity = raw.opacity;
  a.scaling = raw.scaling; a.d_scaling = raw.d_scaling; a.rotation = raw.rotation; a.d_rotation = raw.d_rotation;
  a.features = raw.gaussian_features; a.featn = raw.featn;
  // the fp32 normalised rows are read by the VALU forward and the VALU backward only
  a.write_featn = (c.variant & (TRASE_VARIANT_VALU_BACKWARD | TRASE_VARIANT_VALU_FORWARD)) != 0;
  a.vm = s.viewmatrix; a.pm = s.projmatrix; a.cam = s.campos;
  a.P = raw.P; a.F = raw.F; a.deg = s.sh_degree; a.W = s.image_width; a.H = s.image_height;
  a.norm_features = raw.norm_features; a.tanx = s.tanfovx; a.tany = s.tanfovy; a.mod = s.scale_modifier;
  strip_subtile_rows(s, a.sy_lo, a.sy_hi);
  a.strip = (s.tile_row_begin != 0 || s.tile_row_end != 0) ? 1 : 0;
  a.p_begin = 0; a.p_end = raw.P;
  a.colors = raw.colors_precomp; a.mask = raw.mask; a.se3 = raw.d_xyz_se3; a.sh_dir_raw = raw.sh_dir_undeformed;
  // (TRASE_VARIANT_FORWARD_ONLY is honoured by the compositing kernel only: here it would have to select another instantiation
  // than the training forward's, and two compilations of the same arithmetic are not bit-identical -- a no_grad render must equal
  // the training render of the same view; the 20 bytes per Gaussian at stake are 7 % of this kernel's traffic)
  a.fwd_only = 0;
  const bool infer = a.colors || a.mask || a.se3 || a.sh_dir_raw;
  const int blk = a.strip ? 256 : RAW_BLOCK;
  const dim3 grid((raw.P + blk - 1) / blk), block(blk);
  {
    ProfScope ps("preprocess_fwd", c.stream);
#define TRASE_PRF2(FF, BB, II) hipLaunchKernelGGL((preprocess_fwd_raw_kernel<FF, BB, II>), grid, block, 0, c.stream, a, radii, g.xy, g.conic_o, g.rgbd, g.geo, g.ftab, g.tiles, g.clamped, depth_keys, g.hdr)
#define TRASE_PRF(FF) do { if (a.strip) { if (infer) TRASE_PRF2(FF, 256, true); else TRASE_PRF2(FF, 256, false); } \
                           else { if (infer) TRASE_PRF2(FF, RAW_BLOCK, true); else TRASE_PRF2(FF, RAW_BLOCK, false); } } while (0)
    switch (raw.F) {
      case 0: TRASE_PRF(0); break;
      case 16: TRASE_PRF(16); break;
      case 32: TRASE_PRF(32); break;
      default: set_error("preprocess_fwd_raw: feature width %d not compiled in (0,16,32)", raw.F); return TRASE_ERR_UNSUPPORTED;
    }
#undef TRASE_PRF
#undef TRASE_PRF2
  }
  TRASE_POST_LAUNCH("preprocess_fwd", c.stream, c.debug);
  return TRASE_OK;
}

// ---- backward -----------------------------------------------------------------------------------------
struct RawBwdOut {
  float* d_xyz; float* d_dxyz; float* d_means2D; float* d_f_dc; float* d_f_rest; float* d_opacity;
  float* d_scaling; float* d_dscaling; float* d_rotation; float* d_drotation;
  float* d_colors;           // INFER: dL/doverride_color (P,3)
  float* d_se3;              // INFER: dL/d(the (P,4,4) is_6dof transforms)
  float* d_feat_zero;        // tile-row strips: dL/dgaussian_features rows of the Gaussians WITHOUT a pair are zeroed here
  int F;                     // (reduce_rows only walked the ones that have one), F floats per row
};

// LIVE (tile-row strips with sparse gradients, TRASE_VARIANT_SPARSE_STRIP_GRADS): thread k handles the k-th Gaussian of the
// live list `ids` (those with a pair in the strip, hdr[HDR_WORDS - 1] of them) and ONLY their rows of the gradient tensors are
// written -- the caller keeps every other row zero (trase_rast_zero_live_rows).  The f_rest rows are gathered / scattered row by
// row (one coalesced 180-byte access each) instead of moved as the wave's contiguous slab.
// the SH block of splat_backward (gs_math.h) on its own, for the call pattern whose SH direction comes from another position than
// the splat's (pipe.convert_SHs_python: the undeformed xyz): dL/dsh and the direction's gradient w.r.t. THAT position
__device__ __forceinline__ void sh_backward_at(const View& v, const float p_sh[3], const float* sh, unsigned clamped, const float d_rgb[3],
                                               float* d_sh, float d_p_sh[3]) {
  float dr[3];
  for (int ch = 0; ch < 3; ++ch) dr[ch] = ((clamped >> ch) & 1u) ? 0.f : d_rgb[ch];
  float d[3], il, bb[16], gx[16], gy[16], gz[16];
  sh_dir(p_sh, v.cam, d, il);
  sh_basis(v.deg, d, bb);
  sh_basis_grad(v.deg, d, gx, gy, gz);
  float dd[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float s0 = sh[3 * k], s1 = sh[3 * k + 1], s2 = sh[3 * k + 2];
    d_sh[3 * k] = bb[k] * dr[0]; d_sh[3 * k + 1] = bb[k] * dr[1]; d_sh[3 * k + 2] = bb[k] * dr[2];
    const float w = s0 * dr[0] + s1 * dr[1] + s2 * dr[2];
    dd[0] += gx[k] * w; dd[1] += gy[k] * w; dd[2] += gz[k] * w;
  }
  const float dot = d[0] * dd[0] + d[1] * dd[1] + d[2] * dd[2];
  for (int k = 0; k < 3; ++k) d_p_sh[k] = (dd[k] - d[k] * dot) * il;
}

template <bool LIVE, int BLOCK, bool INFER = false>
__global__ __launch_bounds__(BLOCK) void preprocess_bwd_raw_kernel(RawFwdArgs a, const int32_t* __restrict__ radii,
                                                                   const uint32_t* __restrict__ clamped,
                                                                   const uint32_t* __restrict__ tiles,
                                                                   const float* __restrict__ acc, RawBwdOut o,
                                                                   const uint32_t* __restrict__ ids,
                                                                   const uint32_t* __restrict__ n_live_ptr) {
  const int gidx = a.p_begin + blockIdx.x * blockDim.x + threadIdx.x;
  const int p_end = LIVE ? min(a.P, (int)*n_live_ptr) : a.p_end;
  if (LIVE && (gidx & ~63) >= p_end) return;                        // a whole wave behind the live list
  const bool active = gidx < p_end;                                 // no early return inside a wave: it moves the f_rest rows together
  const int i = LIVE ? (active ? (int)ids[gidx] : 0) : (active ? gidx : a.P - 1);
  // (no pair -- culled, fainter than 1/255 everywhere, or outside the strip being rendered -- means exact zero gradients)
  const bool vis = active && radii[i] > 0 && tiles[i] > 0;
  __shared__ __attribute__((aligned(16))) float slabs[BLOCK / 64][REST_SLAB];
  float* const slab = slabs[threadIdx.x >> 6];
  const int row0 = gidx & ~63;
  const bool wave_vis = __ballot(vis) != 0ull;
  const bool precol = INFER && a.colors != nullptr;          // (wave-uniform) override_color: no SH anywhere
  float d_p_raw[3] = {0.f, 0.f, 0.f};                          // INFER: the part of dL/dposition that belongs to the UNDEFORMED xyz only
  if (LIVE) {
    // (every lane reads and writes its own 180-byte rows: the ids are scattered, there is no slab to move)
  } else if (wave_vis && !precol) {
    RestRegs rr;
    rest_rows_load(a.f_rest, row0, a.P, (reinterpret_cast<uintptr_t>(a.f_rest) & 15) == 0, rr);
    rest_rows_to_lds(rr, slab);
  }
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
  Activated act;
  act.inv_n = 0.f; act.sig = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) act.es[k] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) act.qn[k] = 0.f;
  if (vis) {
    const float4* r4 = reinterpret_cast<const float4*>(acc + (size_t)i * BWD_ACC);
    const float4 r0 = r4[0], r1 = r4[1], r2 = r4[2];
    gi.d_ndcx = r0.x; gi.d_ndcy = r0.y; gi.d_ca = r0.z; gi.d_cb = r0.w;
    gi.d_cc = r1.x; d_op = r1.y; gi.d_rgb[0] = r1.z; gi.d_rgb[1] = r1.w;
    gi.d_rgb[2] = r2.x; gi.d_depth = r2.y;
    View v;
    raw_view(a, v);
    activate<INFER>(a, i, act);
    float shl[48];
    if (precol) {
      // (no coefficients)
    } else if (LIVE) {
      const int n3 = 3 * ncoef(a.deg);
      const float* dc = a.f_dc + 3 * (size_t)i;
      const float* rw = a.f_rest + (size_t)i * REST_W;
      shl[0] = dc[0]; shl[1] = dc[1]; shl[2] = dc[2];
#pragma unroll
      for (int k = 3; k < 48; ++k) shl[k] = (k < n3) ? rw[k - 3] : 0.f;
    } else {
      load_sh_split(a, i, slab, shl);
    }
    const float cv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (precol) {
      splat_backward<false, false>(v, act.p, act.sc, act.q, cv, shl, 0u, gi, go, dsh);     // the colour is an input: its gradient is d_rgb itself
    } else if (INFER && a.sh_dir_raw) {
      const float p0[3] = {a.xyz[3 * (size_t)i], a.xyz[3 * (size_t)i + 1], a.xyz[3 * (size_t)i + 2]};
      splat_backward<false, false>(v, act.p, act.sc, act.q, cv, shl, 0u, gi, go, dsh);
      sh_backward_at(v, p0, shl, clamped[i], gi.d_rgb, dsh, d_p_raw);
    } else {
      splat_backward<false, true>(v, act.p, act.sc, act.q, cv, shl, clamped[i], gi, go, dsh);
    }
  }
  // chain through the activations
  if (active) {
  if (INFER && a.se3) {
    // p = h.xyz / h.w, h = M [x, y, z, 1]: dL/dh = (g / h.w, -<g, p> / h.w); dL/dM = dL/dh [x, y, z, 1]^T; dL/dxyz = M[:, :3]^T dL/dh
    float dx[3] = {0.f, 0.f, 0.f}, dM[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) dM[k] = 0.f;
    if (vis) {
      const float4* M = reinterpret_cast<const float4*>(a.se3) + 4 * (size_t)i;
      const float x = a.xyz[3 * i], y = a.xyz[3 * i + 1], z = a.xyz[3 * i + 2];
      const float4 r0 = M[0], r1 = M[1], r2 = M[2], r3 = M[3];
      const float h3 = r3.x * x + r3.y * y + r3.z * z + r3.w;
      const float inv = 1.0f / h3;
      const float dh[4] = {go.d_p[0] * inv, go.d_p[1] * inv, go.d_p[2] * inv,
                           -(go.d_p[0] * act.p[0] + go.d_p[1] * act.p[1] + go.d_p[2] * act.p[2]) * inv};
      const float ph[4] = {x, y, z, 1.0f};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) dM[4 * r + cc] = dh[r] * ph[cc];
      dx[0] = dh[0] * r0.x + dh[1] * r1.x + dh[2] * r2.x + dh[3] * r3.x;
      dx[1] = dh[0] * r0.y + dh[1] * r1.y + dh[2] * r2.y + dh[3] * r3.y;
      dx[2] = dh[0] * r0.z + dh[1] * r1.z + dh[2] * r2.z + dh[3] * r3.z;
    }
    if (o.d_xyz) { o.d_xyz[3 * i] = dx[0] + d_p_raw[0]; o.d_xyz[3 * i + 1] = dx[1] + d_p_raw[1]; o.d_xyz[3 * i + 2] = dx[2] + d_p_raw[2]; }
    if (o.d_se3) {
      float4* D = reinterpret_cast<float4*>(o.d_se3) + 4 * (size_t)i;
#pragma unroll
      for (int r = 0; r < 4; ++r) D[r] = make_float4(dM[4 * r], dM[4 * r + 1], dM[4 * r + 2], dM[4 * r + 3]);
    }
  } else {
  if (o.d_xyz) {
    if constexpr (INFER) { o.d_xyz[3 * i] = go.d_p[0] + d_p_raw[0]; o.d_xyz[3 * i + 1] = go.d_p[1] + d_p_raw[1]; o.d_xyz[3 * i + 2] = go.d_p[2] + d_p_raw[2]; }
    else { o.d_xyz[3 * i] = go.d_p[0]; o.d_xyz[3 * i + 1] = go.d_p[1]; o.d_xyz[3 * i + 2] = go.d_p[2]; }
  }
  if (o.d_dxyz) { o.d_dxyz[3 * i] = go.d_p[0]; o.d_dxyz[3 * i + 1] = go.d_p[1]; o.d_dxyz[3 * i + 2] = go.d_p[2]; }
  }
  if (INFER && o.d_colors) { o.d_colors[3 * (size_t)i] = gi.d_rgb[0]; o.d_colors[3 * (size_t)i + 1] = gi.d_rgb[1]; o.d_colors[3 * (size_t)i + 2] = gi.d_rgb[2]; }
  if (o.d_means2D) { o.d_means2D[3 * i] = gi.d_ndcx; o.d_means2D[3 * i + 1] = gi.d_ndcy; o.d_means2D[3 * i + 2] = 0.f; }
  if (o.d_opacity) o.d_opacity[i] = d_op * act.sig * (1.0f - act.sig);
  if (o.d_scaling) {
    o.d_scaling[3 * i] = go.d_scale[0] * act.es[0]; o.d_scaling[3 * i + 1] = go.d_scale[1] * act.es[1];
    o.d_scaling[3 * i + 2] = go.d_scale[2] * act.es[2];
  }
  if (o.d_dscaling) { o.d_dscaling[3 * i] = go.d_scale[0]; o.d_dscaling[3 * i + 1] = go.d_scale[1]; o.d_dscaling[3 * i + 2] = go.d_scale[2]; }
  if (o.d_drotation) reinterpret_cast<float4*>(o.d_drotation)[i] = make_float4(go.d_quat[0], go.d_quat[1], go.d_quat[2], go.d_quat[3]);
  if (o.d_rotation) {
    // y = q / ||q||  =>  dq = (g - y <y,g>) / ||q||
    const float dot = act.qn[0] * go.d_quat[0] + act.qn[1] * go.d_quat[1] + act.qn[2] * go.d_quat[2] + act.qn[3] * go.d_quat[3];
    reinterpret_cast<float4*>(o.d_rotation)[i] =
        make_float4((go.d_quat[0] - act.qn[0] * dot) * act.inv_n, (go.d_quat[1] - act.qn[1] * dot) * act.inv_n,
                    (go.d_quat[2] - act.qn[2] * dot) * act.inv_n, (go.d_quat[3] - act.qn[3] * dot) * act.inv_n);
  }
  if (o.d_f_dc && !precol) { o.d_f_dc[3 * i] = dsh[0]; o.d_f_dc[3 * i + 1] = dsh[1]; o.d_f_dc[3 * i + 2] = dsh[2]; }
  }
  if (o.d_feat_zero) {
    // eight lanes per 128-byte row (F = 32; four for F = 16): the wave's rows as contiguous runs
    const int lpg = o.F / 4, gpi = 64 / lpg;
    const unsigned long long dead = __ballot(active && !vis);
    const int lane = threadIdx.x & 63;
    for (int k = 0; k < lpg; ++k) {
      const int gl = k * gpi + lane / lpg;
      if ((dead >> gl) & 1ull) reinterpret_cast<float4*>(o.d_feat_zero + (size_t)(row0 + gl) * o.F)[lane % lpg] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (precol) return;                                      // (override_color: the caller passes no SH gradient tensors)
  if (LIVE) {
    if (o.d_f_rest && active) {
      float* rw = o.d_f_rest + (size_t)i * REST_W;
#pragma unroll
      for (int k = 0; k < REST_W; ++k) rw[k] = dsh[3 + k];
    }
    return;
  }
  if (o.d_f_rest) {
    // 180-byte rows: through the wave's slab (each lane rewrites ITS row, already consumed above), then contiguous stores
    float* row = slab + (threadIdx.x & 63) * REST_W;
#pragma unroll
    for (int k = 0; k < REST_W; ++k) row[k] = dsh[3 + k];
    wave_lds_sync();
    rest_rows_store(o.d_f_rest, row0, a.p_end, (reinterpret_cast<uintptr_t>(o.d_f_rest) & 15) == 0, slab);
  }
}

// rows of the live Gaussians of a PREVIOUS strip backward back to zero (sparse strip gradients keep every other row zero)
struct ZeroRowsArgs { float* t[12]; int w[12]; int n; };
__global__ __launch_bounds__(256) void zero_live_rows_kernel(ZeroRowsArgs z, const uint32_t* __restrict__ ids,
                                                             const uint32_t* __restrict__ n_live_ptr, int P) {
  const int n_live = min(P, (int)*n_live_ptr);
  // 16 lanes per Gaussian: a row of up to 45 floats is cleared by one (or three) partial-wave stores.  A fixed grid strides over
  // the list: the count lives on the device, and workgroups that only find out that there is nothing to do still cost their dispatch
  const int t = threadIdx.x & 15;
  for (int k = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; k < n_live; k += (gridDim.x * blockDim.x) >> 4) {
    const size_t id = ids[k];
    for (int j = 0; j < z.n; ++j)
      for (int c = t; c < z.w[j]; c += 16) z.t[j][id * z.w[j] + c] = 0.f;
  }
}

int launch_zero_live_rows(const LaunchCtx& c, const GeomBuf& g, const PreBuf& pre, int P, int F, const TraseRastRawGrads& gr) {
  ZeroRowsArgs z;
  z.n = 0;
  auto add = [&](float* p, int w) { if (p && w > 0) { z.t[z.n] = p; z.w[z.n] = w; ++z.n; } };
  add(gr.dL_dxyz, 3); add(gr.dL_dd_xyz, 3); add(gr.dL_dmeans2D, 3); add(gr.dL_dfeatures_dc, 3); add(gr.dL_dfeatures_rest, 45);
  add(gr.dL_dopacity, 1); add(gr.dL_dscaling, 3); add(gr.dL_dd_scaling, 3); add(gr.dL_drotation, 4); add(gr.dL_dd_rotation, 4);
  add(gr.dL_dgaussian_features, F);
  if (z.n == 0 || P == 0) return TRASE_OK;
  {
    ProfScope ps("zero_live_rows", c.stream);
    const size_t nb = ((size_t)P * 16 + 255) / 256;
    hipLaunchKernelGGL(zero_live_rows_kernel, dim3((int)(nb < 4096 ? nb : 4096)), dim3(256), 0, c.stream, z, pre.live_ids,
                       g.hdr + (HDR_WORDS - 1), P);
  }
  TRASE_POST_LAUNCH("zero_live_rows", c.stream, c.debug);
  return TRASE_OK;
}

int launch_preprocess_bwd_raw(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastRawInputs& raw,
                              const int32_t* radii, const GeomBuf& g, const float* acc, const TraseRastRawGrads& gr,
                              int p_begin, int p_end, int zero_dead_feats, const uint32_t* live_ids) {
  if (p_end < 0) p_end = raw.P;
  if (p_end <= p_begin) return TRASE_OK;
  RawFwdArgs a;
  a.key27 = 0; a.key_or = 0u; a.key_dead = 0u;
  a.write_featn = 0;
  a.p_begin = p_begin; a.p_end = p_end;
  a.xyz = raw.xyz; a.d_xyz = raw.d_xyz; a.f_dc = raw.features_dc; a.f_rest = raw.features_rest; a.opacity = raw.opacity;
  a.scaling = raw.scaling; a.d_scaling = raw.d_scaling; a.rotation = raw.rotation; a.d_rotation = raw.d_rotation;
  a.features = raw.gaussian_features; a.featn = raw.featn;
  a.vm = s.viewmatrix; a.pm = s.projmatrix; a.cam = s.campos;
  a.P = raw.P; a.F = raw.F; a.deg = s.sh_degree; a.W = s.image_width; a.H = s.image_height;
  a.norm_features = raw.norm_features; a.tanx = s.tanfovx; a.tany = s.tanfovy; a.mod = s.scale_modifier;
  strip_subtile_rows(s, a.sy_lo, a.sy_hi);
  RawBwdOut o;
  o.d_xyz = gr.dL_dxyz; o.d_dxyz = gr.dL_dd_xyz; o.d_means2D = gr.dL_dmeans2D; o.d_f_dc = gr.dL_dfeatures_dc;
  o.d_f_rest = gr.dL_dfeatures_rest; o.d_opacity = gr.dL_dopacity; o.d_scaling = gr.dL_dscaling;
  o.d_dscaling = gr.dL_dd_scaling; o.d_rotation = gr.dL_drotation; o.d_drotation = gr.dL_dd_rotation;
  o.d_feat_zero = (zero_dead_feats && (raw.F == 16 || raw.F == 32)) ? gr.dL_dgaussian_features : nullptr; o.F = raw.F;
  o.d_colors = gr.dL_dcolors_precomp; o.d_se3 = gr.dL_dd_xyz_se3;
  a.colors = raw.colors_precomp; a.mask = raw.mask; a.se3 = raw.d_xyz_se3; a.sh_dir_raw = raw.sh_dir_undeformed; a.fwd_only = 0;
  const bool infer = a.colors || a.se3 || a.sh_dir_raw;       // (a mask needs nothing here: a masked-out Gaussian has radii == 0)
  a.strip = 0;
  {
    ProfScope ps("preprocess_bwd", c.stream);
    if (live_ids) {   // sparse strip gradients: the live list only (grid for P -- the count lives on the device; whole waves behind it leave at once)
      if (infer) { set_error("preprocess_bwd_raw: sparse strip gradients are a training path (no override_color / is_6dof / convert_SHs_python)"); return TRASE_ERR_UNSUPPORTED; }
      hipLaunchKernelGGL((preprocess_bwd_raw_kernel<true, 256>), dim3((raw.P + 255) / 256), dim3(256), 0, c.stream, a, radii, g.clamped, g.tiles,
                         acc, o, live_ids, g.hdr + (HDR_WORDS - 1));
    } else if (infer)
      hipLaunchKernelGGL((preprocess_bwd_raw_kernel<false, RAW_BLOCK, true>), dim3((p_end - p_begin + RAW_BLOCK - 1) / RAW_BLOCK), dim3(RAW_BLOCK), 0,
                         c.stream, a, radii, g.clamped, g.tiles, acc, o, nullptr, nullptr);
    else
      hipLaunchKernelGGL((preprocess_bwd_raw_kernel<false, RAW_BLOCK>), dim3((p_end - p_begin + RAW_BLOCK - 1) / RAW_BLOCK), dim3(RAW_BLOCK), 0,
                         c.stream, a, radii, g.clamped, g.tiles, acc, o, nullptr, nullptr);
  }
  TRASE_POST_LAUNCH("preprocess_bwd", c.stream, c.debug);
  return TRASE_OK;
}

}  // namespace trase
