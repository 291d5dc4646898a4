// gs_math.h -- per-Gaussian arithmetic of the rasterizer path (projection, EWA
// covariance, conic, SH colour) and its hand-derived backward.
//
// Pure scalar float32 functions, usable from HIP device code (the product) and,
// compiled with g++, from tests/hostsim (so the derivatives can be checked
// against the float64 autograd oracle without a GPU).
//
// Semantics follow SURVEY.md Appendix A (public 3DGS lineage); reference
// call-site contract: gaussian_renderer/__init__.py:58-73,137-146.  Python
// fallbacks of two sub-steps exist in the reference and pin the conventions:
// utils/sh_utils.py:57-112 (SH), utils/general_utils.py:122-154 (quaternion ->
// rotation), scene/cameras.py:76-79 (transposed matrices).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define TRASE_HD __host__ __device__ __forceinline__
#define TRASE_HD_NOINLINE inline __host__ __device__ __noinline__
#define TRASE_UNROLL _Pragma("unroll")
#else
#define TRASE_HD inline
#define TRASE_HD_NOINLINE inline
#define TRASE_UNROLL
#endif

namespace trase {

TRASE_HD int imin(int a, int b) { return a < b ? a : b; }
TRASE_HD int imax(int a, int b) { return a > b ? a : b; }

constexpr int TILE = 16;            // tile edge that defines rect membership (visible semantics)
constexpr float NEAR_Z = 0.2f;      // cull if view z <= NEAR_Z
constexpr float LOWPASS = 0.3f;     // px^2 added to the 2D covariance diagonal
constexpr float ALPHA_MAX = 0.99f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_STOP = 1e-4f;

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f,
                SH_C2_2 = 0.31539156525252005f, SH_C2_3 = -1.0925484305920792f,
                SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f,
                SH_C3_2 = -0.4570457994644658f, SH_C3_3 = 0.3731763325901154f,
                SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                SH_C3_6 = -0.5900435899266435f;

// Camera constants of one view; matrices flat as stored by the reference
// (transposed / row-vector form): math M[row][col] == flat[4*col+row].
struct View {
  float V[16];
  float PM[16];
  float cam[3];
  float tanx, tany, fx, fy;
  float mod;      // scale_modifier
  int W, H, gx, gy;
  int deg;        // active SH degree
};

struct Splat {    // forward state of one Gaussian
  float px, py;   // pixel-space centre
  float depth;    // view-space z
  float ca, cb, cc;  // conic (inverse 2D covariance)
  float rgb[3];
  int radius;     // 0 == culled
  int x0, y0, x1, y1;  // tile rect, half open
  unsigned clamped;    // bit c set: colour channel c was clamped at 0
};

TRASE_HD void quat_to_rot(const float q[4], float R[9]) {
  const float r = q[0], x = q[1], y = q[2], z = q[3];   // used as given, NOT normalised
  R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
  R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
  R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = R S S R^T, upper triangle (xx,xy,xz,yy,yz,zz)
TRASE_HD void cov3d_from_scale_rot(const float s[3], float mod, const float q[4], float cov[6]) {
  float R[9];
  quat_to_rot(q, R);
  float L[9];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) L[3 * i + k] = R[3 * i + k] * (mod * s[k]);
  cov[0] = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
  cov[1] = L[0] * L[3] + L[1] * L[4] + L[2] * L[5];
  cov[2] = L[0] * L[6] + L[1] * L[7] + L[2] * L[8];
  cov[3] = L[3] * L[3] + L[4] * L[4] + L[5] * L[5];
  cov[4] = L[3] * L[6] + L[4] * L[7] + L[5] * L[8];
  cov[5] = L[6] * L[6] + L[7] * L[7] + L[8] * L[8];
}

struct Ewa {      // intermediates of the 2D covariance, shared by forward and backward
  float t[3];     // view-space position
  float txc, tyc; // clamped t.x, t.y used inside the Jacobian
  bool clx, cly;  // clamp active
  float A0[3], A1[3];   // rows of J*W
  float SA0[3], SA1[3]; // Sigma*A0, Sigma*A1
  float a, b, c;        // cov2D incl. low-pass
};

TRASE_HD void ewa_forward(const View& v, const float p[3], const float cov[6], Ewa& e) {
  const float* V = v.V;
  e.t[0] = V[0] * p[0] + V[4] * p[1] + V[8] * p[2] + V[12];
  e.t[1] = V[1] * p[0] + V[5] * p[1] + V[9] * p[2] + V[13];
  e.t[2] = V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14];
  const float tz = e.t[2];
  const float limx = 1.3f * v.tanx, limy = 1.3f * v.tany;
  const float txtz = e.t[0] / tz, tytz = e.t[1] / tz;
  e.clx = (txtz < -limx) || (txtz > limx);
  e.cly = (tytz < -limy) || (tytz > limy);
  e.txc = fminf(limx, fmaxf(-limx, txtz)) * tz;
  e.tyc = fminf(limy, fmaxf(-limy, tytz)) * tz;
  const float j00 = v.fx / tz, j02 = -(v.fx * e.txc) / (tz * tz);
  const float j11 = v.fy / tz, j12 = -(v.fy * e.tyc) / (tz * tz);
  // rows of the math rotation W (t = W p + trans): W_r[k] = V[4k + r]
  for (int k = 0; k < 3; ++k) {
    e.A0[k] = j00 * V[4 * k + 0] + j02 * V[4 * k + 2];
    e.A1[k] = j11 * V[4 * k + 1] + j12 * V[4 * k + 2];
  }
  const float S[9] = {cov[0], cov[1], cov[2], cov[1], cov[3], cov[4], cov[2], cov[4], cov[5]};
  for (int i = 0; i < 3; ++i) {
    e.SA0[i] = S[3 * i] * e.A0[0] + S[3 * i + 1] * e.A0[1] + S[3 * i + 2] * e.A0[2];
    e.SA1[i] = S[3 * i] * e.A1[0] + S[3 * i + 1] * e.A1[1] + S[3 * i + 2] * e.A1[2];
  }
  e.a = e.A0[0] * e.SA0[0] + e.A0[1] * e.SA0[1] + e.A0[2] * e.SA0[2] + LOWPASS;
  e.b = e.A0[0] * e.SA1[0] + e.A0[1] * e.SA1[1] + e.A0[2] * e.SA1[2];
  e.c = e.A1[0] * e.SA1[0] + e.A1[1] * e.SA1[1] + e.A1[2] * e.SA1[2] + LOWPASS;
}

struct ShBasis { float b[16]; };

TRASE_HD void sh_dir(const float p[3], const float cam[3], float d[3], float& inv_len) {
  const float vx = p[0] - cam[0], vy = p[1] - cam[1], vz = p[2] - cam[2];
  inv_len = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
  d[0] = vx * inv_len; d[1] = vy * inv_len; d[2] = vz * inv_len;
}

TRASE_HD void sh_basis(int deg, const float d[3], float b[16]) {
  const float x = d[0], y = d[1], z = d[2];
  TRASE_UNROLL
  for (int k = 1; k < 16; ++k) b[k] = 0.f;
  b[0] = SH_C0;
  if (deg > 0) {
    b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      b[4] = SH_C2_0 * xy; b[5] = SH_C2_1 * yz; b[6] = SH_C2_2 * (2.f * zz - xx - yy);
      b[7] = SH_C2_3 * xz; b[8] = SH_C2_4 * (xx - yy);
      if (deg > 2) {
        b[9] = SH_C3_0 * y * (3.f * xx - yy); b[10] = SH_C3_1 * xy * z;
        b[11] = SH_C3_2 * y * (4.f * zz - xx - yy); b[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
        b[13] = SH_C3_4 * x * (4.f * zz - xx - yy); b[14] = SH_C3_5 * z * (xx - yy);
        b[15] = SH_C3_6 * x * (xx - 3.f * yy);
      }
    }
  }
}

// d(basis_k)/d(dir) for k >= 1
TRASE_HD void sh_basis_grad(int deg, const float d[3], float gx[16], float gy[16], float gz[16]) {
  const float x = d[0], y = d[1], z = d[2];
  TRASE_UNROLL
  for (int k = 0; k < 16; ++k) { gx[k] = 0.f; gy[k] = 0.f; gz[k] = 0.f; }
  if (deg > 0) {
    gy[1] = -SH_C1; gz[2] = SH_C1; gx[3] = -SH_C1;
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      gx[4] = SH_C2_0 * y; gy[4] = SH_C2_0 * x;
      gy[5] = SH_C2_1 * z; gz[5] = SH_C2_1 * y;
      gx[6] = SH_C2_2 * -2.f * x; gy[6] = SH_C2_2 * -2.f * y; gz[6] = SH_C2_2 * 4.f * z;
      gx[7] = SH_C2_3 * z; gz[7] = SH_C2_3 * x;
      gx[8] = SH_C2_4 * 2.f * x; gy[8] = SH_C2_4 * -2.f * y;
      if (deg > 2) {
        gx[9] = SH_C3_0 * 6.f * xy; gy[9] = SH_C3_0 * (3.f * xx - 3.f * yy);
        gx[10] = SH_C3_1 * yz; gy[10] = SH_C3_1 * xz; gz[10] = SH_C3_1 * xy;
        gx[11] = SH_C3_2 * -2.f * xy; gy[11] = SH_C3_2 * (4.f * zz - xx - 3.f * yy); gz[11] = SH_C3_2 * 8.f * yz;
        gx[12] = SH_C3_3 * -6.f * xz; gy[12] = SH_C3_3 * -6.f * yz; gz[12] = SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
        gx[13] = SH_C3_4 * (4.f * zz - 3.f * xx - yy); gy[13] = SH_C3_4 * -2.f * xy; gz[13] = SH_C3_4 * 8.f * xz;
        gx[14] = SH_C3_5 * 2.f * xz; gy[14] = SH_C3_5 * -2.f * yz; gz[14] = SH_C3_5 * (xx - yy);
        gx[15] = SH_C3_6 * (3.f * xx - 3.f * yy); gy[15] = SH_C3_6 * -6.f * xy;
      }
    }
  }
}

TRASE_HD int ncoef(int deg) { return (deg + 1) * (deg + 1); }

// Tile rect of a splat (Appendix A.6): C-style truncation of the float quotient.
TRASE_HD void tile_rect(float px, float py, int radius, int gx, int gy, int& x0, int& y0, int& x1, int& y1) {
  const float r = (float)radius;
  x0 = imin(gx, imax(0, (int)((px - r) / (float)TILE)));
  y0 = imin(gy, imax(0, (int)((py - r) / (float)TILE)));
  x1 = imin(gx, imax(0, (int)((px + r + (float)(TILE - 1)) / (float)TILE)));
  y1 = imin(gy, imax(0, (int)((py + r + (float)(TILE - 1)) / (float)TILE)));
}

// ---- exact sub-tile culling -----------------------------------------------------------------
// The lists are binned per 8x8 pixel block ("sub-tile").  Which Gaussians a pixel may see is
// still decided by the 16x16 tile rect above (visible semantics); inside that rect a sub-tile
// drops a Gaussian only when NO pixel of the block can pass the blend gate
// alpha = opacity * exp(power) >= 1/255, i.e. when the minimum of the conic's quadratic form
// q = A dx^2 + 2B dx dy + C dy^2 over the block exceeds 2 ln(255 opacity) (with slack, so that a
// float32 rounding difference can never drop a pair the compositing kernel would have blended).
constexpr int SUB = 8;   // sub-tile edge

// The count (preprocess) and the emit kernel must take bit-identical decisions.  Two inlined copies were once
// contracted into FMAs differently (one pair in ~8 M disagreed), so contraction is switched off for these bodies:
// every operation below is then an individually rounded IEEE op wherever the functions are inlined.
// The per-Gaussian part (logarithm, the two divisions) is hoisted into SubtileCull; the per-block test is
// multiplications, clamps and compares only.
struct SubtileCull {
  float gx, gy, A, B, C, tau, nb_c, nb_a;   // nb_c = -B/C, nb_a = -B/A: slopes of the edge minimisers
  float det, inv_a, xmax, ymax, dy_xmax;    // row-span pruning: extents of the ellipse q <= tau, dy at which dx = +xmax
  int mode;                                 // 0: test blocks, 1: never live, 2: always live
};

TRASE_HD SubtileCull subtile_cull_setup(float gx, float gy, float A, float B, float C, float opacity) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  SubtileCull s;
  s.gx = gx; s.gy = gy; s.A = A; s.B = B; s.C = C; s.nb_c = 0.0f; s.nb_a = 0.0f; s.mode = 0;
  s.det = 0.0f; s.inv_a = 0.0f; s.xmax = 0.0f; s.ymax = 0.0f; s.dy_xmax = 0.0f;
  const float tau_raw = 2.0f * logf(255.0f * opacity);
  s.tau = tau_raw * 1.001f + 1e-3f;
  if (!(tau_raw >= 0.0f)) { s.mode = 1; return s; }   // opacity < 1/255 (or NaN): can never pass the gate
  if (!(A > 0.0f) || !(C > 0.0f) || !(A * C - B * B > 0.0f)) { s.mode = 2; return s; }   // not positive definite: keep
  s.nb_c = -B / C;
  s.nb_a = -B / A;
  s.det = A * C - B * B;
  s.inv_a = 1.0f / A;
  s.xmax = sqrtf(s.tau * C / s.det);        // half-extent of the ellipse in x ...
  s.ymax = sqrtf(s.tau * A / s.det);        // ... and in y
  s.dy_xmax = s.nb_c * s.xmax;              // the point (xmax, dy_xmax) is the ellipse's rightmost point
  return s;
}

// Sub-tile columns [sx0, sx1) of sub-tile row `sy` that can intersect the ellipse q <= tau: the x-extent of the
// ellipse restricted to the row's band of pixel centres (attained at a band end or at the ellipse's extreme point),
// padded by more than any rounding error.  Only a pre-filter: every column inside is still decided by
// subtile_cull_live, so it merely has to be conservative (checked exhaustively on the host build).
TRASE_HD void subtile_row_span(const SubtileCull& s, int sy, int H, int sxmin, int sxmax, int& sx0, int& sx1) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  sx0 = sxmin; sx1 = sxmax;
  if (s.mode == 2) return;                           // not positive definite: every block of the rect is kept
  if (s.mode == 1) { sx1 = sx0; return; }
  const float y0 = (float)(sy * SUB), y1 = (float)imin(sy * SUB + SUB - 1, H - 1);
  const float da = y0 - s.gy, db = y1 - s.gy;        // band of dy = pixel centre - mean
  const float pad = 0.75f;                           // pixels
  if (da > s.ymax + pad || db < -s.ymax - pad) { sx1 = sx0; return; }
  // roots of A dx^2 + 2 B dx dy + C dy^2 = tau at the two band ends (clamped discriminant: an end outside the
  // ellipse contributes its tangent-direction point, which only widens the span)
  const float Da = fmaxf(s.A * s.tau - s.det * da * da, 0.0f), Db = fmaxf(s.A * s.tau - s.det * db * db, 0.0f);
  const float ca = -s.B * da * s.inv_a, cb = -s.B * db * s.inv_a;
  const float ra = sqrtf(Da) * s.inv_a, rb = sqrtf(Db) * s.inv_a;
  float hi = fmaxf(ca + ra, cb + rb), lo = fminf(ca - ra, cb - rb);
  if (s.dy_xmax >= da && s.dy_xmax <= db) hi = s.xmax;       // rightmost point inside the band
  if (-s.dy_xmax >= da && -s.dy_xmax <= db) lo = -s.xmax;    // leftmost point inside the band
  hi = fminf(fmaxf(hi, -s.xmax), s.xmax) + pad;
  lo = fmaxf(fminf(lo, s.xmax), -s.xmax) - pad;
  const float fx0 = floorf((s.gx + lo) * (1.0f / (float)SUB)), fx1 = floorf((s.gx + hi) * (1.0f / (float)SUB));
  // clamp in float first: the products can exceed the int range for degenerate splats
  const int c0 = (int)fminf(fmaxf(fx0, (float)sxmin), (float)sxmax);
  const int c1 = (int)fminf(fmaxf(fx1 + 1.0f, (float)sxmin), (float)sxmax);
  sx0 = c0; sx1 = c1 > c0 ? c1 : c0;
}

// Live sub-tile columns [c0, c1) of sub-tile row `sy`, decided per ROW instead of per block: the blocks of a row that
// the ellipse q <= tau reaches are exactly those whose pixel-centre column range meets the x-extent of the ellipse
// restricted to the row's band of pixel centres (the restriction is convex, so its projection is an interval and every
// x of the interval is attained).  The extent is attained at the ellipse's extreme point when that lies inside the band,
// otherwise at one of the band's (clipped) ends.  Same live set as subtile_cull_live's continuous-box minimum up to
// rounding; `pad` keeps it conservative (the gate threshold already carries its own slack in tau).  O(rows) instead of
// O(rows x columns) per splat -- the count (preprocess) and the emit pass both call THIS function, with FP contraction
// off, so they agree bit for bit.
TRASE_HD void subtile_row_live(const SubtileCull& s, int sy, int W, int H, int sxmin, int sxmax, int& c0, int& c1) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  c0 = sxmin; c1 = sxmax;
  if (s.mode == 1 || sy * SUB >= H) { c1 = c0; return; }
  // columns that exist in the image
  const int sx_img = (W + SUB - 1) / SUB;
  if (c1 > sx_img) c1 = sx_img;
  if (c1 < c0) c1 = c0;
  if (s.mode == 2) return;                           // not positive definite: every block of the rect row is kept
  const float y0 = (float)(sy * SUB), y1 = (float)imin(sy * SUB + SUB - 1, H - 1);
  const float pad = 4e-3f;                           // pixels: covers the rounding of the roots below
  float da = y0 - s.gy, db = y1 - s.gy;              // band of dy = pixel centre - mean
  if (da > s.ymax + pad || db < -s.ymax - pad) { c1 = c0; return; }
  da = fmaxf(da, -s.ymax); db = fminf(db, s.ymax);   // the part of the band inside the ellipse's y-extent
  // x-interval of the ellipse at the two clipped band ends: centre -B dy / A, half-width sqrt(A tau - det dy^2) / A
  const float Da = fmaxf(s.A * s.tau - s.det * da * da, 0.0f), Db = fmaxf(s.A * s.tau - s.det * db * db, 0.0f);
  const float ca = -s.B * da * s.inv_a, cb = -s.B * db * s.inv_a;
  const float ra = sqrtf(Da) * s.inv_a, rb = sqrtf(Db) * s.inv_a;
  float hi = fmaxf(ca + ra, cb + rb), lo = fminf(ca - ra, cb - rb);
  if (s.dy_xmax >= da && s.dy_xmax <= db) hi = s.xmax;       // rightmost point of the ellipse inside the band
  if (-s.dy_xmax >= da && -s.dy_xmax <= db) lo = -s.xmax;    // leftmost point inside the band
  const float xr = s.gx + hi + pad, xl = s.gx + lo - pad;
  // block sx holds the pixel centres 8 sx .. 8 sx + 7: it meets [xl, xr] iff 8 sx <= xr and 8 sx + 7 >= xl
  const float f1 = floorf(xr * (1.0f / (float)SUB));
  const float f0 = ceilf((xl - (float)(SUB - 1)) * (1.0f / (float)SUB));
  // clamp in float first: the products can exceed the int range for degenerate splats
  const int a0 = (int)fminf(fmaxf(f0, (float)c0), (float)c1);
  const int a1 = (int)fminf(fmaxf(f1 + 1.0f, (float)c0), (float)c1);
  c0 = a0; c1 = a1 > a0 ? a1 : a0;
}

TRASE_HD bool subtile_cull_live(const SubtileCull& s, int bx, int by, int W, int H) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  if (s.mode) return s.mode == 2;
  // pixel centres of the block, clipped to the image
  const float x0 = (float)bx, x1 = (float)imin(bx + SUB - 1, W - 1);
  const float y0 = (float)by, y1 = (float)imin(by + SUB - 1, H - 1);
  // closest point of the box to the centre
  const float cx = fminf(fmaxf(s.gx, x0), x1), cy = fminf(fmaxf(s.gy, y0), y1);
  if (cx == s.gx && cy == s.gy) return true;    // centre inside the block
  float best = 3.0e38f;
  // candidate 1: vertical edge facing the centre (dx fixed), optimum dy clamped to the edge
  if (cx != s.gx) {
    const float dx = cx - s.gx;
    float dy = s.nb_c * dx;                      // unconstrained minimiser along the edge
    dy = fminf(fmaxf(dy, y0 - s.gy), y1 - s.gy);
    best = fminf(best, s.A * dx * dx + 2.0f * s.B * dx * dy + s.C * dy * dy);
  }
  // candidate 2: horizontal edge facing the centre
  if (cy != s.gy) {
    const float dy = cy - s.gy;
    float dx = s.nb_a * dy;
    dx = fminf(fmaxf(dx, x0 - s.gx), x1 - s.gx);
    best = fminf(best, s.A * dx * dx + 2.0f * s.B * dx * dy + s.C * dy * dy);
  }
  return best <= s.tau;
}

TRASE_HD bool subtile_live(float gx, float gy, float A, float B, float C, float opacity, int bx, int by, int W, int H) {
  return subtile_cull_live(subtile_cull_setup(gx, gy, A, B, C, opacity), bx, by, W, H);
}

// Forward of one Gaussian.  `sh` points at 48 floats: this Gaussian's (16,3) coefficients,
// zero-padded beyond the active degree, or is null when `color` (precomputed rgb) is given.  `cov_in` null => from scale/rot.
// SH -> RGB of one Gaussian: + 0.5, clamp at 0, per-channel clamp flags.  Its own function so that
// a caller can decide AFTER the geometry whether the colour is needed at all (a tile-row strip: most Gaussians have no pair
// there and their 192 bytes of coefficients are never read).
TRASE_HD void splat_colour_sh(const View& v, const float p[3], const float* sh, Splat& o) {
  float d[3], il, b[16];
  sh_dir(p, v.cam, d, il);
  sh_basis(v.deg, d, b);
  float r0 = 0.f, r1 = 0.f, r2 = 0.f;   // basis entries beyond the active degree are 0
  TRASE_UNROLL
  for (int k = 0; k < 16; ++k) { r0 += b[k] * sh[3 * k]; r1 += b[k] * sh[3 * k + 1]; r2 += b[k] * sh[3 * k + 2]; }
  r0 += 0.5f; r1 += 0.5f; r2 += 0.5f;
  o.clamped = (r0 < 0.f ? 1u : 0u) | (r1 < 0.f ? 2u : 0u) | (r2 < 0.f ? 4u : 0u);
  o.rgb[0] = fmaxf(r0, 0.f); o.rgb[1] = fmaxf(r1, 0.f); o.rgb[2] = fmaxf(r2, 0.f);
}

// Returns false (radius 0) when culled.
// USE_COV / USE_SH are compile-time so that the small per-Gaussian arrays stay in registers.
template <bool USE_COV, bool USE_SH>
TRASE_HD bool splat_forward(const View& v, const float p[3], const float* scale, const float* quat,
                            const float* cov_in, const float* sh, const float* color, Splat& o) {
  o.radius = 0; o.x0 = o.y0 = o.x1 = o.y1 = 0; o.clamped = 0;
  float cov[6];
  if (USE_COV) { for (int i = 0; i < 6; ++i) cov[i] = cov_in[i]; }
  else cov3d_from_scale_rot(scale, v.mod, quat, cov);
  Ewa e;
  ewa_forward(v, p, cov, e);
  if (!(e.t[2] > NEAR_Z)) return false;
  const float* PM = v.PM;
  const float hx = PM[0] * p[0] + PM[4] * p[1] + PM[8] * p[2] + PM[12];
  const float hy = PM[1] * p[0] + PM[5] * p[1] + PM[9] * p[2] + PM[13];
  const float hw = PM[3] * p[0] + PM[7] * p[1] + PM[11] * p[2] + PM[15];
  const float pw = 1.0f / (hw + 0.0000001f);
  const float det = e.a * e.c - e.b * e.b;
  if (det == 0.0f) return false;
  const float det_inv = 1.f / det;
  o.ca = e.c * det_inv; o.cb = -e.b * det_inv; o.cc = e.a * det_inv;
  const float mid = 0.5f * (e.a + e.c);
  const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
  const float rr = ceilf(3.f * sqrtf(lam));
  o.px = ((hx * pw + 1.0f) * (float)v.W - 1.0f) * 0.5f;
  o.py = ((hy * pw + 1.0f) * (float)v.H - 1.0f) * 0.5f;
  // non-finite geometry cannot be binned; treat as culled (the lineage would index out of range)
  if (!(rr < 1.0e9f) || !(fabsf(o.px) < 1.0e9f) || !(fabsf(o.py) < 1.0e9f)) return false;
  const int radius = (int)rr;
  tile_rect(o.px, o.py, radius, v.gx, v.gy, o.x0, o.y0, o.x1, o.y1);
  if ((o.x1 - o.x0) * (o.y1 - o.y0) == 0) { o.x0 = o.y0 = o.x1 = o.y1 = 0; return false; }
  o.depth = e.t[2];
  if (!USE_SH) {
    o.rgb[0] = color[0]; o.rgb[1] = color[1]; o.rgb[2] = color[2];
  } else {
    splat_colour_sh(v, p, sh, o);
  }
  o.radius = radius;
  return true;
}

// Gradients arriving at one Gaussian from the compositing stage.
struct SplatGradIn {
  float d_ca, d_cb, d_cc;  // true partials wrt the conic entries (power = -0.5(ca dx^2 + cc dy^2) - cb dx dy)
  float d_ndcx, d_ndcy;    // wrt the NDC centre (== what means2D.grad carries)
  float d_rgb[3];
  float d_depth;           // wrt view-space z (0 unless depth gradients are enabled)
};

struct SplatGradOut {
  float d_p[3];
  float d_scale[3];
  float d_quat[4];
  float d_cov[6];          // wrt cov3D_precomp's 6 unique entries
};

// Backward of splat_forward for a *visible* Gaussian.  d_sh receives 48 floats ((16,3), zero
// beyond the active degree); `sh` as in splat_forward; `clamped` from forward.
template <bool USE_COV, bool USE_SH>
TRASE_HD void splat_backward(const View& v, const float p[3], const float* scale, const float* quat,
                             const float* cov_in, const float* sh, unsigned clamped,
                             const SplatGradIn& gi, SplatGradOut& go, float* d_sh /* 48 floats when USE_SH */) {
  float cov[6];
  if (USE_COV) { for (int i = 0; i < 6; ++i) cov[i] = cov_in[i]; }
  else cov3d_from_scale_rot(scale, v.mod, quat, cov);
  Ewa e;
  ewa_forward(v, p, cov, e);
  const float a = e.a, b = e.b, c = e.c;
  // conic -> (a,b,c)
  const float det = a * c - b * b;
  const float d2i = 1.0f / (det * det + 0.0000001f);
  const float G = gi.d_ca, Hh = gi.d_cb, I = gi.d_cc;
  const float da = d2i * (-c * c * G + b * c * Hh + (det - a * c) * I);
  const float dc = d2i * ((det - a * c) * G + a * b * Hh - a * a * I);
  const float db = d2i * (2.f * b * c * G - (det + 2.f * b * b) * Hh + 2.f * a * b * I);
  // (a,b,c) -> Sigma (unique entries) and rows of J*W
  const float* A0 = e.A0; const float* A1 = e.A1;
  float dS[6];
  dS[0] = da * A0[0] * A0[0] + db * A0[0] * A1[0] + dc * A1[0] * A1[0];
  dS[3] = da * A0[1] * A0[1] + db * A0[1] * A1[1] + dc * A1[1] * A1[1];
  dS[5] = da * A0[2] * A0[2] + db * A0[2] * A1[2] + dc * A1[2] * A1[2];
  dS[1] = 2.f * da * A0[0] * A0[1] + db * (A0[0] * A1[1] + A0[1] * A1[0]) + 2.f * dc * A1[0] * A1[1];
  dS[2] = 2.f * da * A0[0] * A0[2] + db * (A0[0] * A1[2] + A0[2] * A1[0]) + 2.f * dc * A1[0] * A1[2];
  dS[4] = 2.f * da * A0[1] * A0[2] + db * (A0[1] * A1[2] + A0[2] * A1[1]) + 2.f * dc * A1[1] * A1[2];
  float dA0[3], dA1[3];
  for (int k = 0; k < 3; ++k) {
    dA0[k] = 2.f * da * e.SA0[k] + db * e.SA1[k];
    dA1[k] = 2.f * dc * e.SA1[k] + db * e.SA0[k];
  }
  const float* V = v.V;
  float dj00 = 0.f, dj02 = 0.f, dj11 = 0.f, dj12 = 0.f;
  for (int k = 0; k < 3; ++k) {
    dj00 += dA0[k] * V[4 * k + 0]; dj02 += dA0[k] * V[4 * k + 2];
    dj11 += dA1[k] * V[4 * k + 1]; dj12 += dA1[k] * V[4 * k + 2];
  }
  const float tz = e.t[2];
  const float itz = 1.f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
  // lineage rule: clamped t.x/t.y behave as independent variables with no incoming gradient
  float dt[3];
  dt[0] = e.clx ? 0.f : -v.fx * itz2 * dj02;
  dt[1] = e.cly ? 0.f : -v.fy * itz2 * dj12;
  dt[2] = -v.fx * itz2 * dj00 - v.fy * itz2 * dj11 + 2.f * v.fx * e.txc * itz3 * dj02 +
          2.f * v.fy * e.tyc * itz3 * dj12 + gi.d_depth;
  for (int k = 0; k < 3; ++k) go.d_p[k] = V[4 * k + 0] * dt[0] + V[4 * k + 1] * dt[1] + V[4 * k + 2] * dt[2];
  // NDC centre -> position
  const float* PM = v.PM;
  const float hx = PM[0] * p[0] + PM[4] * p[1] + PM[8] * p[2] + PM[12];
  const float hy = PM[1] * p[0] + PM[5] * p[1] + PM[9] * p[2] + PM[13];
  const float hw = PM[3] * p[0] + PM[7] * p[1] + PM[11] * p[2] + PM[15];
  const float pw = 1.0f / (hw + 0.0000001f);
  const float m1 = hx * pw * pw, m2 = hy * pw * pw;
  for (int k = 0; k < 3; ++k)
    go.d_p[k] += (PM[4 * k + 0] * pw - PM[4 * k + 3] * m1) * gi.d_ndcx + (PM[4 * k + 1] * pw - PM[4 * k + 3] * m2) * gi.d_ndcy;
  // colour -> SH coefficients and view direction
  if (USE_SH) {
    float dr[3];
    for (int ch = 0; ch < 3; ++ch) dr[ch] = ((clamped >> ch) & 1u) ? 0.f : gi.d_rgb[ch];
    float d[3], il, bb[16], gx[16], gy[16], gz[16];
    sh_dir(p, v.cam, d, il);
    sh_basis(v.deg, d, bb);
    sh_basis_grad(v.deg, d, gx, gy, gz);
    float dd[3] = {0.f, 0.f, 0.f};
    TRASE_UNROLL
    for (int k = 0; k < 16; ++k) {
      const float s0 = sh[3 * k], s1 = sh[3 * k + 1], s2 = sh[3 * k + 2];
      d_sh[3 * k] = bb[k] * dr[0]; d_sh[3 * k + 1] = bb[k] * dr[1]; d_sh[3 * k + 2] = bb[k] * dr[2];
      const float w = s0 * dr[0] + s1 * dr[1] + s2 * dr[2];
      dd[0] += gx[k] * w; dd[1] += gy[k] * w; dd[2] += gz[k] * w;
    }
    const float dot = d[0] * dd[0] + d[1] * dd[1] + d[2] * dd[2];
    for (int k = 0; k < 3; ++k) go.d_p[k] += (dd[k] - d[k] * dot) * il;
  }
  // Sigma -> scale, quaternion (or the precomputed covariance itself)
  for (int i = 0; i < 6; ++i) go.d_cov[i] = dS[i];
  for (int k = 0; k < 3; ++k) go.d_scale[k] = 0.f;
  for (int k = 0; k < 4; ++k) go.d_quat[k] = 0.f;
  if (!USE_COV) {
    float R[9];
    quat_to_rot(quat, R);
    const float sm[3] = {v.mod * scale[0], v.mod * scale[1], v.mod * scale[2]};
    // full symmetric gradient matrix
    const float Gm[9] = {dS[0], 0.5f * dS[1], 0.5f * dS[2], 0.5f * dS[1], dS[3], 0.5f * dS[4],
                         0.5f * dS[2], 0.5f * dS[4], dS[5]};
    float dR[9];
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) {
        // dL/dL_ik = 2 * sum_j G_ij L_jk, L_jk = R_jk * sm_k
        const float dL = 2.f * (Gm[3 * i] * R[k] + Gm[3 * i + 1] * R[3 + k] + Gm[3 * i + 2] * R[6 + k]) * sm[k];
        go.d_scale[k] += v.mod * R[3 * i + k] * dL;
        dR[3 * i + k] = sm[k] * dL;
      }
    const float r = quat[0], x = quat[1], y = quat[2], z = quat[3];
    go.d_quat[0] = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
    go.d_quat[1] = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.f * x * dR[8]);
    go.d_quat[2] = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.f * y * dR[8]);
    go.d_quat[3] = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
  }
}

}  // namespace trase
