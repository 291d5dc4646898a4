// TEST INFRASTRUCTURE ONLY: compiles trase_amd/csrc/gs_math.h for the host (g++) so the
// per-Gaussian forward/backward arithmetic that the HIP kernels run can be checked against the
// float64 oracle in the GPU-less container.  Never loaded by the product.
#include "../../trase_amd/csrc/gs_math.h"

using namespace trase;

extern "C" {

struct HsView { float V[16]; float PM[16]; float cam[3]; float tanx, tany, mod; int W, H, deg; };

static View make_view(const HsView* h) {
  View v;
  for (int i = 0; i < 16; ++i) { v.V[i] = h->V[i]; v.PM[i] = h->PM[i]; }
  for (int i = 0; i < 3; ++i) v.cam[i] = h->cam[i];
  v.tanx = h->tanx; v.tany = h->tany; v.mod = h->mod; v.W = h->W; v.H = h->H; v.deg = h->deg;
  v.fx = (float)h->W / (2.0f * h->tanx); v.fy = (float)h->H / (2.0f * h->tany);
  v.gx = (h->W + TILE - 1) / TILE; v.gy = (h->H + TILE - 1) / TILE;
  return v;
}

// out rows: px py depth ca cb cc r g b radius x0 y0 x1 y1 clamped  (15 floats)
void hs_forward(const HsView* hv, int n, const float* p, const float* scale, const float* quat, const float* cov,
                const float* sh /* (n,16,3) or null */, const float* color, float* out) {
  View v = make_view(hv);
  for (int i = 0; i < n; ++i) {
    Splat o;
    float shl[48];
    if (sh) for (int k = 0; k < 48; ++k) shl[k] = (k < 3 * ncoef(v.deg)) ? sh[48 * i + k] : 0.f;
    const float* sc = scale ? scale + 3 * i : nullptr; const float* qq = quat ? quat + 4 * i : nullptr;
    const float* cv = cov ? cov + 6 * i : nullptr; const float* cl = color ? color + 3 * i : nullptr;
    bool vis = cov ? (sh ? splat_forward<true, true>(v, p + 3 * i, sc, qq, cv, shl, cl, o) : splat_forward<true, false>(v, p + 3 * i, sc, qq, cv, shl, cl, o))
                   : (sh ? splat_forward<false, true>(v, p + 3 * i, sc, qq, cv, shl, cl, o) : splat_forward<false, false>(v, p + 3 * i, sc, qq, cv, shl, cl, o));
    float* r = out + 15 * i;
    for (int k = 0; k < 15; ++k) r[k] = 0.f;
    if (!vis) continue;
    r[0] = o.px; r[1] = o.py; r[2] = o.depth; r[3] = o.ca; r[4] = o.cb; r[5] = o.cc;
    r[6] = o.rgb[0]; r[7] = o.rgb[1]; r[8] = o.rgb[2]; r[9] = (float)o.radius;
    r[10] = (float)o.x0; r[11] = (float)o.y0; r[12] = (float)o.x1; r[13] = (float)o.y1; r[14] = (float)o.clamped;
  }
}

// gin rows: d_ca d_cb d_cc d_ndcx d_ndcy d_r d_g d_b d_depth (9); gout rows: d_p(3) d_scale(3) d_quat(4) d_cov(6) (16)
void hs_backward(const HsView* hv, int n, const float* p, const float* scale, const float* quat, const float* cov,
                 const float* sh, const float* clamped, const float* gin, float* gout, float* d_sh /* (n,16,3) */) {
  View v = make_view(hv);
  for (int i = 0; i < n; ++i) {
    SplatGradIn gi;
    const float* g = gin + 9 * i;
    gi.d_ca = g[0]; gi.d_cb = g[1]; gi.d_cc = g[2]; gi.d_ndcx = g[3]; gi.d_ndcy = g[4];
    gi.d_rgb[0] = g[5]; gi.d_rgb[1] = g[6]; gi.d_rgb[2] = g[7]; gi.d_depth = g[8];
    SplatGradOut go;
    float shl[48], dsh[48];
    for (int k = 0; k < 48; ++k) dsh[k] = 0.f;
    if (sh) for (int k = 0; k < 48; ++k) shl[k] = (k < 3 * ncoef(v.deg)) ? sh[48 * i + k] : 0.f;
    const float* sc = scale ? scale + 3 * i : nullptr; const float* qq = quat ? quat + 4 * i : nullptr;
    const float* cv = cov ? cov + 6 * i : nullptr;
    if (cov) { if (sh) splat_backward<true, true>(v, p + 3 * i, sc, qq, cv, shl, (unsigned)clamped[i], gi, go, dsh);
               else splat_backward<true, false>(v, p + 3 * i, sc, qq, cv, shl, (unsigned)clamped[i], gi, go, dsh); }
    else { if (sh) splat_backward<false, true>(v, p + 3 * i, sc, qq, cv, shl, (unsigned)clamped[i], gi, go, dsh);
           else splat_backward<false, false>(v, p + 3 * i, sc, qq, cv, shl, (unsigned)clamped[i], gi, go, dsh); }
    float* r = gout + 16 * i;
    for (int k = 0; k < 3; ++k) r[k] = go.d_p[k];
    for (int k = 0; k < 3; ++k) r[3 + k] = go.d_scale[k];
    for (int k = 0; k < 4; ++k) r[6 + k] = go.d_quat[k];
    for (int k = 0; k < 6; ++k) r[10 + k] = go.d_cov[k];
    if (d_sh) for (int k = 0; k < 48; ++k) d_sh[48 * i + k] = dsh[k];
  }
}
}

extern "C" int hs_subtile_live(float gx, float gy, float A, float B, float C, float opacity, int bx, int by, int W, int H) {
  return trase::subtile_live(gx, gy, A, B, C, opacity, bx, by, W, H) ? 1 : 0;
}

// live sub-tiles of a splat's tile rect, enumerated (a) over the whole rect and (b) through the row-span pre-filter:
// both must give the same set.  Returns the number of live sub-tiles of (a); *mismatch counts the differences.
extern "C" int hs_subtile_enumerate(float gx, float gy, float A, float B, float C, float opacity, int radius, int W, int H,
                                    int* mismatch, int* tested_full, int* tested_pruned) {
  using namespace trase;
  const int gxt = (W + TILE - 1) / TILE, gyt = (H + TILE - 1) / TILE;
  int x0, y0, x1, y1;
  tile_rect(gx, gy, radius, gxt, gyt, x0, y0, x1, y1);
  const SubtileCull cull = subtile_cull_setup(gx, gy, A, B, C, opacity);
  int live = 0, bad = 0, nf = 0, np = 0;
  for (int sy = 2 * y0; sy < 2 * y1; ++sy) {
    int sx0, sx1;
    subtile_row_span(cull, sy, H, 2 * x0, 2 * x1, sx0, sx1);
    for (int sx = 2 * x0; sx < 2 * x1; ++sx) {
      const int bx = sx * SUB, by = sy * SUB;
      const bool full = bx < W && by < H && subtile_cull_live(cull, bx, by, W, H);
      const bool in = sx >= sx0 && sx < sx1;
      ++nf; np += in;
      if (full) { ++live; if (!in) ++bad; }
    }
  }
  *mismatch = bad; *tested_full = nf; *tested_pruned = np;
  return live;
}

// live sub-tiles of a splat's tile rect by the per-row interval rule (subtile_row_live) vs the per-block rule
// (subtile_cull_live): *only_block = blocks the block rule keeps and the row rule drops, *only_row = the converse.
// out_mask (may be null): one byte per candidate of the rect, row major: bit 0 = block rule, bit 1 = row rule.
extern "C" int hs_subtile_rows(float gx, float gy, float A, float B, float C, float opacity, int radius, int W, int H,
                               int* only_block, int* only_row, unsigned char* out_mask, int out_cap, int* rect) {
  using namespace trase;
  const int gxt = (W + TILE - 1) / TILE, gyt = (H + TILE - 1) / TILE;
  int x0, y0, x1, y1;
  tile_rect(gx, gy, radius, gxt, gyt, x0, y0, x1, y1);
  if (rect) { rect[0] = 2 * x0; rect[1] = 2 * y0; rect[2] = 2 * x1; rect[3] = 2 * y1; }
  const SubtileCull cull = subtile_cull_setup(gx, gy, A, B, C, opacity);
  int live = 0, ob = 0, orow = 0, k = 0;
  for (int sy = 2 * y0; sy < 2 * y1; ++sy) {
    int c0, c1;
    subtile_row_live(cull, sy, W, H, 2 * x0, 2 * x1, c0, c1);
    for (int sx = 2 * x0; sx < 2 * x1; ++sx, ++k) {
      const int bx = sx * SUB, by = sy * SUB;
      const bool blk = bx < W && by < H && subtile_cull_live(cull, bx, by, W, H);
      const bool row = sx >= c0 && sx < c1;
      if (out_mask && k < out_cap) out_mask[k] = (unsigned char)((blk ? 1 : 0) | (row ? 2 : 0));
      live += row; ob += (blk && !row); orow += (row && !blk);
    }
  }
  *only_block = ob; *only_row = orow;
  return live;
}
