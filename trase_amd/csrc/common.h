// common.h -- shared device helpers and host-side launch plumbing (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/trase_rast.h"
#include "gs_math.h"

namespace trase {

// ----------------------------------------------------------------------------------------------
// wave64 primitives
// ----------------------------------------------------------------------------------------------
constexpr int WAVE = 64;

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float v) {
  // lanes with no source (or rows masked off) receive 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}

// Sum over the 64 lanes of a wave; the total is valid in lane 63 only.
// GFX9 DPP: row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v += dpp_f<0x111>(v);
  v += dpp_f<0x112>(v);
  v += dpp_f<0x114>(v);
  v += dpp_f<0x118>(v);
  v += dpp_f<0x142, 0xa>(v);
  v += dpp_f<0x143, 0xc>(v);
  return v;
}

// Inclusive scans over the 64 lanes (lane 0 first).  `ident` fills lanes that have no source.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_fill(float v, float ident) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, v),
                                                                CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_scan_mul(float v) {
  v *= dpp_fill<0x111>(v, 1.0f);
  v *= dpp_fill<0x112>(v, 1.0f);
  v *= dpp_fill<0x114>(v, 1.0f);
  v *= dpp_fill<0x118>(v, 1.0f);
  v *= dpp_fill<0x142, 0xa>(v, 1.0f);
  v *= dpp_fill<0x143, 0xc>(v, 1.0f);
  return v;
}
__device__ __forceinline__ float wave_scan_add(float v) {
  v += dpp_fill<0x111>(v, 0.0f);
  v += dpp_fill<0x112>(v, 0.0f);
  v += dpp_fill<0x114>(v, 0.0f);
  v += dpp_fill<0x118>(v, 0.0f);
  v += dpp_fill<0x142, 0xa>(v, 0.0f);
  v += dpp_fill<0x143, 0xc>(v, 0.0f);
  return v;
}
// Single-instruction scan steps (the builtin form costs a v_mov_dpp + the arithmetic op per step).
// A lane whose DPP source is out of range, or whose row is masked off, is simply not written, which
// is the identity for both scans.  s_nop 1 = the 2 wait states a DPP read needs after a VALU write.
__device__ __forceinline__ float wave_scan_mul_asm(float v) {
  asm volatile(
      "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}
__device__ __forceinline__ float wave_scan_add_asm(float v) {
  asm volatile(
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}

// Two independent scans interleaved: the partner's instruction is one of the two wait states a DPP read needs after
// the VALU write of its source, so a step costs 2 DPP + one s_nop 0 instead of 2 DPP + two s_nop 1.
#define TRASE_SCAN2_STEP(OP, CTRL)                         \
  OP " %0, %0, %0 " CTRL "\n\t" OP " %1, %1, %1 " CTRL "\n\ts_nop 0\n\t"
__device__ __forceinline__ void wave_scan_mul2_asm(float& a, float& b) {
  asm volatile("s_nop 1\n\t"
               TRASE_SCAN2_STEP("v_mul_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")
               TRASE_SCAN2_STEP("v_mul_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
               TRASE_SCAN2_STEP("v_mul_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf")
               TRASE_SCAN2_STEP("v_mul_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf")
               TRASE_SCAN2_STEP("v_mul_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
               "v_mul_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "v_mul_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf"
               : "+v"(a), "+v"(b));
}
// Inclusive add-scans of a and b into fresh registers: with bound_ctrl:0 a lane without a source reads 0, the
// identity, so the first step needs no copy of the inputs (they stay live for the caller).
__device__ __forceinline__ void wave_scan_add2_out_asm(float a, float b, float& oa, float& ob) {
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_add_f32_dpp %1, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 0\n\t"
               TRASE_SCAN2_STEP("v_add_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
               TRASE_SCAN2_STEP("v_add_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf")
               TRASE_SCAN2_STEP("v_add_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf")
               TRASE_SCAN2_STEP("v_add_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
               "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf"
               : "=&v"(oa), "=&v"(ob) : "v"(a), "v"(b));
}
__device__ __forceinline__ void wave_scan_add2_asm(float& a, float& b) {
  asm volatile("s_nop 1\n\t"
               TRASE_SCAN2_STEP("v_add_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")
               TRASE_SCAN2_STEP("v_add_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
               TRASE_SCAN2_STEP("v_add_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf")
               TRASE_SCAN2_STEP("v_add_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf")
               TRASE_SCAN2_STEP("v_add_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
               "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf"
               : "+v"(a), "+v"(b));
}
#undef TRASE_SCAN2_STEP

// ---- per-(sub-tile, Gaussian) blend exponent --------------------------------------------------
// alpha_raw = opacity * exp(power) is evaluated as exp2(e) with
//   e(j,i) = log2(e) * power + log2(opacity),   (j,i) = pixel coordinates inside the 8x8 sub-tile,
// expanded once per pair into a polynomial in (j,i) around the sub-tile origin.  Forward and
// backward evaluate the SAME expression tree (explicit fmaf), so their gate decisions agree bit
// for bit:   power <= 0  <=>  e <= thr (= log2 opacity + the polynomial's own rounding allowance, see pair_poly);
// alpha >= 1/255  <=>  e >= LOG2_ALPHA_MIN.
constexpr float LOG2_ALPHA_MIN = -7.994353436858858f;   // log2(1/255)
struct PairPoly { float k0, kj, ki, kjj, kii, kij, thr; };

__device__ __forceinline__ PairPoly pair_poly(float2 gxy, float4 co, float bx, float by) {
  constexpr float L = 1.4426950408889634f;
  const float rx = gxy.x - bx, ry = gxy.y - by;
  const float A = co.x, B = co.y, C = co.z;
  PairPoly k;
  const float lo = __log2f(co.w);
  k.k0 = fmaf(L, fmaf(-0.5f, fmaf(A * rx, rx, C * ry * ry), -(B * rx) * ry), lo);
  k.kj = L * fmaf(A, rx, B * ry);
  k.ki = L * fmaf(C, ry, B * rx);
  k.kjj = -0.5f * L * A;
  k.kii = -0.5f * L * C;
  k.kij = -L * B;
  // The lineage's "power > 0 -> skip" guard: for a positive-definite conic power <= 0 holds exactly, and the lineage's
  // direct evaluation of the quadratic form in (pixel - centre) keeps that sign at the splat's centre.  The polynomial
  // about the sub-tile origin cancels terms of size M = sum |k| * (monomial bound), so its value carries ~1e-6 M of
  // rounding and would trip the guard spuriously at the centre pixel of a large flat splat (sigma > 100 px: long-focal
  // cameras, the camera inside the cloud).  The guard therefore allows exactly that evaluation error; a genuinely positive
  // power (a non-positive-definite conic from a caller-supplied covariance) is still skipped.
  const float M = fabsf(k.k0) + 7.0f * (fabsf(k.kj) + fabsf(k.ki)) + 49.0f * (fabsf(k.kjj) + fabsf(k.kii) + fabsf(k.kij));
  k.thr = fmaf(9.5e-7f, M, lo);
  return k;
}
// row part (pixel row i) and full exponent (pixel column j)
__device__ __forceinline__ float poly_row_base(const PairPoly& k, float i, float ii) { return fmaf(ii, k.kii, fmaf(i, k.ki, k.k0)); }
__device__ __forceinline__ float poly_row_slope(const PairPoly& k, float i) { return fmaf(i, k.kij, k.kj); }
// Horner in j: no j^2 operand (the half-wave backward keeps j per lane in a register; a second one for j^2 costs
// occupancy).  Every kernel -- forward and all backward formulations -- evaluates this same tree.
__device__ __forceinline__ float poly_eval(const PairPoly& k, float base, float slope, float j) {
  return fmaf(j, fmaf(j, k.kjj, slope), base);
}

// Workgroups are dispatched round-robin over the 8 XCDs, each with its own L2.  Mapping block b to the x-th
// CONTIGUOUS eighth of the work (x = b mod 8) keeps spatially adjacent tiles -- which share their Gaussians'
// rows -- on one XCD, so a per-Gaussian row is fetched into one L2 instead of eight.  Bijective for any nb.
__device__ __forceinline__ int xcd_block(int b, int nb) {
#ifdef TRASE_NO_XCD_MAP
  return b;
#else
  const int x = b & 7, loc = b >> 3, q = nb >> 3, r = nb & 7;
  return x * q + (x < r ? x : r) + loc;
#endif
}

// Sub-tile visited by the i-th unit of the dispatch order.  The compositing kernels are sensitive to which sub-tiles are in
// flight together: neighbours share most of their Gaussians' geometry and channel rows, and the wave slots of one XCD hold
// 200-450 sub-tiles at a time.  In image (row-major) order that is one 240-wide row of sub-tiles whose Gaussians (~4 MB of
// rows) overflow the XCD's L2 and share nothing with the row above; in BxB blocks it is a compact patch whose Gaussians
// fit.  (Dispatching the longest lists first instead -- to shorten the tail -- made the forward 28 % SLOWER: locality,
// not the tail, is what the order buys.)  Order: bands of B sub-tile rows; inside a band, blocks of B columns left to
// right; inside a block, row-major.  Ragged right / bottom blocks are narrower / lower.  Bijective on [0, gx * gy).
template <int B>
__device__ __forceinline__ void blocked_tile(int i, int gx, int gy, int& tx, int& ty) {
  const int band = i / (B * gx), rem = i - band * (B * gx);
  const int hb = min(B, gy - band * B);                   // rows of this band
  const int ncol = (gx + B - 1) / B;
  const int sc = min(rem / (hb * B), ncol - 1);           // block column
  const int r2 = rem - sc * hb * B;
  const int wb = min(B, gx - sc * B);                     // columns of this block
  ty = band * B + r2 / wb;
  tx = sc * B + r2 - (r2 / wb) * wb;
}

// value of lane-1 (lane 0 receives `ident`): DPP wave_shr:1
__device__ __forceinline__ float wave_shr1(float v, float ident) { return dpp_fill<0x138>(v, ident); }

__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

__device__ __forceinline__ float wave_sum_all(float v) { return readlane_f(wave_sum_lane63(v), 63); }

__device__ __forceinline__ unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ unsigned long long lanemask_lt() {
  const unsigned l = lane_id();
  return (l == 0) ? 0ull : (~0ull >> (64 - l));
}

// float atomic add that must lower to global_atomic_add_f32 (no CAS loop)
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// ----------------------------------------------------------------------------------------------
// host-side plumbing
// ----------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

// profiling scopes: when enabled, brackets a launch with two events on `stream`
struct ProfScope {
  ProfScope(const char* name, hipStream_t stream);
  ~ProfScope();
  const char* name; hipStream_t stream; void* rec;
};
bool prof_on();

#define TRASE_CHECK(expr)                                  \
  do {                                                     \
    int _rc = ::trase::check_hip((expr), #expr);           \
    if (_rc != 0) return _rc;                              \
  } while (0)

// after a kernel launch: always catch launch errors; in debug mode also synchronise
#define TRASE_POST_LAUNCH(name, stream, debug)                                          \
  do {                                                                                  \
    int _rc = ::trase::check_hip(hipGetLastError(), name);                              \
    if (_rc != 0) return _rc;                                                           \
    if (debug) {                                                                        \
      if ((debug) > 1) fprintf(stderr, "[trase] %s\n", name);                           \
      _rc = ::trase::check_hip(hipStreamSynchronize(stream), name " (debug sync)");     \
      if (_rc != 0) return _rc;                                                         \
    }                                                                                   \
  } while (0)

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// 8x8 sub-tile rows [lo, hi) of the strip the settings select (whole image when tile_row_begin == tile_row_end == 0)
static inline void strip_subtile_rows(const TraseRastSettings& s, int& lo, int& hi) {
  const int gy8 = (s.image_height + SUB - 1) / SUB;
  lo = 0; hi = gy8;
  if (s.tile_row_begin != 0 || s.tile_row_end != 0) {
    lo = 2 * s.tile_row_begin; hi = 2 * s.tile_row_end;
    if (lo < 0) lo = 0;
    if (hi > gy8) hi = gy8;
    if (hi < lo) hi = lo;
  }
}

// ----------------------------------------------------------------------------------------------
// workspace carving (must match trase_rast_sizes)
// ----------------------------------------------------------------------------------------------
enum { HDR_R = 0, HDR_OVERFLOW = 1, HDR_R_EFF = 2, HDR_PACK = 3, HDR_WORDS = 64 };
// HDR_PACK: 0 = list values are emit-order slots (ids through pair_gauss / point_list); jb > 0 = list values are
// (Gaussian id << jb) | (index of the pair among the Gaussian's own pairs): decided on the device (every Gaussian must have
// fewer than 2^jb pairs), it saves the compositing kernels the slot -> id load level
__device__ __forceinline__ uint32_t list_id(uint32_t v, uint32_t jb) { return v >> jb; }
__device__ __forceinline__ uint32_t list_slot(uint32_t v, uint32_t jb, const uint32_t* id_end, const uint32_t* tiles) {
  const uint32_t id = v >> jb;
  return id_end[id] - tiles[id] + (v & ((1u << jb) - 1u));
}

struct GeomBuf {           // saved between forward and backward
  uint32_t* hdr;           // HDR_WORDS counters
  float2* xy;              // (P) pixel centre
  float4* conic_o;         // (P) conic a,b,c + opacity
  float4* rgbd;            // (P) colour + view depth
  uint32_t* tiles;         // (P) tiles touched (0 == culled)
  uint32_t* clamped;       // (P) colour clamp bits
  float4* geo;             // (P, 4) one 64-byte record per Gaussian: {x, y, first row slot, radius}, conic_o, rgbd,
                           // rgbd as bf16 pairs {hi(r,g), hi(b,d), lo(r,g), lo(b,d)}
  uint32_t* ftab;          // (P, 32) F = 32 only: the feature row as bf16 [hi 32 | lo 32] (x = hi + lo to ~2^-17): what the
                           // MFMA compositing kernels multiply with -- split once per Gaussian, not once per pair
};
struct BinBuf {            // saved between forward and backward
  uint32_t* point_list;    // (capacity) Gaussian ids, sub-tile-major, depth-ordered
  uint32_t* pair_slot;     // (capacity) emit-order slot of every list entry (Gaussian-major numbering)
  uint2* ranges;           // (T) [start,end) per sub-tile
  const uint32_t* id_end = nullptr;   // PreBuf::id_end (with GeomBuf::tiles: row slot of a packed list value, HDR_PACK); forward only
};
struct ImgBuf {            // saved between forward and backward
  float* final_T;          // (H*W)
  uint32_t* n_contrib;     // (H*W)
};
struct SortBufs {          // ping-pong storage of one radix sort
  uint32_t* keys[2];
  uint32_t* vals[2];
  uint32_t* hist;          // (digits * nb_max), digits = 256 (2048 for the depth sort)
  uint32_t* digit_total;   // (digits * passes) one row per radix pass
  int nb_max;
  int hist_copies;         // how many (digits * nb_max) histograms `hist` holds: >= the number of passes enables the short sorts' fused passes
};
// Short sorts (at most RS_SMALL_NB workgroups of 2048 items = 32k items: BASELINE config 1, small test scenes): a pass is ONE launch --
// the scatter workgroups scan the per-workgroup histograms themselves and count the NEXT pass's digits at the destinations with
// global integer atomics (deterministic; one per run of equal words inside a wave), so a sort of p passes is 1 + p launches instead
// of 3 p.  Measured (profiles/r6_ab_experiments.txt): 1 000 Gaussians at 128 x 128: radix kernels 0.105 -> 0.060 ms per view.  NOT
// for longer sorts: device-scope atomics are served behind the per-XCD L2s -- at 150k keys (74 workgroups) a fused pass took 28 us
// against 15 us for its three launches
constexpr int RS_SMALL_NB = 16;
constexpr int RS_SMALL_COPIES = 4;
inline int rs_hist_copies(int nb) { return nb <= RS_SMALL_NB ? RS_SMALL_COPIES : 1; }
struct PreBuf {            // stage-1 scratch (P-sized), read again by stage 2
  SortBufs sort;           // depth sort; sorted ids end in sort.vals[0]
  uint32_t* offsets;       // (P) inclusive scan of tiles in depth-rank order
  uint32_t* id_end;        // (P) the same, indexed by Gaussian id: the pairs of Gaussian i are [id_end[i] - tiles[i], id_end[i])
  uint32_t* block_sums;    // scan partials: (P/1024 + 2) of the live sub-tile counts, then as many of the rect areas
  uint32_t* live_ids;      // (P) tile-row strips: the ids of the Gaussians with a pair in the strip, ASCENDING (the sorted list in
                           // sort.vals[0] visits memory in depth order: scattered rows; this one walks it monotonically)
};
struct PairBuf {           // stage-2 scratch (capacity-sized)
  SortBufs sort;           // vals[] = {spare, bin.pair_slot} arranged by the caller
  uint32_t* spare_vals;
  uint32_t* pair_gauss;    // (capacity) Gaussian id of every pair in emit order
};
// backward scratch acc: (P, BWD_ACC) = d_ndc(2) d_conic(3) d_opacity(1) d_rgb(3) d_depth(1) pad(2), written by reduce_rows
constexpr int BWD_ACC = 12;
enum { ACC_NDCX = 0, ACC_NDCY = 1, ACC_CA = 2, ACC_CB = 3, ACC_CC = 4, ACC_OP = 5, ACC_R = 6, ACC_G = 7, ACC_B = 8, ACC_D = 9 };

size_t geom_bytes(int P);
size_t bin_bytes(int64_t cap, int T);
size_t img_bytes(int W, int H);
size_t pre_bytes(int P);
size_t tmp_bytes(int64_t cap);
size_t bwd_tmp_bytes(int P, int F, int64_t cap);
static inline int bwd_row_floats(int F) { return F + 12; }
// distance between the rows of consecutive slots, in floats (experiment knob: 64 puts every F = 32 row on its own pair of
// 128-byte lines)
#ifndef TRASE_ROW_STRIDE32
#define TRASE_ROW_STRIDE32 44
#endif
constexpr int bwd_row_stride(int F) { return F == 32 ? TRASE_ROW_STRIDE32 : F + 12; }
GeomBuf carve_geom(void* p, int P);
BinBuf carve_bin(void* p, int64_t cap, int T);
ImgBuf carve_img(void* p, int W, int H);
PreBuf carve_pre(void* p, int P);
PairBuf carve_tmp(void* p, int64_t cap);

// ----------------------------------------------------------------------------------------------
// kernel launchers (one per .hip file)
// ----------------------------------------------------------------------------------------------
struct LaunchCtx { hipStream_t stream; int debug; int variant; };

// read-once data (the gradient rows in reduce_rows): non-temporal loads, 0.161 -> 0.149 ms there.  Measured and NOT used
// elsewhere: non-temporal STORES of the feature map (forward 0.235 -> 0.427 ms) and of the gradient rows (backward
// 0.43 -> 0.77 ms), non-temporal loads of the cotangent planes (backward 0.43 -> 0.49 ms).
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4* p) {
  const f4v q = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
  return make_float4(q.x, q.y, q.z, q.w);
}
__device__ __forceinline__ void st_stream(float4* p, const float4& v) {
  const f4v q = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(q, reinterpret_cast<f4v*>(p));
}

// (a, b) -> packed bf16 high parts and packed bf16 residuals (round to nearest even, twice)
__device__ __forceinline__ void split_pk(float a, float b, unsigned& hi, unsigned& lo) {
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(ra), "v"(rb));
}
__device__ __forceinline__ float4 split_rgbd(float r, float g, float b, float d) {
  unsigned h01, l01, h23, l23;
  split_pk(r, g, h01, l01);
  split_pk(b, d, h23, l23);
  return make_float4(__uint_as_float(h01), __uint_as_float(h23), __uint_as_float(l01), __uint_as_float(l23));
}
int launch_feature_table(const LaunchCtx& c, const float* feats, const uint32_t* tiles, int P, uint32_t* ftab);

int launch_preprocess_fwd(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, int32_t* radii,
                          const GeomBuf& g, uint32_t* depth_keys, bool key27);
int launch_preprocess_bwd(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, const int32_t* radii,
                          const GeomBuf& g, const float* acc, const TraseRastGrads& gr);


// Gaussians [p_begin, p_end) only (p_begin a multiple of 64; p_end = P or a multiple of 64)
int launch_preprocess_bwd_raw(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastRawInputs& raw,
                              const int32_t* radii, const GeomBuf& g, const float* acc, const TraseRastRawGrads& gr,
                              int p_begin = 0, int p_end = -1, int zero_dead_feats = 0, const uint32_t* live_ids = nullptr);
int launch_zero_live_rows(const LaunchCtx& c, const GeomBuf& g, const PreBuf& pre, int P, int F, const TraseRastRawGrads& gr);

// stable LSD radix sort of (key,val) u32 pairs on bits [bit_lo, bit_hi); n is read on the device
// from *n_ptr and clamped to n_cap.  Result ends in keys[out_idx]/vals[out_idx] (returned).
int radix_sort_pairs(const LaunchCtx& c, const SortBufs& t, const uint32_t* n_ptr, uint32_t n_cap, int bit_lo, int bit_hi,
                     bool vals_are_iota, int* out_idx, int digit_bits = 8, int start = 0, uint32_t flag_key = 0u,
                     uint32_t* flag_word = nullptr);     // flag_word: *flag_word |= 2 when the first pass meets a key == flag_key
// zero-fill as a KERNEL: a hipMemsetAsync captured into a launch graph did not clear its buffer from the second replay on
// (round 5: gradient rows of stale pair flags -- NaN under TRASE_POISON -- on every graph hit after the first; ROCm 7.0.2)
int launch_zero_bytes(void* p, size_t bytes, hipStream_t stream);
int launch_fill_u32(uint32_t* p, uint32_t v, hipStream_t stream);
int launch_split_pair_ids(const LaunchCtx& c, const uint32_t* sorted, int P, uint32_t* ids0, uint32_t* ids1);
int radix_passes(int bit_lo, int bit_hi, int digit_bits = 8);
// The depth sort (round 5).  Default: an order-preserving 27-bit key -- the float32 depth bits ABOVE those of the 0.2 near-cull
// plane (z > 0.2 for every live Gaussian), saturated at 2^27 - 2 -- in THREE 9-bit passes (9 launches; measured -16 us per view
// against four 8-bit passes over the raw float bits, point lists bit-identical).  Exact as long as no live Gaussian lies beyond
// z = 13 107 (2^27 float steps above 0.2): the first histogram pass sees every key and raises bit 1 of the header's overflow word
// when one is saturated, which the callers treat like a pair-buffer overflow -- the iteration is repeated (sync policy: at
// once; sync-free: guarded consumers skip it, the report switches the process over) with TRASE_VARIANT_DEPTH32: the raw float
// bits in four 8-bit passes, which is also what the two-view forward uses (it needs the sign bit for the view index).
struct DepthSortCfg { int key_bits, digit_bits, passes, start; };     // start: ping-pong buffer the sort begins in (ids must END in vals[0])
inline DepthSortCfg depth_sort_cfg(int variant) {
  return (variant & TRASE_VARIANT_DEPTH32) ? DepthSortCfg{32, 8, 4, 0} : DepthSortCfg{27, 9, 3, 1};
}
constexpr int DEPTH_MAX_DIGIT_BITS = 9;                                 // histogram buffers are sized for this
constexpr uint32_t DEPTH27_BASE = 0x3e4ccccdu, DEPTH27_SAT = 0x07fffffeu, DEPTH27_DEAD = 0x07ffffffu;
__host__ __device__ inline uint32_t depth_sort_key(float z, bool key27) {     // key of a LIVE Gaussian
  const uint32_t bits = __builtin_bit_cast(uint32_t, z);
  if (!key27) return bits;
  const uint32_t b = bits - DEPTH27_BASE;
  return b < DEPTH27_SAT ? b : DEPTH27_SAT;
}
inline uint32_t depth_dead_key(bool key27) { return key27 ? DEPTH27_DEAD : 0xffffffffu; }
// key_or / key_dead: see RawFwdArgs (the two-view forward puts the view index into the key's sign bit)
int launch_preprocess_fwd_raw(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastRawInputs& raw, int32_t* radii,
                              const GeomBuf& g, uint32_t* depth_keys, bool key27, uint32_t key_or = 0u, uint32_t key_dead = 0xffffffffu);

// pack_bits: 0 = never pack; jb = pack list values as (id << jb | j) when every Gaussian has fewer than 2^jb pairs
// strip mode: (depth key, id) of the Gaussians with a pair, ascending ids, then the ids without one; hdr[HDR_WORDS - 1] = live count
int launch_compact_live(const LaunchCtx& c, const GeomBuf& g, int P, const PreBuf& t, const uint32_t* keys_raw,
                        uint32_t* keys_out, uint32_t* ids_out);
int launch_scan_tiles(const LaunchCtx& c, const GeomBuf& g, const uint32_t* sorted_ids, int P, const PreBuf& t, uint32_t cap,
                      const int32_t* radii, int gx, int gy, int pack_bits = 0);
int launch_emit_pairs(const LaunchCtx& c, const TraseRastSettings& s, const GeomBuf& g, const int32_t* radii,
                      const uint32_t* sorted_ids, int P, const PreBuf& t, uint32_t* keys, uint32_t* pair_gauss, uint32_t cap,
                      uint2* ranges_to_clear = nullptr, uint32_t* vals = nullptr);
// raw_feats != null: d_feats receives the gradient of the RAW features (backward of f / (||f|| + 1e-9) fused in)
int launch_reduce_rows(const LaunchCtx& c, const GeomBuf& g, const PreBuf& pre, int P, int F, const float* rows,
                       const uint8_t* row_flags, float* acc, float* d_feats, const float* raw_feats = nullptr,
                       int norm_features = 0, int id_begin = -1, int id_end = -1, int live_only = 0);
int launch_tile_ranges(const LaunchCtx& c, const uint32_t* keys, const uint32_t* n_ptr, uint32_t cap, uint2* ranges, int T,
                       uint32_t* dbg = nullptr, bool clear = true);

int launch_render_fwd(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, const TraseRastOutputs& out,
                      const GeomBuf& g, const BinBuf& b, const ImgBuf& im, const uint32_t* pair_gauss = nullptr,
                      uint32_t cap = 0);
int launch_render_fwd_mf(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, const TraseRastOutputs& out,
                         const GeomBuf& g, const BinBuf& b, const ImgBuf& im, const uint32_t* pair_gauss, uint32_t cap);
int launch_render_bwd_gs(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, const GeomBuf& g,
                         const BinBuf& b, const ImgBuf& im, const TraseRastGrads& gr, float* rows, uint8_t* row_flags,
                         const float* out_depth = nullptr);
// half-wave MFMA formulation (32-entry chunks, 128 VGPRs): the default for F = 32
int launch_render_bwd_hw(const LaunchCtx& c, const TraseRastSettings& s, const TraseRastInputs& in, const GeomBuf& g,
                         const BinBuf& b, const ImgBuf& im, const TraseRastGrads& gr, float* rows, uint8_t* row_flags,
                         size_t flag_bytes, const float* out_depth = nullptr);

}  // namespace trase
